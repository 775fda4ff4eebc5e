#!/usr/bin/env python3
"""Generator of the steady-state loop of olasave_kernel (phaserotate.lv2_amd/csrc/prot_olasave.hip) as gfx950 assembly with
hand-allocated registers: hipcc needs ~150 VGPRs for this loop (52..140 B/lane of scratch at the 128 that four workgroups
per CU allow, and a spilled build is slower than the three-per-CU kernel), the loop itself needs 120.

    python3 tools/gen/gen_olasave.py            -> csrc/prot_olasave_asm.inc  (+ a self-test of every variant in the emulator)
    python3 tools/gen/gen_olasave.py --check    -> self-test only

One asm block = a RUN of n >= 1 consecutive "fast" blocks of one channel-stream (every frame the blocks read lies inside the
input array, 8-byte aligned streams, whole blocks): prologue (addresses, twiddle registers, the first block's 16 rows), then
per block: build (p, q) -> forward transform at the half-shifted index -> constants -> inverse -> mix with the dry rows ->
store, with the next block's new rows prefetched behind the first barrier.  The maths, pass by pass, is tools/model/
olasave_model.py; the C++ kernel around the block handles everything else (history, edges, per-frame rotation) with the
same packed instructions in the same order, so the two paths give the same bits.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "model"))
from isa import Prog, Emu, V, VP, S, SP, Lit, VCC  # noqa: E402

KL1, KG = 272, 17                 # LDS exchange strides (prot_dev.h)
LDS_TWM, LDS_TWF = 34816, 36864   # byte offsets of the W256 tables behind the exchange area
LDS_GCAR, LDS_GNX = 32768, 33792  # core blocks: the gathered frames, one float per thread in lane layout, behind the 16 staged rows
MODE_ROT, MODE_HILBERT, MODE_SWEEP = 0, 2, 3   # (SWEEP: the Hilbert branch tested against a bound and its survivors compacted, round 4)

# ---- SGPR plan (the asm block clobbers s34..s99; s32 and s100.. are reserved by the compiler) ----
S_IN, S_OUT = SP(34), SP(36)      # frame (row 0, thread 0).q of the current block; first output frame of the current block
S_NBLK, S_LDS = S(38), S(39)
S_SV = SP(40)
S_ROT = SP(42)                    # (ca, sa * gain): one pair, the half picked by op_sel
S_C32 = {k: SP(44 + 2 * (k - 1)) for k in range(1, 16)}            # s44..s73
S_RR, S_W1, S_W3, S_W9 = SP(74), SP(76), SP(78), SP(80)            # (R, -R), (C1, -S1), (S1, -C1), (-C1, S1)
S_TWI = [SP(82 + 2 * i) for i in range(6)]                         # s82..s93: W8192^{1,2,3}, W2048^{1,2,3}
S_T = [S(94), S(95), S(96), S(97)]                                 # readlane temporaries
S_A0 = SP(98)                                                      # address temporary
SGPR_CLOBBER = range(34, 100)

C1, S1, RR = 0.92387953251128675613, 0.38268343236508977173, 0.70710678118654752440


class Gen:
    def __init__(self, HS, MODE, NA=1, opts=(), kind="run"):
        """opts: timing experiments only (results wrong): "nolds" no LDS traffic and no barriers, "nomem" no stream loads / stores,
        "novalu" no packed arithmetic; "tw2" two twiddle batches in flight; "burstld" the prefetch loads in one burst behind the
        barrier instead of woven into the butterflies; "spreadst" stores issued per DFT-4 of the last stage; "nt" non-temporal
        stream accesses; "prio1" / "prio2" raised priority in front of the first barrier / behind the second"""
        self.HS, self.MODE, self.NA = HS, MODE, NA
        self.opts = set(opts)
        # "run": n >= 1 blocks streamed from / to global memory; "core": ONE block whose 16 rows (and gathered frames) the
        # C++ code has staged in the LDS exchange area, outputs back into it (edge blocks: history, silence, partial stores)
        self.kind = kind
        self.NT = 16 - HS
        self.R0 = self.NT // 2
        self.BLK = 512 * HS
        P = self.P = Prog()
        self.sw_rows, self.sw_label = [], 0
        if "nolds" in self.opts:
            P.ds_write_b64 = lambda *a, **k: None
            P.ds_read_b64 = lambda *a, **k: None
            P.s_barrier = lambda *a, **k: None
        if "novalu" in self.opts:
            P.pk = lambda *a, **k: None
        n = 0
        self.D = [VP(2 * k) for k in range(16)]
        n = 32
        self.K = {}
        for r in range(self.R0, 16):
            self.K[r] = VP(n)
            n += 2
        self.N = []
        for i in range(HS):
            self.N.append(VP(n))
            n += 2
        self.TW = [None] + [VP(n + 2 * i) for i in range(3)] + [None] + [VP(n + 6 + 2 * i) for i in range(3)]  # A[1..3] at [1..3], B[1..3] at [5..7]
        n += 12
        self.RHO = [VP(n + 2 * i) for i in range(NA)]
        n += 2 * NA
        self.G_CAR, self.G_NX = V(n), V(n + 1)
        self.A_W1, self.A_R1, self.A_X2, self.A_TL, self.VOFF, self.VG = V(n + 2), V(n + 3), V(n + 4), V(n + 5), V(n + 6), V(n + 7)
        n += 8
        self.TMP = []
        cap = 119   # the compiler keeps its own live values (and SGPR spill lanes) in what is left of 128
        while n + 1 <= cap:
            self.TMP.append(VP(n))
            n += 2
        self.n_vgpr = n
        if MODE == MODE_SWEEP:    # the running max |x| of the frames that pass through (angle 0, the raw peak) lives across blocks
            self.V_RAW = V(120)   # (one register beyond the plan: the sweep instantiation is checked to stay without scratch)
        assert len(self.TMP) >= 6, "temporaries"
        # roles of the temporaries
        self.TB = self.TMP[:4]                 # twiddle batch (also: derived inverse twiddle, epilogue products)
        self.TR = self.TMP[4:5]                # complex-multiply scratch
        self.TT = self.TMP[5:9] if len(self.TMP) >= 9 else self.TMP[5:6] * 4  # radix-4 scratch (one per butterfly in flight)
        self.nbf = len(set(t.n for t in self.TT))

    # ------------------------------------------------------------ packed helpers (prot_fft.h's pk32 ops)
    def add(self, d, a, b): self.P.pk("add", d, [a, b])
    def sub(self, d, a, b): self.P.pk("add", d, [a, b], neg_lo=[0, 1], neg_hi=[0, 1])
    def add_mj(self, d, a, b): self.P.pk("add", d, [a, b], op_sel=[0, 1], op_sel_hi=[1, 0], neg_hi=[0, 1])
    def sub_mj(self, d, a, b): self.P.pk("add", d, [a, b], op_sel=[0, 1], op_sel_hi=[1, 0], neg_lo=[0, 1])

    def cmul_ops(self, d, a, w, DIR, r):
        """d = a * w (DIR > 0) or a * conj(w) (DIR < 0); r = scratch pair; returns two thunks"""
        P = self.P
        if DIR > 0:
            return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 0], op_sel_hi=[0, 1]),
                    lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 1, 0], op_sel_hi=[1, 0, 1], neg_lo=[0, 1, 0])]
        return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 0], op_sel_hi=[0, 1], neg_hi=[0, 1]),
                lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 1, 0], op_sel_hi=[1, 0, 1])]

    def cmul(self, d, a, w, DIR, r):
        for f in self.cmul_ops(d, a, w, DIR, r):
            f()

    def radix4_ops(self, a, b, c, d, DIR, crot, T):
        """pkt_radix4 in place with one scratch pair (same operations and operand order)"""
        ops = []
        if not crot:
            ops.append(lambda: self.add(T, a, c))
            ops.append(lambda: self.sub(a, a, c))
        elif DIR > 0:
            ops.append(lambda: self.add_mj(T, a, c))
            ops.append(lambda: self.sub_mj(a, a, c))
        else:
            ops.append(lambda: self.sub_mj(T, a, c))
            ops.append(lambda: self.add_mj(a, a, c))
        ops.append(lambda: self.add(c, b, d))      # t2
        ops.append(lambda: self.sub(d, b, d))      # t3
        if DIR > 0:
            ops.append(lambda: self.add_mj(b, a, d))
            ops.append(lambda: self.sub_mj(d, a, d))
        else:
            ops.append(lambda: self.sub_mj(b, a, d))
            ops.append(lambda: self.add_mj(d, a, d))
        ops.append(lambda: self.add(a, T, c))
        ops.append(lambda: self.sub(c, T, c))
        return ops

    @staticmethod
    def interleave(lists):
        out, i = [], 0
        lists = [list(l) for l in lists]
        while any(lists):
            for l in lists:
                if l:
                    out.append(l.pop(0))
        return out

    def run_ops(self, ops):
        """run the thunks; while `pending` thunks are queued (the prefetch of the next block's rows: gen_olasave8k.py, Gen8kDyn with
        "spread") one of them is issued after every `pend_every` arithmetic thunks, across calls"""
        pend = getattr(self, "pending", None)
        for f in ops:
            f()
            if pend and not getattr(self, "no_weave", False):
                self.pend_n = getattr(self, "pend_n", 0) + 1
                if self.pend_n >= self.pend_every:
                    self.pend_n = 0
                    pend.pop(0)()

    def flush_pending(self):
        pend = getattr(self, "pending", None)
        while pend:
            pend.pop(0)()

    @staticmethod
    def weave(main, extra, every):
        """`extra` thunks spread through `main`, one after every `every` of them (the rest at the end)"""
        out, extra = [], list(extra)
        for i, f in enumerate(main):
            out.append(f)
            if extra and (i + 1) % every == 0:
                out.append(extra.pop(0))
        return out + extra

    def stage1_ops(self, v, DIR):
        bf = [self.radix4_ops(v[be], v[4 + be], v[8 + be], v[12 + be], DIR, False, self.TT[be % 4]) for be in range(4)]
        return self.interleave(bf) if self.nbf >= 4 else sum(bf, [])

    def mul_w16_ops(self, x, DIR, m, r):
        """pkt_mul_w16: x *= W16^m (conjugate for DIR < 0), m in {1, 2, 3, 6, 9}.  For m = 2 and 6 only the rotation part
        x -+ j x is done here (in place); the real factor +-R rides in the butterfly that consumes x (stage2_ops: one fma
        instead of a multiply and an add)"""
        if m == 2:
            return [lambda: (self.add_mj if DIR > 0 else self.sub_mj)(x, x, x)]
        if m == 6:
            return [lambda: (self.sub_mj if DIR > 0 else self.add_mj)(x, x, x)]
        w = {1: S_W1, 3: S_W3, 9: S_W9}[m]
        return self.cmul_ops(x, x, w, DIR, r)

    def stage2_ops(self, v, ka, DIR, T, need=(0, 1, 2, 3)):
        """DFT-4 number ka of the second stage on (v[4 ka], .., v[4 ka + 3]) -> X[ka + 4 la] in v[4 ka + la], with the pending
        real factors of stage1_twiddle_ops folded in (v[6]: +R, v[9]: +R, v[11]: -R, v[14]: -R; v[10] carries W16^4 = -+j);
        need = the outputs la that are wanted (the others' instructions are left out)"""
        P = self.P
        a, b, c, d = v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]
        n02, n13 = (0 in need or 2 in need), (1 in need or 3 in need)
        ops = []
        fma_c = lambda dst, sel: (lambda: P.pk("fma", dst, [c, S_RR, a], op_sel=[0, sel, 0], op_sel_hi=[1, sel, 1]))
        if ka == 1 or ka == 3:           # c = R' r with R' = +R (ka 1) / -R (ka 3): t0 = a + R' r, t1 = a - R' r
            sel = 0 if ka == 1 else 1
            if n02: ops.append(fma_c(T, sel))
            if n13: ops.append(fma_c(a, 1 - sel))
        elif ka == 2:                    # c carries -+j
            if DIR > 0:
                if n02: ops.append(lambda: self.add_mj(T, a, c))
                if n13: ops.append(lambda: self.sub_mj(a, a, c))
            else:
                if n02: ops.append(lambda: self.sub_mj(T, a, c))
                if n13: ops.append(lambda: self.add_mj(a, a, c))
        else:
            if n02: ops.append(lambda: self.add(T, a, c))
            if n13: ops.append(lambda: self.sub(a, a, c))
        if ka == 2:                      # b = R rb, d = -R rd: d' = -R rd; t2 = R rb + d'; t3 = R rb - d'
            ops.append(lambda: P.pk("mul", d, [d, S_RR], op_sel=[0, 1], op_sel_hi=[1, 1]))
            if n02: ops.append(lambda: P.pk("fma", c, [b, S_RR, d], op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1]))
            if n13: ops.append(lambda: P.pk("fma", d, [b, S_RR, d], op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1], neg_lo=[0, 0, 1], neg_hi=[0, 0, 1]))
        else:
            if n02: ops.append(lambda: self.add(c, b, d))      # t2
            if n13: ops.append(lambda: self.sub(d, b, d))      # t3
        if DIR > 0:
            if 1 in need: ops.append(lambda: self.add_mj(b, a, d))
            if 3 in need: ops.append(lambda: self.sub_mj(d, a, d))
        else:
            if 1 in need: ops.append(lambda: self.sub_mj(b, a, d))
            if 3 in need: ops.append(lambda: self.add_mj(d, a, d))
        if 0 in need: ops.append(lambda: self.add(a, T, c))
        if 2 in need: ops.append(lambda: self.sub(c, T, c))
        return ops

    def stage1_twiddle_ops(self, v, DIR, rs):
        """pkt_stage1_twiddles; rs = scratch pairs to rotate through"""
        spec = [(5, 1), (6, 2), (7, 3), (9, 2), (11, 6), (13, 3), (14, 6), (15, 9)]
        lists = []
        for i, (k, m) in enumerate(spec):
            lists.append(self.mul_w16_ops(v[k], DIR, m, rs[i % len(rs)]))
        # group so that scratch pairs are not reused while live: interleave in groups of len(rs)
        out = []
        for g in range(0, len(lists), len(rs)):
            out += self.interleave(lists[g:g + len(rs)])
        return out

    def stage1(self, v, DIR):
        bf = [self.radix4_ops(v[be], v[4 + be], v[8 + be], v[12 + be], DIR, False, self.TT[be % 4]) for be in range(4)]
        if self.nbf >= 4:
            self.run_ops(self.interleave(bf))
        else:
            for l in bf:
                self.run_ops(l)

    def stage2_lists(self, v, DIR, need=None):
        return [self.stage2_ops(v, ka, DIR, self.TT[ka % 4], need[ka] if need else (0, 1, 2, 3)) for ka in range(4)]

    def scratch_pairs(self):
        return self.TR + self.TB

    # ------------------------------------------------------------ phases
    def prologue(self):
        P, HS, NT = self.P, self.HS, self.NT
        P.comment("---- prologue: addresses")
        t = self.TMP[0].lo   # holds t (copied from the input operand by the wrapper text)
        tl, th, tmp = self.TMP[0].hi, self.TMP[1].lo, self.TMP[1].hi
        P.v_and(tl, Lit(15), t)
        P.v_lshr(th, Lit(4), t)
        P.v_lshl(self.VOFF, Lit(3), t)                          # 8 t
        P.v_add_u32(self.A_W1, S_LDS, self.VOFF)
        P.v_mul_u24(tmp, Lit(KL1 * 8), th)
        P.v_add_u32(tmp, S_LDS, tmp)                            # lds + 8 * 272 th
        P.v_lshl(self.A_TL, Lit(3), tl)                         # 8 tl
        P.v_add_u32(self.A_R1, tmp, self.A_TL)
        P.v_mul_u24(self.A_X2, Lit(KG * 8), tl)
        P.v_add_u32(self.A_X2, tmp, self.A_X2)
        P.v_add_u32(self.A_TL, S_LDS, self.A_TL)
        # gather offsets: lane l of a wave fetches the frame in front of row (l)'s first sample of this wave
        lane, wv = self.TMP[2].lo, self.TMP[2].hi
        P.v_and(lane, Lit(63), t)
        P.v_lshr(wv, Lit(6), t)
        P.v_lshl(wv, Lit(9), wv)                                # 512 B per wave
        # VG = 2048 * (lane < HS ? lane : 0) + 512 wave        (steady state: lanes 0..HS-1 <-> new rows NT..15)
        # lane < HS  <=>  select by arithmetic: min(lane, HS) then zero if == HS ... done with a small table of moves instead:
        # v_cmp is avoided (VCC): rows = lane clamped by repeated and/shift is awkward -> use v_mul_u24 with a 0/1 mask from (HS-1-lane) >> 31
        m = self.TMP[3].lo
        P.v_add_u32(m, Lit(-(HS)), lane)                        # lane - HS  (negative if lane < HS)
        P.v_lshr(m, Lit(31), m)                                 # 1 if lane < HS
        P.v_mul_u24(m, m, lane)                                 # lane or 0
        P.v_lshl(m, Lit(11), m)
        P.v_add_u32(self.VG, m, wv)
        # first block: rows 0..NT-1 come with gathers in lanes HS-NT .. HS-1 (where the steady state leaves the carried ones)
        vg0 = self.TMP[3].hi
        m2, m3 = self.TMP[4].lo, self.TMP[4].hi
        P.v_add_u32(m2, Lit(-(HS - NT)), lane)                  # lane - (HS - NT)
        P.v_add_u32(m3, Lit(-(HS)), lane)
        P.v_lshr(m3, Lit(31), m3)                               # lane < HS
        P.v_lshr(m, Lit(31), m2)                                # lane < HS-NT  -> invalid
        P.v_add_u32(m, Lit(-1), m)                              # 0 if invalid else 0xffffffff ... (m was 1 -> 0, 0 -> -1)
        P.v_and(m2, m, m2)                                      # lane - (HS-NT) or 0
        P.v_mul_u24(m2, m3, m2)
        P.v_lshl(m2, Lit(11), m2)
        P.v_add_u32(vg0, m2, wv)
        P.comment("---- prologue: twiddle registers, |B| factors")
        for j in range(3):
            P.s_add_u32(S_A0.lo, S_SV.lo, Lit(4096 * j))
            P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
            for e in range(2):
                i = 2 * j + e                                   # table rows 0..5: A[1..3], B[1..3]
                dst = self.TW[1 + i] if i < 3 else self.TW[5 + i - 3]
                P.global_load_dwordx2(dst, self.VOFF, S_A0, 2048 * e)
        for i in range(self.NA):
            off = (13 + i) * 2048
            P.s_add_u32(S_A0.lo, S_SV.lo, Lit(off))
            P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
            P.global_load_dwordx2(self.RHO[i], self.VOFF, S_A0, 0)
        # the W256 tables of the middle passes go to LDS behind the exchange area: half-shifted (forward, table row 12) and plain
        # (inverse, row 14); every wave stores its own 64 entries, the first workgroup barrier of the block orders them
        P.s_add_u32(S_A0.lo, S_SV.lo, Lit(12 * 2048))
        P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
        P.global_load_dwordx2(self.TMP[6], self.VOFF, S_A0, 0)
        P.s_add_u32(S_A0.lo, S_SV.lo, Lit(14 * 2048))
        P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
        P.global_load_dwordx2(self.TMP[7] if len(self.TMP) > 7 else self.TMP[1], self.VOFF, S_A0, 0)
        self.tab_regs = (self.TMP[6], self.TMP[7] if len(self.TMP) > 7 else self.TMP[1])
        if self.kind != "core":
            # the first block's rows are requested right behind the tables, in front of everything that waits for a table:
            # vector memory retires in order, so the waits below are for the (older) table loads only and the two round
            # trips of a launch's start overlap (one-block launches -- BASELINE config 3 -- are little else)
            P.comment("---- prologue: the first block's rows (requested early)")
            # rows 0..NT-1 -> K[HS .. 15] (the registers the steady state carries them in); gathers -> G_CAR lanes HS-NT..
            for mpair in range(NT // 2):
                P.s_add_u32(S_A0.lo, S_IN.lo, Lit(4096 * mpair))
                P.s_addc_u32(S_A0.hi, S_IN.hi, Lit(0))
                for e in range(2):
                    P.global_load_dwordx2(self.K[HS + 2 * mpair + e], self.VOFF, S_A0, 2048 * e)
            P.global_load_dword(self.G_CAR, vg0, S_IN, -4)
            self.load_new_rows(0)
        self.scale_constants()
        P.ds_write_b64(self.A_W1, self.tab_regs[0], LDS_TWF)
        P.ds_write_b64(self.A_W1, self.tab_regs[1], LDS_TWM)
        P.comment("---- prologue: the first block's rows")
        if self.kind == "core":
            # staged by the C++ code: row r's (q, c) pair of thread t at slot r * 256 + t, the gathered frames in lane layout
            for r in range(NT):
                P.ds_read_b64(self.K[HS + r], self.A_W1, 2048 * r)
            for i in range(HS):
                P.ds_read_b64(self.N[i], self.A_W1, 2048 * (NT + i))
            a4 = self.TMP[2].lo
            P.v_lshl(a4, Lit(2), t)
            P.v_add_u32(a4, S_LDS, a4)
            P.ds_read_b32(self.G_CAR, a4, LDS_GCAR)
            P.ds_read_b32(self.G_NX, a4, LDS_GNX)
            P.s_waitcnt(vmcnt=0, lgkmcnt=0)
            P.s_barrier()   # every wave has its rows: the exchange area may be written
            P.comment("---- prologue done")
            return
        P.comment("---- prologue done")

    def scale_constants(self):
        """sa * gain (S_ROT.hi) into the 15 constants i W32^{-k3} and into the |B| factor of row 0 -- every value of the block
        passes through exactly one of them, so the mix at the end is one fma per pair (Hilbert-only blocks: none)"""
        P = self.P
        # through VGPRs, all reads of the constants first, all v_readfirstlane last: on gfx950 a VALU instruction that reads an
        # SGPR right behind a VALU write of an SGPR (v_readfirstlane s44; v_mul_f32 .., s45, ..) gets the wrong value (measured:
        # tools/gen/isa.py checks for it); the D registers are free here
        sa = self.TMP[5].lo
        regs = [(S_C32[k].lo, self.D[k].lo) for k in range(1, 16)] + [(S_C32[k].hi, self.D[k].hi) for k in range(1, 16)]
        P.v_mov(sa, S_ROT.hi)
        for sreg, v in regs:
            P.v_mov(v, sreg)
        for sreg, v in regs:
            P.v_mul_f32(v, v, sa)
        P.v_mul_f32(self.RHO[0].lo, sa, self.RHO[0].lo)
        for sreg, v in regs:
            P.v_readfirstlane(sreg, v)
        P.s_nop(4)

    def load_new_rows(self, blk_ahead, thunks=False, soff=None):
        """rows NT..15 of the block `blk_ahead` blocks after the current one -> N, their gathers -> G_NX.
        thunks: return the instructions as a list of closures (to be woven into arithmetic); soff: an SGPR holding the byte
        offset of that block instead of a constant"""
        P, HS, NT = self.P, self.HS, self.NT
        if "nomem" in self.opts and blk_ahead:
            return []
        nt = "nt" in self.opts or "ntld" in self.opts
        base = (0 if soff is not None else 4 * self.BLK * blk_ahead) + 2048 * NT
        ops = []
        def setbase(extra):
            def f():
                if soff is not None:
                    P.s_add_u32(S_A0.lo, S_IN.lo, soff)
                    P.s_addc_u32(S_A0.hi, S_IN.hi, Lit(0))
                    P.s_add_u32(S_A0.lo, S_A0.lo, Lit(base + extra))
                    P.s_addc_u32(S_A0.hi, S_A0.hi, Lit(0))
                else:
                    P.s_add_u32(S_A0.lo, S_IN.lo, Lit(base + extra))
                    P.s_addc_u32(S_A0.hi, S_IN.hi, Lit(0))
            return f
        for mpair in range(HS // 2):
            for e in range(2):
                def f(mpair=mpair, e=e):
                    if e == 0:
                        setbase(4096 * mpair)()
                    P.global_load_dwordx2(self.N[2 * mpair + e], self.VOFF, S_A0, 2048 * e, nt=nt)
                ops.append(f)
        def g():
            setbase(0)()
            P.global_load_dword(self.G_NX, self.VG, S_A0, -4, nt=nt)
        ops.append(g)
        if thunks:
            return ops
        for f in ops:
            f()
        return []

    def build_pq(self):
        """D[k] = (p, q) of block row k: q = own pair's .x, p = the .y of lane t-1 (lane 0 of a wave: the gathered frame);
        then the (q, c) pairs move to where the block keeps them"""
        P, HS, NT, R0 = self.P, self.HS, self.NT, self.R0
        src = lambda k: self.K[HS + k] if k < NT else self.N[k - NT]
        for g in range(0, 16, 4):
            for k in range(g, g + 4):   # lane 0 <- the gathered frame of row k (lanes 1..3: junk, overwritten below)
                gv, lane = (self.G_CAR, HS - NT + k) if k < NT else (self.G_NX, k - NT)
                if lane == 0:
                    P.v_mov(self.D[k].lo, gv)
                else:
                    P.v_mov_dpp_rowshl(self.D[k].lo, gv, lane)
            for k in range(g, g + 4):
                P.v_mov(self.D[k].hi, src(k).lo)
            for k in range(g, g + 4):   # (two wait states after the VALU write of D[k].lo: the other moves are in between)
                P.v_mov_dpp_shr1(self.D[k].lo, src(k).hi)
        for r in range(R0, NT):         # context rows that are also dry rows of this block
            P.v_mov(self.K[r].lo, self.K[HS + r].lo)
            P.v_mov(self.K[r].hi, self.K[HS + r].hi)
        for i in range(HS):
            P.v_mov(self.K[NT + i].lo, self.N[i].lo)
            P.v_mov(self.K[NT + i].hi, self.N[i].hi)
        P.v_mov(self.G_CAR, self.G_NX)

    def fwd_pass1(self):
        """pkt_fft16_fwd_pruned_twout<16> on rows, outputs straight into the first exchange"""
        P = self.P
        v = self.D
        A, B = self.TW[0:4], self.TW[4:8]
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        sc = self.scratch_pairs()
        lists = []
        for ka in range(1, 4):
            for be in range(4):
                lists.append(self.cmul_ops(v[4 * ka + be], v[4 * ka + be], A[ka], 1, sc[len(lists) % len(sc)]))
        for g in range(0, len(lists), len(sc)):
            self.run_ops(self.interleave(lists[g:g + len(sc)]))
        # stage 2, then B on the outputs X[ka + 4 la] (at v[4 ka + la]) and the store of logical k = ka + 4 la
        bfl = self.stage2_lists(v, 1)
        if self.nbf >= 4:
            self.run_ops(self.interleave(bfl))
        else:
            for l in bfl:
                self.run_ops(l)
        n = 0
        for ka in range(4):
            P.ds_write_b64(self.A_W1, v[4 * ka], (ka) * KL1 * 8)
        for la in range(1, 4):
            for ka in range(4):
                x = v[4 * ka + la]
                self.cmul(x, x, B[la], 1, sc[n % len(sc)])
                n += 1
                P.ds_write_b64(self.A_W1, x, (ka + 4 * la) * KL1 * 8)

    def fft16_plain(self, v, DIR, before_stage2=None):
        """pkt_fft16 on the logical list v; returns the logical list of the outputs (a renaming)"""
        self.stage1(v, DIR)
        self.run_ops(self.stage1_twiddle_ops(v, DIR, self.scratch_pairs()))
        if before_stage2:
            before_stage2()
        return v

    def fwd_pass2(self):
        P = self.P
        v = self.D
        order = [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
        for k in order:
            P.ds_read_b64(v[k], self.A_R1, 128 * k)
        return v

    def twiddled_exchange(self, v, DIR, tw_base, waddr, wstride, raddr, rstride):
        """second half of a middle pass: stage 2 one DFT-4 at a time, outputs X[ka + 4 la] times the W256 table entry, stored
        to the wave-local exchange; then the transposed read back into D"""
        P = self.P
        two = "tw2" in self.opts and len(self.TMP) >= 10
        sets = [self.TB, self.TT[1:4] + [self.TMP[9]]] if two else [self.TB]
        def fetch(ka):
            tb = sets[ka % len(sets)]
            for la in range(4):
                k = ka + 4 * la
                if k:
                    P.ds_read_b64(tb[la], self.A_TL, tw_base + 128 * k)
        bfl = [self.stage2_ops(v, ka, DIR, self.TT[0]) for ka in range(4)]
        r = self.TR[0]
        fetch(0)
        if two:
            fetch(1)
        for ka in range(4):
            tb = sets[ka % len(sets)]
            self.run_ops(bfl[ka])
            for la in range(4):
                k = ka + 4 * la
                x = v[4 * ka + la]
                if k:
                    self.cmul(x, x, tb[la], DIR, r)
                P.ds_write_b64(waddr, x, wstride * k)
            if ka + len(sets) < 4:
                fetch(ka + len(sets))
        order = [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
        for k in order:
            P.ds_read_b64(self.D[k], raddr, rstride * k)

    def pointwise(self, v):
        """v = logical outputs k3 of the last forward pass: times i W32^{-k3}, and the |B| factors of the rows next to DC"""
        P = self.P
        sc = self.scratch_pairs()
        # k3 = 0: (i rho0) v
        P.pk("mul", v[0], [v[0], self.RHO[0]], op_sel=[1, 0], op_sel_hi=[0, 0], neg_lo=[1, 0])
        lists = []
        for k in range(1, 16):
            if k == 8:
                lists.append([lambda x=v[8]: P.pk("mul", x, [x, S_C32[8]])])
            else:
                lists.append(self.cmul_ops(v[k], v[k], S_C32[k], 1, sc[len(lists) % len(sc)]))
        for g in range(0, len(lists), len(sc)):
            self.run_ops(self.interleave(lists[g:g + len(sc)]))
        P.pk("mul", v[15], [v[15], self.RHO[0]], op_sel=[0, 1], op_sel_hi=[1, 1])
        if self.NA == 2:
            P.pk("mul", v[1], [v[1], self.RHO[1]], op_sel=[0, 0], op_sel_hi=[1, 0])
            P.pk("mul", v[14], [v[14], self.RHO[1]], op_sel=[0, 1], op_sel_hi=[1, 1])

    def natural(self, v):
        """logical outputs of pkt_fft16: X[ka + 4 la] sits in v[4 ka + la]"""
        return [v[4 * (k % 4) + k // 4] for k in range(16)]

    def inv_pass3(self):
        """pkt_fft16_inv_twin with the plain twiddles derived from the half-shifted ones, one at a time"""
        P = self.P
        v = self.D
        A, B = self.TW[0:4], self.TW[4:8]
        w, r = self.TB[0], self.TR[0]
        sc = self.TB[1:] + self.TT[:1]
        for al in range(1, 4):      # inputs n = 4 al + be: conj(W^{4 t al})
            self.cmul(w, B[al], S_TWI[3 + al - 1], 1, r)
            ls = [self.cmul_ops(v[4 * al + be], v[4 * al + be], w, -1, sc[be % len(sc)]) for be in range(4)]
            self.run_ops(self.interleave(ls) if len(set(x.n for x in sc)) >= 4 else sum(ls, []))
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()[1:] if False else [self.TR[0]] + self.TB[1:]))
        for be in range(1, 4):      # y[be][ka] sits in v[4 ka + be]: conj(W^{t be})
            self.cmul(w, A[be], S_TWI[be - 1], 1, r)
            ls = [self.cmul_ops(v[4 * ka + be], v[4 * ka + be], w, -1, sc[ka % len(sc)]) for ka in range(4)]
            self.run_ops(self.interleave(ls) if len(set(x.n for x in sc)) >= 4 else sum(ls, []))
        rows = range(self.R0, self.R0 + self.HS)   # only these output rows exist (the others are circular wrap-around)
        bfl = self.stage2_lists(v, -1, need=[[la for la in range(4) if ka + 4 * la in rows] for ka in range(4)])
        if "spreadst" in self.opts and self.kind != "core":
            # each DFT-4 of the last stage yields rows ka, ka + 4, ka + 8, ka + 12: mixed and stored while the next one computes
            u = self.natural(v)
            for ka in range(4):
                self.run_ops(bfl[ka])
                for la in range(4):
                    row = ka + 4 * la
                    if self.R0 <= row < self.R0 + self.HS:
                        self.epilogue_row(u, row - self.R0)
            self.stored = True
            return u
        if self.nbf >= 4:
            self.run_ops(self.interleave(bfl))
        else:
            for l in bfl:
                self.run_ops(l)
        return self.natural(v)

    def epilogue_row(self, u, k):
        """output row R0 + k: out = ca dry + sa u (or sa u), stored as aligned pairs"""
        P, R0 = self.P, self.R0
        nt = "nt" in self.opts or "ntst" in self.opts
        m, e = k // 2, k % 2
        x = u[R0 + k]
        if self.MODE == MODE_SWEEP:
            self.sweep_row(self.K[R0 + k], x)
            return
        if self.MODE == MODE_ROT:   # (sa * gain rides in the constants of the pointwise stage: scale_constants)
            P.pk("fma", x, [self.K[R0 + k], S_ROT, x], op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1])
        if self.kind == "core":
            P.ds_write_b64(self.A_W1, x, 2048 * k)
        elif "nomem" not in self.opts:
            P.s_add_u32(S_A0.lo, S_OUT.lo, Lit(4096 * m))
            P.s_addc_u32(S_A0.hi, S_OUT.hi, Lit(0))
            P.global_store_dwordx2(self.VOFF, x, S_A0, 2048 * e, nt=nt)

    def sweep_row(self, d, h):
        """MODE_SWEEP: one output row is not stored -- d (dry pair) and h (Hilbert pair) of a frame go to the survivor list of this
        wave (a plain append at the SGPR cursor S_OUT: lanes compacted by ballot + mbcnt) if d^2 + h^2 exceeds the bound in
        S_ROT.lo: |ca d + sa h| <= sqrt(d^2 + h^2) for every angle, so the other frames cannot raise any maximum that has reached the
        bound (prot_sweep.hip evaluates the list).  max |d| rides along (the raw peak)."""
        self.sw_rows.append((d, h))

    def sweep_flush(self):
        """the rows sweep_row collected, tested one behind the next row's norm (round 4, last change): the two compares of a row,
        its raw maximum and the NEXT row's packed norm (four instructions between a compare and the first reader of its lane mask --
        isa.py: lane-mask hazard), the two population counts, and the transposing moves, lane ranks and appends only if the wave has a
        survivor in this row at all -- on material that prunes well most rows of most blocks have none, and the ten vector
        instructions and two masked-off stores of a row were a tenth of the loop.  The appends sit behind a branch, so the loop top
        cannot count them: every load is waited for here (they were issued passes ago), and the top waits for nothing in flight."""
        P = self.P
        rows, self.sw_rows = self.sw_rows, []
        if not rows:
            return
        P.s_waitcnt(vmcnt=0)
        r2s, prs, idxs = (self.TMP[0], self.TMP[4]), (self.TMP[1], self.TMP[3]), (self.TMP[2].lo, self.TMP[2].hi)
        masks, cnts, tot = (VCC, S_A0), (S_T[0], S_T[1]), S_T[2]   # (no store of this mode goes through the address temporary)

        def norm(k):
            d, h = rows[k]
            P.pk("mul", r2s[k % 2], [d, d])
            P.pk("fma", r2s[k % 2], [h, h, r2s[k % 2]])
        norm(0)
        for k, (d, h) in enumerate(rows):
            for j, half in enumerate(("lo", "hi")):
                P.v_cmp_lt_f32(masks[j], S_ROT.lo, getattr(r2s[k % 2], half))
            P.v_max3_abs(self.V_RAW, d.lo, d.hi)
            if k + 1 < len(rows):
                norm(k + 1)
            else:   # (the last row: two of its moves as the filler)
                P.v_mov(prs[0].lo, d.lo)
                P.v_mov(prs[0].hi, h.lo)
            for j in range(2):
                P.s_bcnt1(cnts[j], masks[j])
            P.s_add_u32(tot, cnts[0], cnts[1])
            P.s_cmp_eq_u32(tot, Lit(0))
            lab = "L_sw%d" % self.sw_label
            self.sw_label += 1
            P.s_cbranch_scc1(lab)
            mark = len(P.vmq)
            for j, half in enumerate(("lo", "hi")):
                P.v_mov(prs[j].lo, getattr(d, half))
                P.v_mov(prs[j].hi, getattr(h, half))
            for j in range(2):
                P.v_mbcnt(idxs[j], masks[j])
                P.v_lshl(idxs[j], Lit(3), idxs[j])
                P.s_lshl_b32(cnts[j], cnts[j], Lit(3))
            for j in range(2):
                P.exec_from(masks[j])
                P.global_store_dwordx2(idxs[j], prs[j], S_OUT, 0)
                P.exec_all()
                P.s_add_u32(S_OUT.lo, S_OUT.lo, cnts[j])
                P.s_addc_u32(S_OUT.hi, S_OUT.hi, Lit(0))
            P.label(lab)
            del P.vmq[mark:]   # (conservative for whoever waits next: the appends may not have been issued)

    def epilogue(self, u):
        for k in range(self.HS):
            self.epilogue_row(u, k)

    def loop(self):
        P, HS = self.P, self.HS
        core = self.kind == "core"
        if not core:
            P.s_waitcnt(vmcnt=0)
            P.label("L_top")
            # whichever way the top is reached, what may still be in flight: the previous block's HS stores (younger than its
            # prefetch, which the wait below therefore completes)
            n_st = 0 if self.MODE == MODE_SWEEP else HS   # (SWEEP: sweep_flush has waited for the loads; appends are not counted)
            if self.MODE == MODE_SWEEP:
                P.s_waitcnt(lgkmcnt=0)
            else:
                P.s_waitcnt(vmcnt=n_st, lgkmcnt=0)
            P.reset_queues(vm_stores=n_st)
        P.comment("---- (p, q) pairs")
        if "prio1" in self.opts:
            P.s_setprio(3)
        self.build_pq()
        P.comment("---- forward pass 1 + exchange 1")
        self.fwd_pass1()
        P.s_waitcnt(lgkmcnt=0)
        if "prio1" in self.opts:
            P.s_setprio(0)
        P.s_barrier()
        P.comment("---- forward pass 2")
        v = self.fwd_pass2()
        # the next block's new rows go out right behind the first barrier
        if core:
            self.stage1(v, 1)
            self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        elif "burstld" not in self.opts:
            # ... woven into the butterflies, one load every few instructions; the last block of a run re-reads its own rows
            # (in range by definition) instead of branching around the loads
            P.s_cmp_eq_u32(S_NBLK, Lit(1))
            P.s_cselect(S_T[3], Lit(0), Lit(4 * self.BLK))
            lds_ = self.load_new_rows(1, thunks=True, soff=S_T[3])
            main = self.stage1_ops(v, 1) + self.stage1_twiddle_ops(v, 1, self.scratch_pairs())
            self.run_ops(self.weave(main, lds_, max(1, len(main) // (len(lds_) + 1)) if lds_ else 1))
        else:
            P.s_cmp_eq_u32(S_NBLK, Lit(1))
            P.s_cbranch_scc1("L_nopf")
            self.load_new_rows(1)
            P.label("L_nopf")
            self.stage1(v, 1)
            self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        self.twiddled_exchange(v, 1, LDS_TWF, self.A_X2, 8, self.A_R1, KG * 8)
        P.comment("---- forward pass 3")
        v = self.D
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, 1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        v = self.natural(v)
        P.comment("---- constants")
        self.pointwise(v)
        P.comment("---- inverse pass 1")
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        self.twiddled_exchange(v, -1, LDS_TWM, self.A_R1, KG * 8, self.A_X2, 8)
        P.comment("---- inverse pass 2 + exchange 4")
        v = self.D
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, -1)
        for ka in range(4):
            self.run_ops(bfl[ka])
            for la in range(4):
                P.ds_write_b64(self.A_R1, v[4 * ka + la], 128 * (ka + 4 * la))
        P.s_waitcnt(lgkmcnt=0)
        P.s_barrier()
        order = [4, 8, 12, 0, 5, 9, 13, 1, 6, 10, 14, 2, 7, 11, 15, 3]   # rows 4 al + be with al >= 1 get their twiddle first
        for k in order:
            P.ds_read_b64(self.D[k], self.A_W1, KL1 * 8 * k)
        P.comment("---- inverse pass 3")
        self.stored = False
        if "prio2" in self.opts:
            P.s_setprio(3)
        u = self.inv_pass3()
        P.comment("---- mix and store")
        if core:
            P.s_barrier()   # every wave has read its share of the last exchange: the area takes the output rows
            self.epilogue(u)
            P.s_waitcnt(lgkmcnt=0)
            return
        if not self.stored:
            self.epilogue(u)
        self.sweep_flush()
        if "prio2" in self.opts:
            P.s_setprio(0)
        P.s_add_u32(S_IN.lo, S_IN.lo, Lit(4 * self.BLK))
        P.s_addc_u32(S_IN.hi, S_IN.hi, Lit(0))
        if self.MODE != MODE_SWEEP:    # (SWEEP: S_OUT is the survivor cursor, advanced by what was appended)
            P.s_add_u32(S_OUT.lo, S_OUT.lo, Lit(4 * self.BLK))
            P.s_addc_u32(S_OUT.hi, S_OUT.hi, Lit(0))
        P.s_sub_u32(S_NBLK, S_NBLK, Lit(1))
        P.s_cmp_lg_u32(S_NBLK, Lit(0))
        P.s_cbranch_scc1("L_top")

    def constants(self):
        """SGPR constants (set by the wrapper from literals)"""
        c = {}
        for k in range(1, 16):
            th = math.pi / 2 + math.pi * k / 16
            c[S_C32[k].n] = (math.cos(th), math.sin(th)) if k != 8 else (-1.0, -1.0)
        c[S_RR.n] = (RR, -RR)
        c[S_W1.n] = (C1, -S1)
        c[S_W3.n] = (S1, -C1)
        c[S_W9.n] = (-C1, S1)
        for i in range(1, 4):
            c[S_TWI[i - 1].n] = (math.cos(2 * math.pi * i / 8192), -math.sin(2 * math.pi * i / 8192))
            c[S_TWI[3 + i - 1].n] = (math.cos(2 * math.pi * 2 * i / 4096), -math.sin(2 * math.pi * 2 * i / 4096))
        return c

    def build(self):
        self.prologue()
        self.loop()
        return self.P


# ---------------------------------------------------------------------------------------------------- self-test
def self_test(HS, MODE, NA=1, nblk=3, seed=1, verbose=True, opts=(), kind="run", thr2=0.30):
    import olasave_model as M
    g = Gen(HS, MODE, NA, opts, kind)
    if kind == "core":
        nblk = 1
    P = g.build()
    NT, R0, BLK = g.NT, g.R0, g.BLK
    firlen = 512 * NT
    fir = M.hilbert_fir(firlen)
    gt = fir[1::2]
    B = M.tables(gt)
    rho = -B
    mean = rho[256 * NA:4096 - 256 * NA].mean()
    # device table (prot_engine.hip save_tables)
    tab = np.zeros(2 * 15 * 256, np.float32)
    def put(idx, ang):
        tab[2 * idx], tab[2 * idx + 1] = math.cos(ang), math.sin(ang)
    for t in range(256):
        for i in range(1, 4):
            put((i - 1) * 256 + t, -2 * math.pi * (t - 0.5) * i / 4096)
            put((3 + i - 1) * 256 + t, -2 * math.pi * (t - 0.5) * 4 * i / 4096)
        q = (t >> 4) + 16 * (t & 15)
        for i in range(NA):
            tab[2 * ((13 + i) * 256 + t)] = rho[q + 256 * i] / mean
            tab[2 * ((13 + i) * 256 + t) + 1] = rho[q + 256 * (15 - i)] / mean
    gain = np.float32(mean / 4096)
    rng = np.random.default_rng(seed)
    n_frames = 8192 + BLK * nblk + 64
    x = (rng.random(n_frames, dtype=np.float32) - 0.5)
    # frame index of (row 0, thread 0).q of the first block
    a0 = 2
    out = np.zeros(BLK * nblk + 16, np.float32)
    emu = Emu(lds_bytes=40960)
    IN_BASE, OUT_BASE, SV_BASE = 0x10000000, 0x20000000, 0x30000000
    emu.map(IN_BASE, x.view(np.uint32))
    emu.map(OUT_BASE, out.view(np.uint32))
    emu.map(SV_BASE, tab.view(np.uint32))
    ldsf = emu.lds.view(np.float32)
    for t in range(256):   # table rows 12 (W256 at the half-shifted index) and 14 (plain W256)
        put(12 * 256 + t, -2 * math.pi * ((t & 15) - 0.5) * (t >> 4) / 256)
        put(14 * 256 + t, -2 * math.pi * (((t & 15) * (t >> 4)) & 255) / 256)
    ca, sa = 0.6, -0.8
    if MODE == MODE_SWEEP:   # S_ROT.lo is the bound, the Hilbert branch is unscaled, every wave appends to a list of its own
        ca, sa = thr2, 1.0
        SURV_BASE, CAP = 0x40000000, 2 * BLK * nblk
        surv = np.zeros(4 * CAP, np.float32)
        emu.map(SURV_BASE, surv.view(np.uint32))
    def sset(sp, lo, hi):
        emu.s[:, sp.n] = np.float32(lo).view(np.uint32)
        emu.s[:, sp.n + 1] = np.float32(hi).view(np.uint32)
    for n, (lo, hi) in g.constants().items():
        sset(SP(n), lo, hi)
    sset(S_ROT, ca, np.float32(sa) * gain)
    addr = IN_BASE + 4 * a0
    emu.s[:, S_IN.n], emu.s[:, S_IN.n + 1] = addr & 0xffffffff, addr >> 32
    emu.s[:, S_OUT.n], emu.s[:, S_OUT.n + 1] = OUT_BASE & 0xffffffff, OUT_BASE >> 32
    emu.s[:, S_SV.n], emu.s[:, S_SV.n + 1] = SV_BASE & 0xffffffff, SV_BASE >> 32
    emu.s[:, S_NBLK.n] = nblk
    emu.s[:, S_LDS.n] = 0
    emu.v[g.TMP[0].n] = np.arange(256, dtype=np.uint32)
    if MODE == MODE_SWEEP:
        for w in range(4):
            a_ = SURV_BASE + 4 * CAP * w
            emu.s[w, S_OUT.n], emu.s[w, S_OUT.n + 1] = a_ & 0xffffffff, a_ >> 32
        emu.v[g.V_RAW.n] = np.zeros(256, np.float32).view(np.uint32)
    if kind == "core":   # what the C++ code stages: rows as (q, c) pairs, the gathered frames in lane layout
        tt = np.arange(256)
        for r in range(16):
            a = a0 + 512 * r + 2 * tt
            ldsf[(r * 256 + tt) * 2], ldsf[(r * 256 + tt) * 2 + 1] = x[a], x[a + 1]
        l, w = tt & 63, tt >> 6
        for arr, off, lo in ((LDS_GCAR, HS - NT, 0), (LDS_GNX, 0, NT)):
            n_rows = NT if arr == LDS_GCAR else HS
            ok = (l >= off) & (l < off + n_rows)
            row = np.where(ok, lo + l - off, 0)
            val = np.where(ok, x[a0 + 512 * row + 2 * 64 * w - 1], 0)
            ldsf[arr // 4 + tt] = val
    steps = emu.run(P)
    if kind == "core":
        tt = np.arange(256)
        for k in range(HS):
            out[512 * k + 2 * tt], out[512 * k + 2 * tt + 1] = ldsf[(k * 256 + tt) * 2], ldsf[(k * 256 + tt) * 2 + 1]
    # reference: direct convolution in float64.  z[j] = x[a0 + 2j - 1] + i x[a0 + 2j]; block b's local sample n is
    # stream sample j = b * 256 HS + n; output row R0 + k of block b <-> frames b BLK + 512 k + 2 t (+1)
    xd = x.astype(np.float64)
    nz = (n_frames - a0) // 2 - 1
    z = xd[a0 - 1:a0 - 1 + 2 * nz:2] + 1j * xd[a0:a0 + 2 * nz:2]
    full = np.convolve(z, gt)               # full[m] = sum g[s] z[m - s]
    Lg = len(gt)
    want = np.zeros(BLK * nblk)
    for b in range(nblk):
        for k in range(HS):
            j = b * 256 * HS + 256 * (R0 + k) + np.arange(256)          # stream complex index of the output samples
            y = full[j + Lg // 2]
            dq = xd[a0 + 2 * j]
            dc = xd[a0 + 2 * j + 1]
            pos = b * BLK + 512 * k + 2 * np.arange(256)
            if MODE == MODE_SWEEP:     # keep (d, h) per frame for the survivor check below
                want[pos], want[pos + 1] = y.real, y.imag
                if b == 0 and k == 0:
                    dry = np.zeros(BLK * nblk)
                dry[pos], dry[pos + 1] = dq, dc
            elif MODE == MODE_ROT:
                want[pos] = ca * dq + sa * y.real
                want[pos + 1] = ca * dc + sa * y.imag
            else:
                want[pos] = sa * y.real
                want[pos + 1] = sa * y.imag
    if MODE == MODE_SWEEP:
        # every wave's list: frames in the order (block, row, first / second frame of the pairs, lane); exact d, h to rounding,
        # membership by d^2 + h^2 > bound (frames within rounding of the bound may go either way)
        err = 0.0
        for w in range(4):
            end = (int(emu.s[w, S_OUT.n]) | (int(emu.s[w, S_OUT.n + 1]) << 32)) - (SURV_BASE + 4 * CAP * w)
            got = surv[CAP * w:CAP * w + end // 4].reshape(-1, 2)
            exp = []
            for b in range(nblk):
                for k in range(HS):
                    for half in range(2):
                        pos = b * BLK + 512 * k + 2 * (64 * w + np.arange(64)) + half
                        exp += [(dry[p_], want[p_]) for p_ in pos]
            exp = np.array(exp)
            r2 = exp[:, 0] ** 2 + exp[:, 1] ** 2
            sure, maybe = r2 > thr2 + 1e-5, np.abs(r2 - thr2) <= 1e-5
            gi = 0
            for j in range(len(exp)):
                hit = gi < len(got) and got[gi, 0] == np.float32(exp[j, 0]) and abs(got[gi, 1] - exp[j, 1]) < 1e-6
                if sure[j] or (maybe[j] and hit):
                    assert hit, ("survivor list of wave %d differs at entry %d" % (w, gi))
                    err = max(err, abs(got[gi, 1] - exp[j, 1]))
                    gi += 1
            assert gi == len(got), "wave %d appended %d entries, expected %d" % (w, len(got), gi)
            raw = emu.v[g.V_RAW.n].view(np.float32)[64 * w:64 * w + 64]
            lanes = 64 * w + np.arange(64)
            rexp = np.zeros(64)
            for b in range(nblk):
                for k in range(HS):
                    pos = b * BLK + 512 * k + 2 * lanes
                    rexp = np.maximum(rexp, np.maximum(np.abs(dry[pos]), np.abs(dry[pos + 1])))
            assert np.array_equal(raw, rexp.astype(np.float32)), "running max |x|"
        frac = sum(((int(emu.s[w, S_OUT.n]) | (int(emu.s[w, S_OUT.n + 1]) << 32)) - (SURV_BASE + 4 * CAP * w)) // 8 for w in range(4)) / (BLK * nblk)
        kind = kind + " sweep (%.1f %% of the frames survive)" % (100 * frac)
    else:
        err = np.abs(out[:BLK * nblk] - want).max()
    st = P.stats()
    npk = sum(v for k, v in st.items() if k.startswith("pk_"))
    nval = sum(v for k, v in st.items() if k.startswith("v_"))
    if verbose:
        print("%s HS %2d mode %d NA %d: max err %.2e over %d blocks; VGPRs %d (temporaries %d pairs); loop+prologue: %d packed, %d other VALU, "
              "%d ds, %d vmem, %d waits" % (kind, HS, MODE, NA, err, nblk, g.n_vgpr, len(g.TMP), npk, nval, st.get("ds_read_b64", 0) + st.get("ds_write_b64", 0) + st.get("ds_read_b32", 0),
                                            st.get("gload2", 0) + st.get("gload1", 0) + st.get("gstore2", 0), st.get("s_waitcnt", 0)))
    assert err < 1e-6, err
    return g, P


# (HS = 14, the 512 odd taps of CLI blksiz 1024, needs 124 VGPRs before any temporary: that engine keeps olapoly_kernel)
VARIANTS = [(hs, mode, 1) for hs in (8, 10, 12) for mode in (MODE_ROT, MODE_HILBERT)]


def emit(path, opts=()):
    """csrc/prot_olasave_asm.inc: one macro per variant = the asm text of a run of blocks, its inputs taken from the inline-asm
    operands %0..%7 (in, out: 64-bit SGPR pairs; nblk, lds byte offset: SGPRs; sv: pair; ca, sa * gain: SGPRs; t: VGPR)"""
    out = ["// GENERATED by tools/gen/gen_olasave.py -- do not edit.  The steady-state loop of olasave_kernel as gfx950 assembly",
           "// with hand-allocated registers (see the generator for the register plan and the schedule); the emulator",
           "// (tools/gen/isa.py) has run every variant against the float64 convolution (tests/test_gen_olasave.py).",
           "// operands: %0 in (frame (row 0, thread 0).q of the first block), %1 out (first output frame), %2 blocks, %3 LDS byte",
           "// offset of the exchange area, %4 constants table, %5 ca, %6 sa * gain, %7 thread id",
           "#pragma once", ""]
    for HS, MODE, NA, kind in [(h, m, n, k) for k in ("run", "core") for (h, m, n) in VARIANTS]:
        g = Gen(HS, MODE, NA, opts if kind == "run" else [o for o in opts if o in ("nolds", "novalu")], kind)
        P = g.build()
        head = ["s_mov_b64 %r, %%0" % S_IN, "s_mov_b64 %r, %%1" % S_OUT, "s_mov_b32 %r, %%2" % S_NBLK, "s_mov_b32 %r, %%3" % S_LDS,
                "s_mov_b64 %r, %%4" % S_SV, "s_mov_b32 %r, %%5" % S_ROT.lo, "s_mov_b32 %r, %%6" % S_ROT.hi, "v_mov_b32 %r, %%7" % g.TMP[0].lo,
                "s_waitcnt vmcnt(0) lgkmcnt(0)"]   # (whatever the compiler's code left in flight into registers this block uses)
        for n, (lo, hi) in sorted(g.constants().items()):
            head.append("s_mov_b32 s%d, %r" % (n, Lit(float(lo))))
            head.append("s_mov_b32 s%d, %r" % (n + 1, Lit(float(hi))))
        lines = head + P.text(label_suffix="%=")
        st = P.stats()
        out.append("// %s, HS %d, mode %d, NA %d: %d VGPRs, %d instructions" % (kind, HS, MODE, NA, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label"))))
        out.append("#define PROT_SV_%s_%d_%d_%d \\" % ("ASM" if kind == "run" else "CORE", HS, MODE, NA))
        for l in lines:
            if l.startswith(";"):
                continue
            out.append('\t"%s\\n\t" \\' % l)
        out.append('\t""')
        out.append("")
    # MODE_SWEEP (run kind only; edge blocks take the Hilbert core and are tested by the C++ code): operands %0 survivor cursor
    # (out), %1 running max |x| (out), %2 in, %3 blocks, %4 LDS byte offset, %5 constants table, %6 bound, %7 gain, %8 thread id, %9 and
    # %10 the cursor and the running maximum on entry
    for HS in (8, 12):
        g = Gen(HS, MODE_SWEEP, 1, opts, "run")
        P = g.build()
        head = ["s_mov_b64 %r, %%2" % S_IN, "s_mov_b64 %r, %%9" % S_OUT, "s_mov_b32 %r, %%3" % S_NBLK, "s_mov_b32 %r, %%4" % S_LDS,
                "s_mov_b64 %r, %%5" % S_SV, "s_mov_b32 %r, %%6" % S_ROT.lo, "s_mov_b32 %r, %%7" % S_ROT.hi, "v_mov_b32 %r, %%8" % g.TMP[0].lo,
                "v_mov_b32 %r, %%10" % g.V_RAW, "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        for n, (lo, hi) in sorted(g.constants().items()):
            head.append("s_mov_b32 s%d, %r" % (n, Lit(float(lo))))
            head.append("s_mov_b32 s%d, %r" % (n + 1, Lit(float(hi))))
        tail = ["s_mov_b64 %%0, %r" % S_OUT, "v_mov_b32 %%1, %r" % g.V_RAW]
        lines = head + P.text(label_suffix="%=") + tail
        st = P.stats()
        out.append("// sweep, HS %d: %d VGPRs, %d instructions" % (HS, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label", "probe"))))
        out.append("#define PROT_SV_SWEEP_%d \\" % HS)
        for l in lines:
            if l.startswith(";"):
                continue
            out.append('\t"%s\\n\t" \\' % l)
        out.append('\t""')
        out.append("")
    for HS in (8, 10, 12):
        nv = max(Gen(HS, m, na, opts, k).n_vgpr for (h, m, na) in VARIANTS if h == HS for k in ("run", "core"))
        clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % i for i in SGPR_CLOBBER] + ['"v%d"' % i for i in range(0, nv)]
        out.append("#define PROT_SV_ASM_CLOBBERS_%d " % HS + ", ".join(clob))
        out.append("#define PROT_SV_SWEEP_CLOBBERS_%d PROT_SV_ASM_CLOBBERS_%d, \"v120\"" % (HS, HS))
    out.append("")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "..", "phaserotate.lv2_amd", "csrc")
    if "--variant" in sys.argv:   # timing experiments: python3 gen_olasave.py --variant NAME --opts a,b  -> csrc/variants/
        name = sys.argv[sys.argv.index("--variant") + 1]
        opts = [o for o in sys.argv[sys.argv.index("--opts") + 1].split(",") if o] if "--opts" in sys.argv else []
        wrong = {"nolds", "nomem", "novalu"} & set(opts)
        if not wrong:
            for HS, MODE, NA in VARIANTS:
                self_test(HS, MODE, NA, nblk=2, opts=opts, verbose=False)
        os.makedirs(os.path.join(csrc, "variants"), exist_ok=True)
        dst = os.path.join(csrc, "variants", "prot_olasave_asm_%s.inc" % name)
        emit(dst, opts)
        print("wrote", os.path.normpath(dst), "(results WRONG by design)" if wrong else "(self-test passed)")
        sys.exit(0)
    for HS, MODE, NA in VARIANTS:
        self_test(HS, MODE, NA, nblk=2 if "--quick" in sys.argv else 3)
        self_test(HS, MODE, NA, kind="core")
    for HS in (8, 12):
        self_test(HS, MODE_SWEEP, 1, nblk=2)
    if "--check" not in sys.argv:
        dst = os.path.join(csrc, "prot_olasave_asm.inc")
        emit(dst)
        print("wrote", os.path.normpath(dst))
