#!/usr/bin/env python3
"""Generator of the steady-state loop of olasave8k_kernel (phaserotate.lv2_amd/csrc/prot_olasave8k.hip): the overlap-save,
table-free polyphase block of gen_olasave.py on an 8192-POINT transform pair -- 512 threads (8 waves), 16 points per thread,
two workgroups per CU -- so that 13 of 16 rows of a block are new for the LV2 48 kHz filter (10 of 16 on 4096 points): ~12 %
fewer VALU issue cycles and ~7 % fewer LDS instructions per frame for the same 16 waves per CU.

    python3 tools/gen/gen_olasave8k.py            -> csrc/prot_olasave8k_asm.inc  (+ self-test of every variant in the emulator)
    python3 tools/gen/gen_olasave8k.py --check    -> self-test only

The maths, pass by pass, is tools/model/olasave_model8k.py: thread T keeps positions p = 512 a + T; pass 1 over a; ONE radix-2
layer over the top bit e of T, done by the READER of the first LDS exchange (thread (k1, ke, c) reads both halves of its
column and takes u0 +- u1; the layer's twiddle W32^{b ke} is 15 constant multiplies in the waves with ke = 1 only: ke is
wave-uniform, T >> 8); passes 2 and 3 over b and c as on 4096 points; the inverse mirrors it (the reader of the last exchange
takes Y0 +- Y1).  What crosses waves in LDS is fenced by four workgroup barriers per block (two more than the 4096-point
loop: both halves of a column are read by two waves); the emulator (isa.py) checks that nothing else does.

For an odd number of context rows (NT = 3: LV2 48 kHz) the valid outputs would straddle rows, so the block is kept circularly
rotated by half a row: position p holds local sample p + 256 (mod 8192).  The only consequences are in the loads: row 15 of
the upper four waves is the block's first half row (carried from row HS - 1 of the block before), and the lower waves' row
NT - 1 is carried from row 15 -- one "mixed" load per block whose row depends on the wave, two small wave-uniform branches.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "model"))
from isa import Prog, Emu, V, VP, S, SP, Lit  # noqa: E402
import gen_olasave as G4  # noqa: E402  (the butterfly network, the packed helpers and the SGPR numbers of its constants)
from gen_olasave import Gen, S_IN, S_OUT, S_NBLK, S_LDS, S_SV, S_ROT, S_C32, S_RR, S_W1, S_W3, S_W9, S_T, S_A0, MODE_ROT, MODE_HILBERT, MODE_SWEEP  # noqa: E402

NTHR = 512
ROWS = 560                          # slots (8 B) per k1 row of the exchange area: two 16 x 17 chunks + 16 (rows 4480 B apart: no bank conflict)
HALF = 272                          # slots between the two halves (e / ke = 0, 1) of a row = one chunk
KG = 17
LDS_EX = 16 * ROWS * 8              # bytes of the exchange area
LDS_TWF, LDS_TWI = LDS_EX, LDS_EX + 4096       # the two twiddle tables (512 entries each) behind it
LDS_BYTES = LDS_EX + 8192
LDS_GC = 65536                      # core blocks: the gathered frames behind the 16 staged rows (one float per thread, lane layout)
HI_ROW = 8                          # rows from here on are addressed through the second base register (DS offsets are 16 bits)

# ---- SGPRs beyond gen_olasave's plan (s34..s81 as there) ----
S_TWA, S_TWB = SP(82), SP(84)       # W16384^1, W16384^4: plain twiddle = half-shifted one times these
S_SIG = SP(86)                      # (sigma, sigma), sigma = +1 in waves 0-3, -1 in waves 4-7: the radix-2 layer's sign
S_HI = S(88)                        # 1 in waves 4-7 (e = ke = 1), else 0
S_MIX = S(89)                       # byte offset of the mixed row within a block (odd NT): row 15 (waves 0-3) or row NT - 1 (waves 4-7)
S_A1 = SP(90)                       # second address temporary
SGPR_CLOBBER = range(34, 100)


class Gen8k(Gen):
    def __init__(self, HS, MODE, opts=(), kind="run"):
        self.HS, self.MODE, self.NA = HS, MODE, 1
        self.opts = set(opts)
        self.kind = kind
        self.NT = 16 - HS
        self.R0 = self.NT // 2
        self.ODD = self.NT % 2
        self.ROT = 256 if self.ODD else 0      # complex samples the block is rotated by
        self.BLK = 1024 * HS                   # frames per block
        P = self.P = Prog()
        self.sw_rows, self.sw_label = [], 0
        # "cyclic" (timing experiment, results wrong): the workgroup's next block lies 512 blocks on -- the resident workgroups
        # move through the stream as ONE dense window instead of 512 runs of their own (the context rows are NOT reloaded)
        self.adv = 4 * self.BLK * (512 if "cyclic" in opts else 1)   # bytes from a block to the workgroup's next one
        if "nolds" in self.opts:
            P.ds_write_b64 = lambda *a, **k: None
            P.ds_read_b64 = lambda *a, **k: None
            P.s_barrier = lambda *a, **k: None
        if "novalu" in self.opts:
            P.pk = lambda *a, **k: None
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
        self.TA1, self.TB1 = VP(n), VP(n + 2)          # W_N^{(T - 1/2)}, W_N^{4 (T - 1/2)}; the powers are made when they are used
        n += 4
        self.RHO = [VP(n)]
        n += 2
        self.G_CAR, self.G_NX = V(n), V(n + 1)
        n += 2
        (self.A_RZ, self.A_RZH, self.A_R1, self.A_CH, self.A_X2, self.A_TL, self.VOFF, self.VG) = [V(n + i) for i in range(8)]
        n += 8
        self.TMP = []
        cap = 119
        while n + 1 <= cap:
            self.TMP.append(VP(n))
            n += 2
        self.n_vgpr = n
        assert len(self.TMP) >= 7, "temporaries: %d" % len(self.TMP)
        if MODE == MODE_SWEEP:      # the running max |x| of the frames that pass through (the raw peak) lives across blocks: one
            self.V_RAW = V(120)     # register beyond the plan (gen_olasave.py's sweep_row is inherited as it is)
        self.TB = self.TMP[:4]
        self.TR = self.TMP[4:5]
        self.TT = self.TMP[5:9] if len(self.TMP) >= 9 else self.TMP[5:6] * 4
        self.nbf = len(set(t.n for t in self.TT))
        self.lab = 0

    def tap(self, name, regs):
        """debugging: opts "probe" = snapshot in the emulator; "dump:<name>" = the 16 pairs to global memory (row k of the output
        buffer <- regs[k]) and out: tools/attic/sv8_core_test.py compares the hardware with the emulator stage by stage"""
        P = self.P
        regs = list(regs)
        if "probe" in self.opts:
            P.probe(name, regs)
        if "dump:" + name in self.opts:
            P.s_waitcnt(vmcnt=0, lgkmcnt=0)
            for k, r in enumerate(regs):
                P.s_add_u32(S_A0.lo, S_OUT.lo, Lit(4096 * k))
                P.s_addc_u32(S_A0.hi, S_OUT.hi, Lit(0))
                P.global_store_dwordx2(self.VOFF, r, S_A0, 0)
            P.s_waitcnt(vmcnt=0)
            P.s_branch("L_end")

    def label(self, stem):
        self.lab += 1
        return "%s_%d" % (stem, self.lab)

    STAMP_NAMES = ["(loop top)", "wait: next block's rows (vmcnt)", "(p, q) pairs + forward pass 1", "barrier 1 (last exchange read by all)",
                   "pass-1 twiddles + exchange-1 writes + lgkmcnt(0)", "barrier 2 (exchange 1 written)",
                   "exchange-1 reads, radix-2 fold, W32 twiddle, pass-2 stage 1 + prefetch issue", "barrier 3 (columns read; chunks free)",
                   "pass-2 stage 2 + twiddles + wave-local exchange 2 (drained)", "forward pass 3 + constants",
                   "inverse pass 1 + wave-local exchange 3 (drained)", "inverse pass 2 + W32 twiddle + exchange-4 writes + lgkmcnt(0)",
                   "barrier 4 (exchange 4 written)", "exchange-4 reads, fold, inverse pass 3", "mix + stores issued + loop end"]

    S_STAMP = 92     # (s[92:93] are free in this generator's SGPR plan)

    def tick_lane(self):
        return S_NBLK

    def stamp(self, i):
        """opts "stamps" (diagnostic variant, tools/sv8_stamps.py): shader-clock stamp number i of the block into lane i of v121.  Text
        the emulator does not see (P.raw): s_memtime returns through lgkmcnt, so a stamp drains the LDS queue -- the points are
        chosen where it is empty or about to be waited for anyway; s[92:93] are free in this generator's SGPR plan"""
        if "stamps" not in self.opts or self.kind == "core":
            return
        P = self.P
        P.raw("s_memtime s[%d:%d]" % (self.S_STAMP, self.S_STAMP + 1))
        P.raw("s_waitcnt lgkmcnt(0)")
        P.raw("v_writelane_b32 v121, s%d, %d" % (self.S_STAMP, i))

    def tick(self):
        """opts "tick" (diagnostic variant): ONE 100 MHz stamp per block at the loop's end, into lane S_NBLK (blocks left, this one
        included) of operand %0 -- the progress of every workgroup through its run (tools/sv8_progress.py)"""
        if "tick" not in self.opts or self.kind == "core":
            return
        P = self.P
        a = self.S_STAMP
        P.raw("s_memrealtime s[%d:%d]" % (a, a + 1))
        P.raw("s_waitcnt lgkmcnt(0)")
        P.raw("s_mov_b32 s%d, m0" % (a + 1))   # (one SGPR per VOP3 on gfx9: the lane select goes through M0)
        P.raw("s_mov_b32 m0, %r" % self.tick_lane())
        P.raw("s_nop 1")
        P.raw("v_writelane_b32 %%0, s%d, m0" % a)
        P.raw("s_mov_b32 m0, s%d" % (a + 1))

    def stamps_fold(self):
        """at the loop's end: operand %0 (a VGPR)[i] += stamp[i] - stamp[i - 1] (lanes 1..14; low 32 bits: a block is ~25 K cycles)"""
        if "stamps" not in self.opts or self.kind == "core":
            return
        P = self.P
        t = self.TMP[0].lo     # (free here: the stores read the D registers)
        P.raw("s_nop 1")
        P.raw("v_mov_b32_dpp %r, v121 wave_shr:1 row_mask:0xf bank_mask:0xf" % t)
        P.raw("s_nop 1")
        P.raw("v_sub_u32 %r, v121, %r" % (t, t))
        P.raw("v_writelane_b32 %r, 1, 0" % t)      # (lane 0 counts the blocks)
        P.raw("v_add_u32 %%0, %%0, %r" % t)

    # ------------------------------------------------------------ helpers
    def cmul_c32(self, d, a, k, conj, r):
        """d = a * W32^k (conj: W32^-k), k = 1..15, from the pointwise constants S_C32[k] = (-sin phi, cos phi), phi = pi k / 16
        (W32^k = cos phi - i sin phi: real part in .hi, imaginary part in .lo); r = scratch pair"""
        P = self.P
        if k == 8:      # -i (conj: +i): a swizzle and a sign, from the pair (-1, -1)
            if not conj:
                return [lambda: P.pk("mul", d, [a, S_C32[8]], op_sel=[1, 0], op_sel_hi=[0, 0], neg_lo=[1, 0])]   # (a.hi, -a.lo)
            return [lambda: P.pk("mul", d, [a, S_C32[8]], op_sel=[1, 0], op_sel_hi=[0, 0], neg_hi=[1, 0])]       # (-a.hi, a.lo)
        w = S_C32[k]
        # r = (a.lo wr, a.lo wi); d = (-a.hi wi + r.lo, a.hi wr + r.hi)     [wr = w.hi, wi = w.lo; conj: wi -> -wi]
        if not conj:
            return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 1], op_sel_hi=[0, 0]),
                    lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 0, 0], op_sel_hi=[1, 1, 1], neg_lo=[0, 1, 0])]
        return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 1], op_sel_hi=[0, 0], neg_hi=[0, 1]),
                lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 0, 0], op_sel_hi=[1, 1, 1])]

    def row_slot(self, k1):
        """(base register for reads of a column / writes of thread T, immediate) of row k1 of the exchange area"""
        if k1 < HI_ROW:
            return self.A_RZ, 8 * ROWS * k1
        return self.A_RZH, 8 * ROWS * (k1 - HI_ROW)

    def ke_twiddle(self, v, conj):
        """the radix-2 layer's twiddle: v[b] *= W32^{b} (conj: W32^{-b}) in the waves with ke = 1, skipped by the others"""
        P = self.P
        if "novalu" in self.opts:
            return
        skip = self.label("L_ke")
        sc = self.TMP[:]
        P.settle(list(v) + sc)
        P.s_cmp_eq_u32(S_HI, Lit(0))
        P.s_cbranch_scc1(skip)
        st = P.fork()
        lists = [self.cmul_c32(v[b], v[b], b, conj, sc[(b - 1) % len(sc)]) for b in range(1, 16)]
        self.no_weave = True     # (nothing that every wave must issue inside a region some waves skip)
        for g in range(0, len(lists), len(sc)):
            self.run_ops(self.interleave(lists[g:g + len(sc)]))
        self.no_weave = False
        P.assert_no_waits_since(st)
        P.label(skip)

    # ------------------------------------------------------------ phases
    def prologue(self):
        P, HS, NT = self.P, self.HS, self.NT
        P.comment("---- prologue: wave type, addresses")
        T_ = self.TMP
        t, a, b, c, d, lane, wv, m, z, vg0 = T_[0].lo, T_[0].hi, T_[1].lo, T_[1].hi, T_[2].lo, T_[2].hi, T_[3].lo, T_[3].hi, T_[4].lo, T_[4].hi
        tabs = (T_[5], T_[6])
        P.v_lshr(a, Lit(8), t)                                  # e = ke = T >> 8: wave-uniform
        P.v_and(b, Lit(255), t)
        P.v_lshl(self.VOFF, Lit(3), t)                          # 8 T
        P.v_lshl(b, Lit(3), b)
        P.v_add_u32(self.A_RZ, S_LDS, b)                        # lds + 8 (T & 255): half 0 of a column
        P.v_add_u32(self.A_RZH, Lit(8 * ROWS * HI_ROW), self.A_RZ)
        P.v_readfirstlane(S_HI, a)                              # (well behind the write of its source: isa.py's lane-read hazard)
        P.s_nop(4)
        P.s_cmp_eq_u32(S_HI, Lit(1))
        P.s_cselect(S_SIG.lo, Lit(-1.0), Lit(1.0))
        P.s_cselect(S_SIG.hi, Lit(-1.0), Lit(1.0))
        if self.ODD:
            P.s_mov(S_MIX, Lit(4096 * 15))                         # (one literal per scalar instruction)
            P.s_cselect(S_MIX, Lit(4096 * (NT - 1)), S_MIX)
        # passes 2 and 3: thread (k1, ke, c), k1 = (T >> 4) & 15, c = T & 15
        P.v_lshr(b, Lit(4), t)
        P.v_and(b, Lit(15), b)                                  # k1
        P.v_and(c, Lit(15), t)                                  # c
        P.v_mul_u24(b, Lit(8 * ROWS), b)
        P.v_add_u32(b, S_LDS, b)                                # lds + 8 * 560 k1
        P.v_lshl(d, Lit(3), c)                                  # 8 c
        P.v_add_u32(self.A_R1, b, d)                            # half 0 of row k1, column c
        P.v_mul_u24(m, Lit(8 * HALF), a)                        # 2176 ke
        P.v_add_u32(self.A_CH, self.A_R1, m)                    # the group's own chunk (= its half of the row), + c
        P.v_mul_u24(d, Lit(8 * KG), c)
        P.v_add_u32(self.A_X2, b, d)
        P.v_add_u32(self.A_X2, self.A_X2, m)                    # chunk + 17 c
        P.v_lshl(m, Lit(11), a)                                 # 2048 ke
        P.v_lshl(d, Lit(3), c)
        P.v_add_u32(m, m, d)
        P.v_add_u32(self.A_TL, S_LDS, m)
        P.v_add_u32(self.A_TL, Lit(LDS_TWF), self.A_TL)         # lds + tables + 8 (256 ke + c)   (+ 0 / 4096 + 128 k2 as immediates)
        # the half row the block is rotated by rides in the input pointer
        if self.ROT:
            P.s_add_u32(S_IN.lo, S_IN.lo, Lit(8 * self.ROT))
            P.s_addc_u32(S_IN.hi, S_IN.hi, Lit(0))
        # gather offsets: lane l of a wave fetches the frame in front of this wave's first sample of the row N[l] holds
        P.v_and(lane, Lit(63), t)
        P.v_lshr(wv, Lit(6), t)
        P.v_lshl(wv, Lit(9), wv)                                # 512 B per wave
        if not self.ODD:
            P.v_add_u32(a, Lit(NT), lane)                       # N[l] = row NT + l
        else:
            P.v_add_u32(a, Lit(NT - 1), lane)                   # N[l] = row NT + l - 1 (l >= 1) ...
        P.v_lshl(a, Lit(12), a)
        if self.ODD:
            P.v_mov(b, S_MIX)
            P.v_cmp_eq_u32(Lit(0), lane)
            P.v_cndmask(a, a, b)                                # ... N[0] = the mixed row
        P.v_cmp_lt_u32(lane, Lit(HS)) if False else None
        P.v_mov(b, Lit(HS))
        P.v_cmp_lt_u32(lane, b)                                 # lane < HS
        P.v_mov(b, Lit(0))
        P.v_cndmask(a, b, a)                                    # lanes without a row: offset 0 (a valid address)
        P.v_add_u32(self.VG, a, wv)
        if self.kind != "core":
            self.first_gather_offsets(vg0, lane, wv, [a, b, c])
        P.comment("---- prologue: twiddle registers, |B| factors, the two tables of the middle passes")
        for row, dst in ((0, self.TA1), (1, self.TB1), (4, self.RHO[0]), (2, tabs[0]), (3, tabs[1])):
            P.s_add_u32(S_A0.lo, S_SV.lo, Lit(4096 * row))
            P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
            P.global_load_dwordx2(dst, self.VOFF, S_A0, 0)
        return tabs, t, vg0

    def first_gather_offsets(self, vg0, lane, wv, tmp):
        """vg0[lane] = byte offset (within the first block) of the row whose front frame that lane of G_CAR must hold for the
        first build_pq (the lanes the steady state finds them in), + 512 wave; lanes without a carried row: 0"""
        P, HS, NT = self.P, self.HS, self.NT
        x, y, z = tmp
        lo = (HS - NT) if not self.ODD else (HS - NT + 1)       # carried row k (k = 0 .. n_common - 1) <-> lane lo + k
        n_common = NT if not self.ODD else NT - 1
        # (offsets are relative to one row BELOW the block: the VGPR offset of a global access is unsigned)
        P.v_add_u32(x, Lit(1 - lo), lane)                       # row candidate + 1
        P.v_lshl(x, Lit(12), x)
        P.v_mov(z, Lit(4096))
        P.v_mov(y, Lit(lo))
        P.v_cmp_lt_u32(lane, y)
        P.v_cndmask(x, x, z)                                    # lanes below: 0
        P.v_mov(y, Lit(lo + n_common - 1))
        P.v_cmp_lt_u32(y, lane)
        P.v_cndmask(x, x, z)                                    # lanes above: 0
        if self.ODD:
            # lower waves: lane 0 <-> row NT - 1; upper waves: lane HS - NT <-> row 15, wrapped (the block's first half row)
            P.s_cmp_eq_u32(S_HI, Lit(1))
            P.s_cselect(S_T[0], Lit(HS - NT), Lit(0))
            P.s_mov(S_T[1], Lit(4096 * (NT - 1) + 4096))
            P.s_cselect(S_T[1], Lit(4096 * 15 - 65536 + 4096), S_T[1])
            P.v_mov(y, S_T[1])
            P.v_cmp_eq_u32(S_T[0], lane)
            P.v_cndmask(x, x, y)
        P.v_add_u32(vg0, x, wv)

    def load_row(self, dst, row_bytes, soff=None, sreg=None):
        """one row of a block: dst <- [S_IN + soff + sreg + row_bytes + VOFF] (soff, sreg: SGPRs or None; row_bytes may be negative)"""
        P = self.P
        nt = "nt" in self.opts
        src = S_IN
        for extra in (soff, sreg):
            if extra is not None:
                P.s_add_u32(S_A0.lo, src.lo, extra)
                P.s_addc_u32(S_A0.hi, src.hi, Lit(0))
                src = S_A0
        if row_bytes < 0:
            P.s_sub_u32(S_A0.lo, src.lo, Lit(-row_bytes))
            P.s_subb_u32(S_A0.hi, src.hi, Lit(0))
        else:
            P.s_add_u32(S_A0.lo, src.lo, Lit(row_bytes))
            P.s_addc_u32(S_A0.hi, src.hi, Lit(0))
        P.global_load_dwordx2(dst, self.VOFF, S_A0, 0, nt=nt)

    def new_row_bytes(self, i):
        """(row byte offset, needs S_MIX) of the row N[i] takes"""
        if not self.ODD:
            return 4096 * (self.NT + i), False
        if i == 0:
            return 0, True
        return 4096 * (self.NT + i - 1), False

    def load_new_rows(self, blk_ahead, thunks=False, soff=None):
        P, HS = self.P, self.HS
        if "nomem" in self.opts and blk_ahead:
            return []
        assert soff is not None or blk_ahead == 0
        ops = []
        for i in range(HS):
            rb, mix = self.new_row_bytes(i)
            ops.append(lambda i=i, rb=rb, mix=mix: self.load_row(self.N[i], rb, soff=soff, sreg=S_MIX if mix else None))
        def g():
            if soff is not None:
                P.s_add_u32(S_A0.lo, S_IN.lo, soff)
                P.s_addc_u32(S_A0.hi, S_IN.hi, Lit(0))
                P.global_load_dword(self.G_NX, self.VG, S_A0, -4)
            else:
                P.global_load_dword(self.G_NX, self.VG, S_IN, -4)
        ops.append(g)
        if thunks:
            return ops
        for f in ops:
            f()
        return []

    def prologue_rows(self, tabs, t, vg0):
        """the first block's rows into the registers the steady state finds them in; the tables into LDS"""
        P, HS, NT = self.P, self.HS, self.NT
        core = self.kind == "core"
        if not core:
            P.comment("---- prologue: the first block's rows (requested early)")
            for k in range(NT if not self.ODD else NT - 1):
                self.load_row(self.K[HS + k], 4096 * k)
            if self.ODD:
                up, done = self.label("L_pu"), self.label("L_pd")
                P.s_cmp_eq_u32(S_HI, Lit(1))
                P.s_cbranch_scc1(up)
                st0 = P.fork()
                self.load_row(self.K[15], 4096 * (NT - 1))              # lower waves: row NT - 1, where a carried row 15 would be
                st1 = P.fork()
                P.s_branch(done)
                P.label(up)
                P.restore(st0)
                self.load_row(self.K[HS - 1], 4096 * 15 - 65536)        # upper waves: row 15 = the block's first half row (wrapped)
                P.label(done)
                P.join(st1)
            P.s_sub_u32(S_A0.lo, S_IN.lo, Lit(4096))
            P.s_subb_u32(S_A0.hi, S_IN.hi, Lit(0))
            P.global_load_dword(self.G_CAR, vg0, S_A0, -4)
            self.load_new_rows(0)
        # table entries to LDS: slot T of each (the layout [ke][k2][c] is the host's); the inverse one carries sa * gain
        # (S_ROT.hi): every value of the block passes through exactly one of its entries
        w = self.TMP[1].lo
        wt = self.TMP[1].hi
        P.v_add_u32(w, S_LDS, self.VOFF)
        P.v_add_u32(wt, Lit(LDS_TWF), w)
        P.pk("mul", tabs[1], [tabs[1], S_ROT], op_sel=[0, 1], op_sel_hi=[1, 1])
        P.ds_write_b64(wt, tabs[0], 0)
        P.ds_write_b64(wt, tabs[1], LDS_TWI - LDS_TWF)
        if core:
            P.comment("---- prologue: the staged block")
            self.SRC = {}
            for r in range(16):
                self.SRC[r] = self.K[r] if r >= self.R0 else self.N[r]
                P.ds_read_b64(self.SRC[r], w, 4096 * r)
            a4 = self.TMP[2].lo
            P.v_lshl(a4, Lit(2), t)
            P.v_add_u32(a4, S_LDS, a4)
            P.v_add_u32(a4, Lit(LDS_GC), a4)
            P.ds_read_b32(self.G_CAR, a4, 0)
            P.s_waitcnt(vmcnt=0, lgkmcnt=0)
            P.s_barrier()   # every wave has its rows: the exchange area may be written
        P.comment("---- prologue done")

    # ---- (p, q) pairs
    def pq_row(self, k, src, gv, lane):
        """three thunk lists (phases) for D[k] = (p, q) of the row kept in `src` (aligned (q, c) pairs), lane 0's p from lane `lane` of gv"""
        P = self.P
        a = [lambda: P.v_mov(self.D[k].lo, gv) if lane == 0 else P.v_mov_dpp_rowshl(self.D[k].lo, gv, lane)]
        b = [lambda: P.v_mov(self.D[k].hi, src.lo)]
        c = [lambda: P.v_mov_dpp_shr1(self.D[k].lo, src.hi)]
        return a, b, c

    def build_rows(self, plan):
        """plan: list of (k, src, gv, lane); in groups of four rows so that the DPP moves keep their two wait states"""
        for g in range(0, len(plan), 4):
            ph = [self.pq_row(*x) for x in plan[g:g + 4]]
            for phase in range(3):
                for r in ph:
                    self.run_ops(r[phase])
            if len(plan[g:g + 4]) < 2:
                self.P.s_nop(1)

    def build_pq(self):
        P, HS, NT, R0 = self.P, self.HS, self.NT, self.R0
        if self.kind == "core":
            self.build_rows([(k, self.SRC[k], self.G_CAR, k) for k in range(16)])
            return
        if not self.ODD:
            src = lambda k: self.K[HS + k] if k < NT else self.N[k - NT]
            plan = [(k, src(k), self.G_CAR, HS - NT + k) if k < NT else (k, src(k), self.G_NX, k - NT) for k in range(16)]
            self.build_rows(plan)
            for r in range(R0, NT):
                P.v_mov(self.K[r].lo, self.K[HS + r].lo)
                P.v_mov(self.K[r].hi, self.K[HS + r].hi)
            for i in range(HS):
                P.v_mov(self.K[NT + i].lo, self.N[i].lo)
                P.v_mov(self.K[NT + i].hi, self.N[i].hi)
            P.v_mov(self.G_CAR, self.G_NX)
            return
        # odd NT: rows NT - 1 and 15 depend on the wave (see the file header); everything else is common
        common = []
        for k in range(16):
            if k <= NT - 2:
                common.append((k, self.K[HS + k], self.G_CAR, HS + k - NT + 1))
            elif NT <= k <= 14:
                common.append((k, self.N[k - NT + 1], self.G_NX, k - NT + 1))
        self.build_rows(common)
        up, done = self.label("L_bu"), self.label("L_bd")
        P.settle([self.K[15], self.K[HS - 1], self.N[0], self.G_CAR, self.G_NX, self.D[NT - 1], self.D[15]] + ([self.K[NT - 1]] if NT - 1 >= R0 else []))
        stb = P.fork()
        P.s_cmp_eq_u32(S_HI, Lit(1))
        P.s_cbranch_scc1(up)
        # lower waves: row NT - 1 <- carried row 15, row 15 <- the mixed load
        self.build_rows([(NT - 1, self.K[15], self.G_CAR, 0), (15, self.N[0], self.G_NX, 0)])
        if NT - 1 >= R0:
            P.v_mov(self.K[NT - 1].lo, self.K[15].lo)
            P.v_mov(self.K[NT - 1].hi, self.K[15].hi)
        P.v_mov(self.K[15].lo, self.N[0].lo)
        P.v_mov(self.K[15].hi, self.N[0].hi)
        P.s_branch(done)
        P.label(up)
        # upper waves: row NT - 1 <- the mixed load, row 15 <- carried row HS - 1
        self.build_rows([(NT - 1, self.N[0], self.G_NX, 0), (15, self.K[HS - 1], self.G_CAR, HS - NT)])
        P.v_mov(self.K[15].lo, self.K[HS - 1].lo)
        P.v_mov(self.K[15].hi, self.K[HS - 1].hi)
        if NT - 1 >= R0:
            P.v_mov(self.K[NT - 1].lo, self.N[0].lo)
            P.v_mov(self.K[NT - 1].hi, self.N[0].hi)
        P.label(done)
        P.assert_no_waits_since(stb)
        for r in range(R0, NT - 1):
            P.v_mov(self.K[r].lo, self.K[HS + r].lo)
            P.v_mov(self.K[r].hi, self.K[HS + r].hi)
        for r in range(NT, 15):
            P.v_mov(self.K[r].lo, self.N[r - NT + 1].lo)
            P.v_mov(self.K[r].hi, self.N[r - NT + 1].hi)
        P.v_mov(self.G_CAR, self.G_NX)

    # ---- forward pass 1: rows -> k1, outer twiddle, first exchange
    def fwd_pass1(self):
        P = self.P
        v = self.D
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        # W^{(T - 1/2) ka}: A2 = A1^2, A3 = A1 A2 made here (two scratch pairs hold them while they are used)
        A = [None, self.TA1, self.TB[2], self.TB[3]]
        sc = [self.TR[0], self.TB[0], self.TB[1]]
        self.cmul(A[2], self.TA1, self.TA1, 1, sc[0])
        self.cmul(A[3], A[2], self.TA1, 1, sc[1])
        lists = []
        for ka in range(1, 4):
            for be in range(4):
                lists.append(self.cmul_ops(v[4 * ka + be], v[4 * ka + be], A[ka], 1, sc[len(lists) % len(sc)]))
        for g in range(0, len(lists), len(sc)):
            self.run_ops(self.interleave(lists[g:g + len(sc)]))
        bfl = self.stage2_lists(v, 1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        B = [None, self.TB1, self.TB[2], self.TB[3]]
        self.cmul(B[2], self.TB1, self.TB1, 1, sc[0])
        self.cmul(B[3], B[2], self.TB1, 1, sc[1])
        # the previous block's last exchange has been read by both waves that share a column: its slots may be written
        self.stamp(2)
        if "nob1" not in self.opts:     # (timing experiment: results may be wrong)
            P.s_barrier()
        self.stamp(3)
        up, done = self.label("L_wu"), self.label("L_wd")
        P.settle(list(v) + sc + [B[2], B[3]])
        P.s_cmp_eq_u32(S_HI, Lit(1))
        P.s_cbranch_scc1(up)
        st0 = P.fork()
        st1 = None
        for half, lab in ((0, None), (1, up)):
            if lab:
                P.label(lab)
                P.restore(st0)
            n = 0
            for ka in range(4):
                base, imm = self.row_slot(ka)
                P.ds_write_b64(base, v[4 * ka], imm + 8 * HALF * half)
            for la in range(1, 4):
                for ka in range(4):
                    x = v[4 * ka + la]
                    self.cmul(x, x, B[la], 1, sc[n % len(sc)])   # (common to both kinds of wave; each runs one of the two copies)
                    n += 1
                    base, imm = self.row_slot(ka + 4 * la)
                    P.ds_write_b64(base, x, imm + 8 * HALF * half)
            if half == 0:
                st1 = P.fork()
                P.s_branch(done)
        P.label(done)
        P.join(st1)
        if "probe" in self.opts or "dump:p1out" in self.opts:
            # (debugging) what pass 1 handed to the exchange, and the wave-uniform state as the registers see it
            P.v_mov(self.TMP[0].lo, S_SIG.lo)
            P.v_mov(self.TMP[0].hi, S_HI)
            self.tap("p1out", self.natural(v)[:15] + [self.TMP[0]])

    def fwd_pass2_reads(self):
        """the reader's half of the radix-2 layer: D[b] = u0[b] + sigma u1[b], then W32^{b ke}"""
        P = self.P
        v = self.D
        order = [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
        sets = [self.TMP[:4], self.TMP[4:8] if len(self.TMP) >= 8 else []]
        nset = len(sets[1])
        def issue(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                P.ds_read_b64(v[k], self.A_R1, 128 * k)
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                P.ds_read_b64(sets[j % 2][i], self.A_R1, 128 * k + 8 * HALF)
        def fold(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                P.pk("fma", v[k], [sets[j % 2][i], S_SIG, v[k]])
        if nset < 4:   # not enough temporaries for two batches in flight
            for j in range(4):
                sets[1] = sets[0]
                issue(j)
                fold(j)
        else:
            issue(0)
            for j in range(1, 4):
                issue(j)
                fold(j - 1)
            fold(3)
        self.ke_twiddle(v, False)
        return v

    def twiddled_exchange8(self, v, DIR, tw_off, waddr, wstride, raddr, rstride):
        """second half of a middle pass: stage 2 one DFT-4 at a time, every output times its table entry (k = 0 too: the radix-2
        layer's c-dependent factor, and the block's gain in the inverse table), the wave-local exchange, the transposed read"""
        P = self.P
        def fetch(ka):
            for la in range(4):
                P.ds_read_b64(self.TB[la], self.A_TL, tw_off + 128 * (ka + 4 * la))
        bfl = [self.stage2_ops(v, ka, DIR, self.TT[0]) for ka in range(4)]
        r = self.TR[0]
        if DIR > 0:
            self.stamp(6)
        fetch(0)
        if DIR > 0:
            # every wave has read both halves of its columns of the first exchange: the chunks (which lie in them) may be written
            if "nob3" not in self.opts:     # (timing experiment: results may be wrong)
                P.s_barrier()
            self.stamp(7)
        for ka in range(4):
            self.run_ops(bfl[ka])
            for la in range(4):
                k = ka + 4 * la
                x = v[4 * ka + la]
                self.cmul(x, x, self.TB[la], DIR, r)
                P.ds_write_b64(waddr, x, wstride * k)
            if ka + 1 < 4:
                fetch(ka + 1)
        order = [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
        for k in order:
            P.ds_read_b64(self.D[k], raddr, rstride * k)

    def inv_pass3(self):
        """the last inverse pass: plain outer twiddles derived from the half-shifted ones (and their powers) one at a time"""
        P = self.P
        v = self.D
        w, w1, r = self.TB[0], self.TB[1], self.TR[0]
        sc = [x for x in self.TMP if x.n not in (w.n, w1.n, r.n)][:4]
        uniq = len(sc) >= 4
        def power(al, base, cst):
            if al == 1:
                self.cmul(w1, base, cst, 1, r)
                return w1
            if al == 2:
                self.cmul(w, w1, w1, 1, r)
            else:
                self.cmul(w, w, w1, 1, r)
            return w
        for al in range(1, 4):      # inputs n = 4 al + be: conj(W^{4 T al})
            ww = power(al, self.TB1, S_TWB)
            ls = [self.cmul_ops(v[4 * al + be], v[4 * al + be], ww, -1, sc[be % len(sc)]) for be in range(4)]
            self.run_ops(self.interleave(ls) if uniq else sum(ls, []))
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, [self.TR[0]] + self.TB[:]))
        for be in range(1, 4):      # y[be][ka] sits in v[4 ka + be]: conj(W^{T be})
            ww = power(be, self.TA1, S_TWA)
            ls = [self.cmul_ops(v[4 * ka + be], v[4 * ka + be], ww, -1, sc[ka % len(sc)]) for ka in range(4)]
            self.run_ops(self.interleave(ls) if uniq else sum(ls, []))
        rows = range(self.R0, self.R0 + self.HS)
        bfl = self.stage2_lists(v, -1, need=[[la for la in range(4) if ka + 4 * la in rows] for ka in range(4)])
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        return self.natural(v)

    def epilogue_row(self, u, k):
        P, R0 = self.P, self.R0
        nt = "nt" in self.opts
        x = u[R0 + k]
        if self.MODE == MODE_SWEEP: # (run kind only: edge blocks take the Hilbert core and are tested by the C++ code)
            self.sweep_row(self.K[R0 + k], x)
            return
        if self.MODE == MODE_ROT:   # (sa * gain rides in the inverse table: prologue_rows)
            P.pk("fma", x, [self.K[R0 + k], S_ROT, x], op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1])
        if self.kind == "core":
            P.ds_write_b64(self.A_OUT, x, 4096 * k)
        elif "nomem" not in self.opts:
            P.s_add_u32(S_A0.lo, S_OUT.lo, Lit(4096 * k))
            P.s_addc_u32(S_A0.hi, S_OUT.hi, Lit(0))
            P.global_store_dwordx2(self.VOFF, x, S_A0, 0, nt=nt)

    def loop(self):
        P, HS = self.P, self.HS
        core = self.kind == "core"
        if not core:
            P.s_waitcnt(vmcnt=0)
            P.label("L_top")
            self.stamp(0)
            n_st = 0 if self.MODE == MODE_SWEEP else HS   # (SWEEP: sweep_flush has waited for the loads; appends are not counted)
            if self.MODE == MODE_SWEEP:
                P.s_waitcnt(lgkmcnt=0)
            else:
                P.s_waitcnt(vmcnt=n_st, lgkmcnt=0)
            P.reset_queues(vm_stores=n_st)
            self.stamp(1)
        P.comment("---- (p, q) pairs")
        self.build_pq()
        self.tap("pq", self.D)
        P.comment("---- forward pass 1 + exchange 1")
        self.fwd_pass1()
        P.s_waitcnt(lgkmcnt=0)
        self.stamp(4)
        P.s_barrier()
        self.stamp(5)
        P.comment("---- forward pass 2 (the radix-2 layer first)")
        v = self.fwd_pass2_reads()
        self.tap("p2in", self.D)
        if core:
            self.stage1(v, 1)
            self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        else:
            P.s_cmp_eq_u32(S_NBLK, Lit(1))
            P.s_cselect(S_T[3], Lit(0), Lit(self.adv))
            lds_ = self.load_new_rows(1, thunks=True, soff=S_T[3])
            main = self.stage1_ops(v, 1) + self.stage1_twiddle_ops(v, 1, self.scratch_pairs())
            self.run_ops(self.weave(main, lds_, max(1, len(main) // (len(lds_) + 1)) if lds_ else 1))
        self.twiddled_exchange8(v, 1, 0, self.A_X2, 8, self.A_CH, KG * 8)
        self.stamp(8)
        P.comment("---- forward pass 3")
        v = self.D
        self.tap("p3in", self.D)
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, 1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        v = self.natural(v)
        self.tap("p3out", v)
        P.comment("---- constants")
        self.pointwise(v)
        self.tap("pw", v)
        self.stamp(9)
        P.comment("---- inverse pass 1")
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        self.twiddled_exchange8(v, -1, LDS_TWI - LDS_TWF, self.A_CH, KG * 8, self.A_X2, 8)
        self.stamp(10)
        P.comment("---- inverse pass 2, the radix-2 layer's twiddle, exchange 4")
        v = self.D
        self.tap("ip2in", self.D)
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, -1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        u = self.natural(v)          # u[b] = output b of the pass
        self.tap("ip2out", u)
        self.ke_twiddle(u, True)
        self.tap("ip2tw", u)
        for b in range(16):
            P.ds_write_b64(self.A_CH, u[b], 128 * b)   # into the group's own chunk: wave-local, no barrier in front
        P.s_waitcnt(lgkmcnt=0)
        self.stamp(11)
        P.s_barrier()
        self.stamp(12)
        # the reader's half of the inverse layer: thread T takes Y0 + sigma Y1 of its column (rows 4 al + be with al >= 1 first)
        order = [4, 8, 12, 0, 5, 9, 13, 1, 6, 10, 14, 2, 7, 11, 15, 3]
        sets = [self.TMP[:4], self.TMP[4:8] if len(self.TMP) >= 8 else []]
        def issue(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                base, imm = self.row_slot(k)
                P.ds_read_b64(self.D[k], base, imm)
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                base, imm = self.row_slot(k)
                P.ds_read_b64(sets[j % 2][i], base, imm + 8 * HALF)
        def fold(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                P.pk("fma", self.D[k], [sets[j % 2][i], S_SIG, self.D[k]])
        if len(sets[1]) < 4:
            for j in range(4):
                sets[1] = sets[0]
                issue(j)
                fold(j)
        else:
            issue(0)
            for j in range(1, 4):
                issue(j)
                fold(j - 1)
            fold(3)
        self.tap("ip3in", self.D)
        P.comment("---- inverse pass 3")
        u = self.inv_pass3()
        self.tap("ip3out", u)
        self.stamp(13)
        P.comment("---- mix and store")
        if core:
            P.s_barrier()   # every wave has read its share of the last exchange: the area takes the output rows
            self.A_OUT = self.TMP[0].lo
            P.v_add_u32(self.A_OUT, S_LDS, self.VOFF)
            for k in range(HS):
                self.epilogue_row(u, k)
            P.s_waitcnt(lgkmcnt=0)
            return
        for k in range(HS):
            self.epilogue_row(u, k)
        self.sweep_flush()
        self.stamp(14)
        self.stamps_fold()
        self.tick()
        P.s_add_u32(S_IN.lo, S_IN.lo, Lit(self.adv))
        P.s_addc_u32(S_IN.hi, S_IN.hi, Lit(0))
        if self.MODE != MODE_SWEEP:    # (SWEEP: S_OUT is the survivor cursor, advanced by what was appended)
            P.s_add_u32(S_OUT.lo, S_OUT.lo, Lit(self.adv))
            P.s_addc_u32(S_OUT.hi, S_OUT.hi, Lit(0))
        P.s_sub_u32(S_NBLK, S_NBLK, Lit(1))
        P.s_cmp_lg_u32(S_NBLK, Lit(0))
        P.s_cbranch_scc1("L_top")

    def constants(self):
        c = {}
        for k in range(1, 16):
            th = math.pi / 2 + math.pi * k / 16
            c[S_C32[k].n] = (math.cos(th), math.sin(th)) if k != 8 else (-1.0, -1.0)
        c[S_RR.n] = (G4.RR, -G4.RR)
        c[S_W1.n] = (G4.C1, -G4.S1)
        c[S_W3.n] = (G4.S1, -G4.C1)
        c[S_W9.n] = (-G4.C1, G4.S1)
        c[S_TWA.n] = (math.cos(2 * math.pi / 16384), -math.sin(2 * math.pi / 16384))
        c[S_TWB.n] = (math.cos(2 * math.pi * 4 / 16384), -math.sin(2 * math.pi * 4 / 16384))
        return c

    def build(self):
        tabs, t, vg0 = self.prologue()
        self.prologue_rows(tabs, t, vg0)
        self.loop()
        if any(o.startswith("dump:") for o in self.opts):
            self.P.label("L_end")
        return self.P


# ---------------------------------------------------------------------------------------------------- dynamic blocks (round 5)
# SGPRs of the dynamic loop (the plan above minus S_NBLK, S_MIX, S_A1; S_T[0..2] are persistent here, S_T[3] stays a temporary)
S_BLK = S(38)                       # the block being transformed (index among the launch's whole in-range blocks of this channel)
S_NEXT = S(89)                      # the block after it, or 0xffffffff: none left
S_CTRX = SP(90)                     # wave 0: address of the counter of the region it draws from
S_TK = S(92)                        # wave 0: the ticket in flight / arrived
S_XR = S(93)                        # wave 0: the region it draws from (0..7: one per XCD, its own first)
S_TRIES, S_PER, S_NB = S(94), S(95), S(96)   # wave 0: regions left to try; blocks per region; blocks in all
S_ADV = S_SV                        # byte offset from the current block to the next one (the table pointer is dead behind the prologue)
NONE = 0xffffffff
SPREAD = 16                         # arithmetic instructions between two loads of the prefetch (option "spread<n>"; 0: all inside one stage)
LDS_MAIL = 8 * 544                  # byte offset of the mailbox in the exchange area: the 16 spare slots at the end of row 0


class Gen8kDyn(Gen8k):
    """The same block with the workgroup's NEXT block drawn from a counter instead of lying behind the current one (round 5).

    tools/sv8_progress.py on the static run plan: while both workgroups of a CU are resident the one dispatched first takes 7.5 us
    per block and the other 17.4 (the hardware serves the oldest wave first), the first finishes its run at 198 us and the second
    works on alone -- at 0.144 blocks/us per CU instead of 0.191 -- until 252: the launch averages 40 blocks/us against 48.5 in its
    steady state, whatever the run lengths (1.0 : 1.0 ... 1.55 : 0.45 all measure the same).  Here every block is claimed when it is
    needed: the stream's whole in-range blocks are dealt out in eight contiguous regions, one per XCD (a block's context rows are
    the rows its predecessor's workgroup has just read: they hit in that XCD's L2), each behind a counter of its own; wave 0 of a
    workgroup draws a ticket with a scalar atomic (s_atomic_add ... glc: wave-uniform, through the scalar cache, lgkmcnt) a block
    ahead -- issued behind the last LDS read of block B, collected in front of the first LDS write of block B + 1, published through
    a mailbox word in LDS in front of that block's second barrier and read by all eight waves behind it, in time for the prefetch of
    block B + 2's rows -- and moves on to the next region when its own is exhausted.  All 16 rows of a block are loaded (nothing is
    carried from block to block, so the odd-context "mixed row" is just an address), the dry rows are copied aside as before."""

    def __init__(self, HS, MODE, opts=(), kind="run"):
        assert kind == "run" and MODE != MODE_SWEEP
        self.HS, self.MODE, self.NA = HS, MODE, 1
        self.opts = set(opts)
        self.kind = kind
        self.NT = 16 - HS
        self.R0 = self.NT // 2
        self.ODD = self.NT % 2
        self.ROT = 256 if self.ODD else 0
        self.BLK = 1024 * HS
        P = self.P = Prog()
        self.sw_rows, self.sw_label = [], 0
        if "nolds" in self.opts:
            P.ds_write_b64 = lambda *a, **k: None
            P.ds_read_b64 = lambda *a, **k: None
            P.s_barrier = lambda *a, **k: None
        if "novalu" in self.opts:
            P.pk = lambda *a, **k: None
        self.D = [VP(2 * k) for k in range(16)]
        n = 32
        self.K = {}
        for r in range(self.R0, self.R0 + HS):       # the dry rows = the valid output rows
            self.K[r] = VP(n)
            n += 2
        self.N = []
        for i in range(16):                          # the next block's rows, all of them
            self.N.append(VP(n))
            n += 2
        self.TA1, self.TB1 = VP(n), VP(n + 2)
        n += 4
        self.RHO = [VP(n)]
        n += 2
        self.G_NX, self.VT = V(n), V(n + 1)          # gathered frames (lanes 0..15); mailbox value
        n += 2
        (self.A_RZ, self.A_RZH, self.A_R1, self.A_CH, self.A_X2, self.A_TL, self.VOFF, self.VG) = [V(n + i) for i in range(8)]
        n += 8
        self.TMP = []
        cap = 119
        while n + 1 <= cap:
            self.TMP.append(VP(n))
            n += 2
        self.n_vgpr = n
        assert len(self.TMP) >= 7, "temporaries: %d" % len(self.TMP)
        self.TB = self.TMP[:4]
        self.TR = self.TMP[4:5]
        self.TT = self.TMP[5:9] if len(self.TMP) >= 9 else self.TMP[5:6] * 4
        self.nbf = len(set(t.n for t in self.TT))
        self.lab = 0
        self.n_tick = 0

    S_STAMP = 40     # (S_SV is dead behind the prologue; S_ADV below shares it and is not live across any stamp)

    def tick_lane(self):
        return S_TRIES   # (diagnostic "tick" variant only: see tick())

    # ------------------------------------------------------------ wave 0's side of the claims
    def if_wave0(self, body):
        """body() in wave 0 only (VOFF = 8 T: lane 0 of wave w holds 512 w); no wait may be inserted inside"""
        P = self.P
        skip = self.label("L_w0")
        P.v_readfirstlane(S_T[3], self.VOFF)
        P.s_nop(4)
        P.s_cmp_eq_u32(S_T[3], Lit(0))
        P.s_cbranch_scc0(skip)
        st = P.fork()
        body()
        P.assert_no_waits_since(st)
        P.restore(st)     # (what wave 0 alone has put in flight is not in the other waves' queues: the bookkeeping forgets it -- wave
        P.label(skip)     # 0's own waits become stricter by that much, nobody's laxer)

    def claim_issue(self):
        """wave 0: draw the next ticket of the current region (if the block after this one exists: there is nothing to draw behind
        the end).  Nothing waits for it here."""
        P = self.P
        def body():
            none = self.label("L_ci")
            P.s_cmp_eq_u32(S_NEXT, Lit(NONE))
            P.s_cbranch_scc1(none)
            P.s_mov(S_TK, Lit(1))
            P.s_atomic_add(S_TK, S_CTRX)
            P.label(none)
        self.if_wave0(body)

    def claim_collect_publish(self):
        """wave 0: the ticket has had a block's time to arrive; turn it into a block (or move on to the next region, or find that
        nothing is left) and put the answer into the mailbox.  The LDS queue is empty here (the explicit wait is for the scalar
        atomic alone) and the workgroup barrier that follows the first exchange's writes orders the mailbox for the readers."""
        P = self.P
        def body():
            retry, valid, none, pub = [self.label(x) for x in ("L_cr", "L_cv", "L_cn", "L_cp")]
            P.s_wait_lgkm0_here()
            P.label(retry)
            P.s_mul_i32(S_T[3], S_XR, S_PER)                    # first block of the region
            P.s_sub_u32(S_A0.lo, S_NB, S_T[3])
            P.s_min_u32(S_A0.lo, S_A0.lo, S_PER)                # its length ...
            P.s_cmp_ge_u32(S_T[3], S_NB)
            P.s_cselect(S_A0.lo, Lit(0), S_A0.lo)               # ... 0 if it starts behind the end
            P.s_cmp_lt_u32(S_TK, S_A0.lo)
            P.s_cbranch_scc1(valid)
            P.s_sub_u32(S_TRIES, S_TRIES, Lit(1))
            P.s_cmp_eq_u32(S_TRIES, Lit(0))
            P.s_cbranch_scc1(none)
            P.s_add_u32(S_XR, S_XR, Lit(1))
            P.s_and_b32(S_XR, S_XR, Lit(7))
            P.s_add_u32(S_CTRX.lo, S_CTRX.lo, Lit(128))         # the counters lie 128 bytes apart, region 0's first
            P.s_addc_u32(S_CTRX.hi, S_CTRX.hi, Lit(0))
            P.s_cmp_eq_u32(S_XR, Lit(0))
            P.s_cselect(S_A0.lo, Lit(1024), Lit(0))
            P.s_sub_u32(S_CTRX.lo, S_CTRX.lo, S_A0.lo)
            P.s_subb_u32(S_CTRX.hi, S_CTRX.hi, Lit(0))
            P.s_mov(S_TK, Lit(1))
            P.s_atomic_add(S_TK, S_CTRX)
            P.s_wait_lgkm0_here()
            P.s_branch(retry)
            P.label(valid)
            P.s_add_u32(S_NEXT, S_T[3], S_TK)
            P.s_branch(pub)
            P.label(none)
            P.s_mov(S_NEXT, Lit(NONE))
            P.label(pub)
            ad = self.TMP[0].lo                      # (the first exchange's writes have read their data: the temporaries are free)
            P.v_mov(ad, S_LDS)
            P.v_mov(self.VT, S_NEXT)
            P.exec_lane0()
            P.ds_write_b32(ad, self.VT, LDS_MAIL)
            P.exec_all()
        self.if_wave0(body)

    # ------------------------------------------------------------ phases
    def prologue(self):
        P, HS, NT = self.P, self.HS, self.NT
        P.comment("---- prologue: wave type, addresses")
        T_ = self.TMP
        t, a, b, c, d, lane, wv, m = T_[0].lo, T_[0].hi, T_[1].lo, T_[1].hi, T_[2].lo, T_[2].hi, T_[3].lo, T_[3].hi
        tabs = (T_[5], T_[6])
        P.v_lshr(a, Lit(8), t)                                  # e = ke = T >> 8: wave-uniform
        P.v_and(b, Lit(255), t)
        P.v_lshl(self.VOFF, Lit(3), t)                          # 8 T
        P.v_lshl(b, Lit(3), b)
        P.v_add_u32(self.A_RZ, S_LDS, b)
        P.v_add_u32(self.A_RZH, Lit(8 * ROWS * HI_ROW), self.A_RZ)
        P.v_readfirstlane(S_HI, a)
        P.s_nop(4)
        P.s_cmp_eq_u32(S_HI, Lit(1))
        P.s_cselect(S_SIG.lo, Lit(-1.0), Lit(1.0))
        P.s_cselect(S_SIG.hi, Lit(-1.0), Lit(1.0))
        P.v_lshr(b, Lit(4), t)
        P.v_and(b, Lit(15), b)                                  # k1
        P.v_and(c, Lit(15), t)                                  # c
        P.v_mul_u24(b, Lit(8 * ROWS), b)
        P.v_add_u32(b, S_LDS, b)
        P.v_lshl(d, Lit(3), c)
        P.v_add_u32(self.A_R1, b, d)
        P.v_mul_u24(m, Lit(8 * HALF), a)
        P.v_add_u32(self.A_CH, self.A_R1, m)
        P.v_mul_u24(d, Lit(8 * KG), c)
        P.v_add_u32(self.A_X2, b, d)
        P.v_add_u32(self.A_X2, self.A_X2, m)
        P.v_lshl(m, Lit(11), a)
        P.v_lshl(d, Lit(3), c)
        P.v_add_u32(m, m, d)
        P.v_add_u32(self.A_TL, S_LDS, m)
        P.v_add_u32(self.A_TL, Lit(LDS_TWF), self.A_TL)
        # the first block: S_IN / S_OUT come as block 0's, the block's number in S_BLK
        P.s_mul_i32(S_A0.lo, S_BLK, Lit(4 * self.BLK))
        P.s_mul_hi_u32(S_A0.hi, S_BLK, Lit(4 * self.BLK))
        P.s_add_u32(S_IN.lo, S_IN.lo, S_A0.lo)
        P.s_addc_u32(S_IN.hi, S_IN.hi, S_A0.hi)
        P.s_add_u32(S_OUT.lo, S_OUT.lo, S_A0.lo)
        P.s_addc_u32(S_OUT.hi, S_OUT.hi, S_A0.hi)
        if self.ROT:
            P.s_add_u32(S_IN.lo, S_IN.lo, Lit(8 * self.ROT))
            P.s_addc_u32(S_IN.hi, S_IN.hi, Lit(0))
        # gather offsets: lane l (< 16) of a wave fetches the frame in front of this wave's first sample of row l; with an odd
        # number of context rows row 15 of the upper waves is the block's first half row (the rotation wraps there)
        P.v_and(lane, Lit(63), t)
        P.v_lshr(wv, Lit(6), t)
        P.v_lshl(wv, Lit(9), wv)                                # 512 B per wave
        P.v_lshl(a, Lit(12), lane)                              # 4096 lane
        P.v_add_u32(a, a, wv)
        P.v_mov(b, Lit(16))
        P.v_cmp_lt_u32(lane, b)                                 # lane < 16
        P.v_mov(b, Lit(0))
        P.v_cndmask(a, b, a)                                    # lanes without a row: offset 0 (a valid address)
        if self.ODD:
            # the upper waves' row 15 lies 65536 bytes lower (load_block_rows); the VGPR offset of a global access is unsigned, so
            # those waves gather from S_IN - 65536 with every other lane's offset 65536 higher
            P.v_mov(c, Lit(65536))
            P.v_add_u32(c, c, a)
            P.v_cmp_eq_u32(Lit(15), lane)
            P.v_cndmask(c, c, a)                                # lane 15: as it is
            P.v_mov(d, S_HI)
            P.v_cmp_eq_u32(Lit(1), d)                           # (every lane: an upper wave?)
            P.v_cndmask(self.VG, a, c)
        else:
            P.v_mov(self.VG, a)
        P.comment("---- prologue: twiddle registers, |B| factors, the two tables of the middle passes")
        for row, dst in ((0, self.TA1), (1, self.TB1), (4, self.RHO[0]), (2, tabs[0]), (3, tabs[1])):
            P.s_add_u32(S_A0.lo, S_SV.lo, Lit(4096 * row))
            P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
            P.global_load_dwordx2(dst, self.VOFF, S_A0, 0)
        return tabs, t

    def load_block_rows(self, soff, thunks=False):
        """all 16 rows of the block at S_IN + soff (an SGPR pair holding a 64-bit byte offset, or None) -> N; the gathered frames -> G_NX"""
        P = self.P
        if "nomem" in self.opts and soff is not None:
            return []
        nt = "nt" in self.opts
        ops = []
        def base(extra):
            # S_A0 = S_IN + soff + extra (extra may be negative)
            src = S_IN
            if soff is not None:
                P.s_add_u32(S_A0.lo, S_IN.lo, soff.lo)
                P.s_addc_u32(S_A0.hi, S_IN.hi, soff.hi)
                src = S_A0
            if extra < 0:
                P.s_sub_u32(S_A0.lo, src.lo, Lit(-extra))
                P.s_subb_u32(S_A0.hi, src.hi, Lit(0))
            else:
                P.s_add_u32(S_A0.lo, src.lo, Lit(extra))
                P.s_addc_u32(S_A0.hi, src.hi, Lit(0))
        def step(nbytes, sub=False):
            if sub:
                P.s_sub_u32(S_A0.lo, S_A0.lo, nbytes)
                P.s_subb_u32(S_A0.hi, S_A0.hi, Lit(0))
            else:
                P.s_add_u32(S_A0.lo, S_A0.lo, nbytes)
                P.s_addc_u32(S_A0.hi, S_A0.hi, Lit(0))
        for r in range(16):
            def f(r=r):
                if r == 0:
                    base(0)
                else:
                    step(Lit(4096))                 # (the thunks run in order: S_A0 walks from row to row)
                if self.ODD and r == 15:            # the upper waves' row 15 is the block's first half row
                    P.s_cmp_eq_u32(S_HI, Lit(1))
                    P.s_cselect(S_T[3], Lit(65536), Lit(0))
                    step(S_T[3], sub=True)
                P.global_load_dwordx2(self.N[r], self.VOFF, S_A0, 0, nt=nt)
            ops.append(f)
        def g():
            step(Lit(4096 * 15), sub=True)          # back to row 0 (less 65536 in the upper waves of an odd-context block: VG there)
            P.global_load_dword(self.G_NX, self.VG, S_A0, -4, nt=nt)
        ops.append(g)
        if thunks:
            return ops
        for f in ops:
            f()
        return []

    def prologue_rows(self, tabs, t):
        P = self.P
        P.comment("---- prologue: the first block's rows (requested early), the first ticket")
        self.load_block_rows(None)
        # wave 0 draws the ticket of the block after the first one
        P.s_mov(S_NEXT, Lit(0))                  # (anything but NONE: claim_issue draws)
        self.claim_issue()
        w = self.TMP[1].lo
        wt = self.TMP[1].hi
        P.v_add_u32(w, S_LDS, self.VOFF)
        P.v_add_u32(wt, Lit(LDS_TWF), w)
        P.pk("mul", tabs[1], [tabs[1], S_ROT], op_sel=[0, 1], op_sel_hi=[1, 1])
        P.ds_write_b64(wt, tabs[0], 0)
        P.ds_write_b64(wt, tabs[1], LDS_TWI - LDS_TWF)
        P.comment("---- prologue done")

    def build_pq(self):
        P, HS, R0 = self.P, self.HS, self.R0
        self.build_rows([(k, self.N[k], self.G_NX, k) for k in range(16)])
        for r in range(R0, R0 + HS):
            P.v_mov(self.K[r].lo, self.N[r].lo)
            P.v_mov(self.K[r].hi, self.N[r].hi)

    def loop(self):
        P, HS = self.P, self.HS
        P.s_waitcnt(vmcnt=0)
        P.label("L_top")
        self.stamp(0)
        # the next block's rows have arrived (its 16 row loads and the gather were issued in front of this block's HS stores);
        # no LDS operation is in flight here (the last ones were the reads of the fourth exchange, all consumed), and the scalar
        # atomic of wave 0 may be: lgkmcnt is not waited for
        P.s_waitcnt(vmcnt=HS)
        P.reset_queues(vm_stores=HS)
        self.stamp(1)
        P.comment("---- (p, q) pairs")
        self.build_pq()
        self.tap("pq", self.D)
        P.comment("---- forward pass 1 + exchange 1")
        self.fwd_pass1()
        self.claim_collect_publish()
        P.s_waitcnt(lgkmcnt=0)
        self.stamp(4)
        P.s_barrier()
        self.stamp(5)
        P.comment("---- forward pass 2 (the radix-2 layer first); the mailbox")
        P.v_mov(self.VT, S_LDS)
        P.ds_read_b32(self.VT, self.VT, LDS_MAIL)
        v = self.fwd_pass2_reads()
        self.tap("p2in", self.D)
        # byte offset from this block to the next one (0 behind the end: the prefetch then re-reads this block, in range)
        P.settle([self.VT])
        P.v_readfirstlane(S_NEXT, self.VT)
        P.s_nop(4)
        P.s_cmp_eq_u32(S_NEXT, Lit(NONE))
        P.s_cselect(S_T[3], S_BLK, S_NEXT)
        P.s_sub_u32(S_T[3], S_T[3], S_BLK)                      # blocks ahead: negative when the regions wrap around (a signed
        P.s_mul_hi_i32(S_ADV.hi, S_T[3], Lit(4 * self.BLK))     # 64-bit byte offset, added with carry)
        P.s_mul_i32(S_ADV.lo, S_T[3], Lit(4 * self.BLK))
        lds_ = self.load_block_rows(S_ADV, thunks=True)
        main = self.stage1_ops(v, 1) + self.stage1_twiddle_ops(v, 1, self.scratch_pairs())
        spread = [int(o[6:]) for o in self.opts if o.startswith("spread")] or [SPREAD]
        if spread[0] > 0:
            # the 17 loads of the prefetch one after every `spread` arithmetic instructions from here on (through the second and third
            # forward passes, the constants and the first inverse pass if need be) instead of all of them inside this stage: eight
            # waves that issue 17 x 512 bytes each at the same moment queue on the CU's one address path, and whatever stands behind
            # a load that cannot issue waits with it.  Measured (tools/ab_bench.py, ms per 2^27 frames, two rounds): all in the stage
            # 0.2460 / 0.2453, every 4: 0.2433 / 0.2426, 8: 0.2383 / 0.2392, 12: 0.2392 / 0.2366, 20: 0.2365 / 0.2379
            self.pending, self.pend_every, self.pend_n = list(lds_), spread[0], 0
            self.run_ops(main)
        else:
            self.run_ops(self.weave(main, lds_, max(1, len(main) // (len(lds_) + 1)) if lds_ else 1))
        self.twiddled_exchange8(v, 1, 0, self.A_X2, 8, self.A_CH, KG * 8)
        self.stamp(8)
        P.comment("---- forward pass 3")
        v = self.D
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, 1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        v = self.natural(v)
        P.comment("---- constants")
        self.pointwise(v)
        self.stamp(9)
        P.comment("---- inverse pass 1")
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        self.twiddled_exchange8(v, -1, LDS_TWI - LDS_TWF, self.A_CH, KG * 8, self.A_X2, 8)
        self.stamp(10)
        P.comment("---- inverse pass 2, the radix-2 layer's twiddle, exchange 4")
        self.flush_pending()
        v = self.D
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, -1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        u = self.natural(v)
        self.ke_twiddle(u, True)
        for b in range(16):
            P.ds_write_b64(self.A_CH, u[b], 128 * b)
        P.s_waitcnt(lgkmcnt=0)
        self.stamp(11)
        P.s_barrier()
        self.stamp(12)
        order = [4, 8, 12, 0, 5, 9, 13, 1, 6, 10, 14, 2, 7, 11, 15, 3]
        sets = [self.TMP[:4], self.TMP[4:8] if len(self.TMP) >= 8 else []]
        def issue(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                base, imm = self.row_slot(k)
                P.ds_read_b64(self.D[k], base, imm)
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                base, imm = self.row_slot(k)
                P.ds_read_b64(sets[j % 2][i], base, imm + 8 * HALF)
        def fold(j):
            for i, k in enumerate(order[4 * j:4 * j + 4]):
                P.pk("fma", self.D[k], [sets[j % 2][i], S_SIG, self.D[k]])
        if len(sets[1]) < 4:
            for j in range(4):
                sets[1] = sets[0]
                issue(j)
                fold(j)
        else:
            issue(0)
            for j in range(1, 4):
                issue(j)
                fold(j - 1)
            fold(3)
        # every LDS read of the block has been consumed: from here to the next block's first exchange no wait counts lgkmcnt --
        # the ticket of the block after next has that long to arrive
        P.settle(self.D)
        self.claim_issue()
        P.comment("---- inverse pass 3")
        u = self.inv_pass3()
        self.stamp(13)
        P.comment("---- mix and store")
        for k in range(HS):
            self.epilogue_row(u, k)
        self.stamp(14)
        self.stamps_fold()
        self.tick()
        # on to the next block
        P.s_cmp_eq_u32(S_NEXT, Lit(NONE))
        P.s_cbranch_scc1("L_done")
        P.s_sub_u32(S_T[3], S_NEXT, S_BLK)
        P.s_mul_hi_i32(S_ADV.hi, S_T[3], Lit(4 * self.BLK))
        P.s_mul_i32(S_ADV.lo, S_T[3], Lit(4 * self.BLK))
        P.s_add_u32(S_IN.lo, S_IN.lo, S_ADV.lo)
        P.s_addc_u32(S_IN.hi, S_IN.hi, S_ADV.hi)
        P.s_add_u32(S_OUT.lo, S_OUT.lo, S_ADV.lo)
        P.s_addc_u32(S_OUT.hi, S_OUT.hi, S_ADV.hi)
        P.s_mov(S_BLK, S_NEXT)
        P.s_branch("L_top")
        P.label("L_done")

    def build(self):
        tabs, t = self.prologue()
        self.prologue_rows(tabs, t)
        self.loop()
        return self.P


# ---------------------------------------------------------------------------------------------------- self-test
def device_table(HS, fir=None):
    """the per-engine constants as prot_engine.hip (save_tables8k) lays them out: five rows of 512 float2 -- W_N^{(T - 1/2)},
    W_N^{4 (T - 1/2)}, the half-shifted middle table, the plain middle table, the |B| factors of the rows next to DC --
    and the gain mean|B| / 8192"""
    import olasave_model8k as M
    NT = 16 - HS
    if fir is None:
        fir = M.hilbert_fir(1024 * NT)
    gt = fir[1::2]
    rho = -M.tables(gt)
    mean = rho[512:8192 - 512].mean()
    tab = np.zeros(2 * 5 * 512, np.float32)
    def put(idx, ang):
        tab[2 * idx], tab[2 * idx + 1] = math.cos(ang), math.sin(ang)
    for T in range(512):
        put(T, -2 * math.pi * (T - 0.5) / 8192)
        put(512 + T, -2 * math.pi * (T - 0.5) * 4 / 8192)
        ke, k2, c = T >> 8, (T >> 4) & 15, T & 15
        put(2 * 512 + T, -2 * math.pi * ((c - 0.5) * ke / 512 + (c - 0.5) * k2 / 256))
        put(3 * 512 + T, -2 * math.pi * (k2 * ke / 512 + c * k2 / 256))   # (inverse: the lane's digit is k2, the entry's row is c: transposed)
        k1, k2 = (T >> 4) & 15, T & 15          # pass-3 layout: the lane's low digit is k2
        q = k1 + 16 * ke + 32 * k2
        tab[2 * (4 * 512 + T)] = rho[q] / mean
        tab[2 * (4 * 512 + T) + 1] = rho[q + 7680] / mean
    return tab, np.float32(mean / 8192), gt


def self_test(HS, MODE, nblk=3, seed=1, verbose=True, opts=(), kind="run", thr2=0.30):
    g = Gen8k(HS, MODE, opts, kind)
    if kind == "core":
        nblk = 1
    P = g.build()
    NT, R0, BLK, ROT = g.NT, g.R0, g.BLK, g.ROT
    tab, gain, gt = device_table(HS)
    rng = np.random.default_rng(seed)
    n_frames = 16384 + BLK * nblk + 64
    x = (rng.random(n_frames, dtype=np.float32) - 0.5)
    a0 = 2          # frame index of local complex sample 0's aligned pair (its q) in the first block
    out = np.zeros(BLK * nblk + 16, np.float32)
    emu = Emu(lds_bytes=LDS_BYTES, n_threads=NTHR)
    IN_BASE, OUT_BASE, SV_BASE = 0x10000000, 0x20000000, 0x30000000
    emu.map(IN_BASE, x.view(np.uint32))
    emu.map(OUT_BASE, out.view(np.uint32))
    emu.map(SV_BASE, tab.view(np.uint32))
    ldsf = emu.lds.view(np.float32)
    ca, sa = 0.6, -0.8
    if MODE == MODE_SWEEP:   # S_ROT.lo is the bound, the Hilbert branch is unscaled, every wave appends to a list of its own
        ca, sa = thr2, 1.0
        SURV_BASE, CAP = 0x40000000, 2 * BLK * nblk
        surv = np.zeros(8 * CAP, np.float32)
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
    emu.v[g.TMP[0].n] = np.arange(NTHR, dtype=np.uint32)
    if MODE == MODE_SWEEP:
        for w in range(8):
            a_ = SURV_BASE + 4 * CAP * w
            emu.s[w, S_OUT.n], emu.s[w, S_OUT.n + 1] = a_ & 0xffffffff, a_ >> 32
        emu.v[g.V_RAW.n] = np.zeros(NTHR, np.float32).view(np.uint32)
    tt = np.arange(NTHR)
    if kind == "core":   # what the C++ code stages: position (r, T) <-> local sample (512 r + T + ROT) mod 8192, as (q, c) pairs
        for r in range(16):
            nloc = (512 * r + tt + ROT) % 8192
            a = a0 + 2 * nloc
            ldsf[(r * 512 + tt) * 2], ldsf[(r * 512 + tt) * 2 + 1] = x[a], x[a + 1]
        l, w = tt & 63, tt >> 6
        nloc = (512 * np.minimum(l, 15) + 64 * w + ROT) % 8192
        ldsf[LDS_GC // 4 + tt] = np.where(l < 16, x[a0 + 2 * nloc - 1], 0)
    emu.run(P)
    self_test.emu, self_test.x, self_test.a0, self_test.gen = emu, x, a0, g
    if kind == "core":
        for k in range(HS):
            out[1024 * k + 2 * tt], out[1024 * k + 2 * tt + 1] = ldsf[(k * 512 + tt) * 2], ldsf[(k * 512 + tt) * 2 + 1]
    # reference: direct convolution in float64.  z[j] = x[a0 + 2j - 1] + i x[a0 + 2j]; block b's local sample n is stream sample
    # j = 512 HS b + n; output row R0 + k, thread T <-> local n = 512 (R0 + k) + T + ROT <-> output frames b BLK + 1024 k + 2 T (+1)
    xd = x.astype(np.float64)
    nz = (n_frames - a0) // 2 - 1
    z = xd[a0 - 1:a0 - 1 + 2 * nz:2] + 1j * xd[a0:a0 + 2 * nz:2]
    full = np.convolve(z, gt)
    Lg = len(gt)
    want = np.zeros(BLK * nblk)
    for b in range(nblk):
        for k in range(HS):
            j = b * 512 * HS + 512 * (R0 + k) + ROT + np.arange(512)
            y = full[j + Lg // 2]
            dq, dc = xd[a0 + 2 * j], xd[a0 + 2 * j + 1]
            pos = b * BLK + 1024 * k + 2 * np.arange(512)
            if MODE == MODE_SWEEP:     # keep (d, h) per frame for the survivor check below
                want[pos], want[pos + 1] = y.real, y.imag
                if b == 0 and k == 0:
                    dry = np.zeros(BLK * nblk)
                dry[pos], dry[pos + 1] = dq, dc
            elif MODE == MODE_ROT:
                want[pos], want[pos + 1] = ca * dq + sa * y.real, ca * dc + sa * y.imag
            else:
                want[pos], want[pos + 1] = sa * y.real, sa * y.imag
    if MODE == MODE_SWEEP:
        # every wave's list: frames in the order (block, row, first / second frame of the pairs, lane); exact d, h to rounding,
        # membership by d^2 + h^2 > bound (frames within rounding of the bound may go either way)
        err = 0.0
        for w in range(8):
            end = (int(emu.s[w, S_OUT.n]) | (int(emu.s[w, S_OUT.n + 1]) << 32)) - (SURV_BASE + 4 * CAP * w)
            got = surv[CAP * w:CAP * w + end // 4].reshape(-1, 2)
            exp = []
            for b in range(nblk):
                for k in range(HS):
                    for half in range(2):
                        pos = b * BLK + 1024 * k + 2 * (64 * w + np.arange(64)) + half
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
                    pos = b * BLK + 1024 * k + 2 * lanes
                    rexp = np.maximum(rexp, np.maximum(np.abs(dry[pos]), np.abs(dry[pos + 1])))
            assert np.array_equal(raw, rexp.astype(np.float32)), "running max |x|"
        frac = sum(((int(emu.s[w, S_OUT.n]) | (int(emu.s[w, S_OUT.n + 1]) << 32)) - (SURV_BASE + 4 * CAP * w)) // 8 for w in range(8)) / (BLK * nblk)
        kind = kind + " sweep (%.1f %% of the frames survive)" % (100 * frac)
    else:
        err = np.abs(out[:BLK * nblk] - want).max()
    st = P.stats()
    npk = sum(v for k, v in st.items() if k.startswith("pk_"))
    nval = sum(v for k, v in st.items() if k.startswith("v_"))
    if verbose:
        print("8k %s HS %2d mode %d: max err %.2e over %d blocks; VGPRs %d (temporaries %d pairs); loop+prologue: %d packed, %d other VALU, "
              "%d ds, %d vmem, %d waits, %d barriers" % (kind, HS, MODE, err, nblk, g.n_vgpr, len(g.TMP), npk, nval,
                                                        st.get("ds_read_b64", 0) + st.get("ds_write_b64", 0) + st.get("ds_read_b32", 0),
                                                        st.get("gload2", 0) + st.get("gload1", 0) + st.get("gstore2", 0), st.get("s_waitcnt", 0), st.get("s_barrier", 0)))
    assert err < 1e-6, err
    return g, P


def dyn_claims(nb, xr, stride, ctr=None):
    """the blocks one workgroup transforms, in order, when every draw advances a counter by `stride` (the other workgroups' draws):
    the claim logic of Gen8kDyn / olasave8k_kernel's dynamic path restated"""
    per = (nb + 7) // 8
    ctr = [0] * 8 if ctr is None else ctr
    st = {"xr": xr, "tries": 8}
    def draw():
        t = ctr[st["xr"]]
        ctr[st["xr"]] += stride
        return t
    def claim():
        t = draw()
        while True:
            start = st["xr"] * per
            ln = 0 if start >= nb else min(per, nb - start)
            if t < ln:
                return start + t
            st["tries"] -= 1
            if st["tries"] == 0:
                return None
            st["xr"] = (st["xr"] + 1) & 7
            t = draw()
    out = []
    while True:
        b = claim()
        if b is None:
            return out, per
        out.append(b)


def self_test_dyn(HS, MODE, nb=11, xr=2, stride=3, seed=1, verbose=True, opts=()):
    """one workgroup of the dynamic loop in the emulator: it must transform exactly the blocks its tickets name (and leave the
    others alone), each against the float64 convolution"""
    g = Gen8kDyn(HS, MODE, opts)
    P = g.build()
    NT, R0, BLK, ROT = g.NT, g.R0, g.BLK, g.ROT
    tab, gain, gt = device_table(HS)
    rng = np.random.default_rng(seed)
    n_frames = 16384 + BLK * nb + 64
    x = (rng.random(n_frames, dtype=np.float32) - 0.5)
    a0 = 2
    out = np.zeros(BLK * nb + 16, np.float32)
    emu = Emu(lds_bytes=LDS_BYTES, n_threads=NTHR)
    emu.atomic_stride = stride
    IN_BASE, OUT_BASE, SV_BASE, CTR_BASE = 0x10000000, 0x20000000, 0x30000000, 0x50000000
    ctr = np.zeros(8 * 32, np.uint32)
    emu.map(IN_BASE, x.view(np.uint32))
    emu.map(OUT_BASE, out.view(np.uint32))
    emu.map(SV_BASE, tab.view(np.uint32))
    emu.map(CTR_BASE, ctr)
    ca, sa = 0.6, -0.8
    def sset(sp, lo, hi):
        emu.s[:, sp.n] = np.float32(lo).view(np.uint32)
        emu.s[:, sp.n + 1] = np.float32(hi).view(np.uint32)
    for n, (lo, hi) in g.constants().items():
        sset(SP(n), lo, hi)
    sset(S_ROT, ca, np.float32(sa) * gain)
    blocks, per = dyn_claims(nb, xr, stride)
    # what the C++ side does in front of the generated code: the workgroup's first claim
    st_ctr = [0] * 8
    first_xr, tries = xr, 8
    while True:
        t = st_ctr[first_xr]
        st_ctr[first_xr] += stride
        start = first_xr * per
        ln = 0 if start >= nb else min(per, nb - start)
        if t < ln:
            first = start + t
            break
        tries -= 1
        first_xr = (first_xr + 1) & 7
    assert first == blocks[0]
    for r in range(8):
        ctr[32 * r] = st_ctr[r]
    addr = IN_BASE + 4 * a0
    emu.s[:, S_IN.n], emu.s[:, S_IN.n + 1] = addr & 0xffffffff, addr >> 32
    emu.s[:, S_OUT.n], emu.s[:, S_OUT.n + 1] = OUT_BASE & 0xffffffff, OUT_BASE >> 32
    emu.s[:, S_SV.n], emu.s[:, S_SV.n + 1] = SV_BASE & 0xffffffff, SV_BASE >> 32
    emu.s[:, S_BLK.n] = first
    emu.s[:, S_LDS.n] = 0
    ca_ = CTR_BASE + 128 * first_xr
    emu.s[:, S_CTRX.n], emu.s[:, S_CTRX.n + 1] = ca_ & 0xffffffff, ca_ >> 32
    emu.s[:, S_XR.n] = first_xr
    emu.s[:, S_TRIES.n] = tries
    emu.s[:, S_PER.n] = per
    emu.s[:, S_NB.n] = nb
    emu.v[g.TMP[0].n] = np.arange(NTHR, dtype=np.uint32)
    emu.run(P)
    xd = x.astype(np.float64)
    nz = (n_frames - a0) // 2 - 1
    z = xd[a0 - 1:a0 - 1 + 2 * nz:2] + 1j * xd[a0:a0 + 2 * nz:2]
    full = np.convolve(z, gt)
    Lg = len(gt)
    want = np.zeros(BLK * nb)
    for b in blocks:
        for k in range(HS):
            j = b * 512 * HS + 512 * (R0 + k) + ROT + np.arange(512)
            y = full[j + Lg // 2]
            dq, dc = xd[a0 + 2 * j], xd[a0 + 2 * j + 1]
            pos = b * BLK + 1024 * k + 2 * np.arange(512)
            if MODE == MODE_ROT:
                want[pos], want[pos + 1] = ca * dq + sa * y.real, ca * dc + sa * y.imag
            else:
                want[pos], want[pos + 1] = sa * y.real, sa * y.imag
    err = np.abs(out[:BLK * nb] - want).max()
    st = P.stats()
    npk = sum(v for k, v in st.items() if k.startswith("pk_"))
    nval = sum(v for k, v in st.items() if k.startswith("v_"))
    if verbose:
        print("8k dyn HS %2d mode %d: blocks %s of %d (region %d, stride %d): max err %.2e; VGPRs %d (temporaries %d pairs); loop+prologue: %d packed, "
              "%d other VALU, %d ds, %d vmem, %d salu, %d waits, %d barriers" % (
                  HS, MODE, blocks, nb, xr, stride, err, g.n_vgpr, len(g.TMP), npk, nval,
                  sum(v for k, v in st.items() if k.startswith("ds_")), st.get("gload2", 0) + st.get("gload1", 0) + st.get("gstore2", 0),
                  sum(v for k, v in st.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_barrier", "s_nop", "s_waitcnt_x")),
                  st.get("s_waitcnt", 0), st.get("s_barrier", 0)))
    assert err < 1e-6, err
    return g, P


# HS = 8: the CLI's default engine at 48 kHz (blksiz 8192: 4096 odd taps = 8 context rows) as ONE table-free partition.
# (HS = 14, CLI blksiz 2048, passes the self-test too and is not emitted: the 4096-point kernel that serves it is faster)
VARIANTS = [(hs, mode) for hs in (8, 12, 13) for mode in (MODE_ROT, MODE_HILBERT)]
SWEEP_HS = (8, 12)   # MODE_SWEEP: the fused sweep of the CLI engines on these blocks (blksiz 8192, 4096)


def emit(path, opts=()):
    """csrc/prot_olasave8k_asm.inc: one macro per variant = the asm text of a run of blocks (ASM) or of one staged block (CORE),
    inputs from the inline-asm operands %0..%7 as in prot_olasave_asm.inc"""
    out = ["// GENERATED by tools/gen/gen_olasave8k.py -- do not edit.  The steady-state loop of olasave8k_kernel (8192-point blocks, 512",
           "// threads) as gfx950 assembly with hand-allocated registers; the emulator (tools/gen/isa.py) has run every variant",
           "// against the float64 convolution (tests/test_gen_olasave.py).",
           "// operands: %0 in (aligned pair of local sample 0 of the first block), %1 out (first output frame), %2 blocks, %3 LDS byte",
           "// offset of the exchange area, %4 constants table, %5 ca, %6 sa * gain, %7 thread id",
           "#pragma once", ""]
    nv = {}
    # "stamps": one more operand IN FRONT of the others (an in/out VGPR, "+v": lane i = cycles spent in interval i, summed over the
    # blocks), so the inputs are %1..%8 there
    sh = 1 if ("stamps" in opts or "tick" in opts) else 0
    for kind in ("run", "core"):
        for HS, MODE in VARIANTS:
            g = Gen8k(HS, MODE, opts if kind == "run" else [o for o in opts if o in ("nolds", "novalu")], kind)
            P = g.build()
            nv[HS] = max(nv.get(HS, 0), g.n_vgpr)
            head = ["s_mov_b64 %r, %%%d" % (S_IN, sh), "s_mov_b64 %r, %%%d" % (S_OUT, 1 + sh), "s_mov_b32 %r, %%%d" % (S_NBLK, 2 + sh),
                    "s_mov_b32 %r, %%%d" % (S_LDS, 3 + sh), "s_mov_b64 %r, %%%d" % (S_SV, 4 + sh), "s_mov_b32 %r, %%%d" % (S_ROT.lo, 5 + sh),
                    "s_mov_b32 %r, %%%d" % (S_ROT.hi, 6 + sh), "v_mov_b32 %r, %%%d" % (g.TMP[0].lo, 7 + sh), "s_waitcnt vmcnt(0) lgkmcnt(0)"]
            for n, (lo, hi) in sorted(g.constants().items()):
                head.append("s_mov_b32 s%d, %r" % (n, Lit(float(lo))))
                head.append("s_mov_b32 s%d, %r" % (n + 1, Lit(float(hi))))
            lines = head + P.text(label_suffix="%=")
            st = P.stats()
            out.append("// %s, HS %d, mode %d: %d VGPRs, %d instructions" % (kind, HS, MODE, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label", "probe"))))
            out.append("#define PROT_SV8_%s_%d_%d \\" % ("ASM" if kind == "run" else "CORE", HS, MODE))
            for l in lines:
                if l.startswith(";"):
                    continue
                out.append('\t"%s\\n\t" \\' % l)
            out.append('\t""')
            out.append("")
    # the dynamic loop (Gen8kDyn): operands %0 in (aligned pair of local sample 0 of the channel's whole-block 0), %1 out (first output
    # frame of block 0), %2 the workgroup's first block, %3 LDS byte offset, %4 constants table, %5 ca, %6 sa * gain, %7 thread id, %8
    # address of the counter of the region the first block was drawn from, %9 that region, %10 regions left to try, %11 blocks per
    # region, %12 blocks in all
    for HS, MODE in VARIANTS:
        g = Gen8kDyn(HS, MODE, [o for o in opts if o != "cyclic"])
        P = g.build()
        nv[HS] = max(nv.get(HS, 0), g.n_vgpr)
        head = ["s_mov_b64 %r, %%%d" % (S_IN, sh), "s_mov_b64 %r, %%%d" % (S_OUT, 1 + sh), "s_mov_b32 %r, %%%d" % (S_BLK, 2 + sh),
                "s_mov_b32 %r, %%%d" % (S_LDS, 3 + sh), "s_mov_b64 %r, %%%d" % (S_SV, 4 + sh), "s_mov_b32 %r, %%%d" % (S_ROT.lo, 5 + sh),
                "s_mov_b32 %r, %%%d" % (S_ROT.hi, 6 + sh), "v_mov_b32 %r, %%%d" % (g.TMP[0].lo, 7 + sh), "s_mov_b64 %r, %%%d" % (S_CTRX, 8 + sh),
                "s_mov_b32 %r, %%%d" % (S_XR, 9 + sh), "s_mov_b32 %r, %%%d" % (S_TRIES, 10 + sh), "s_mov_b32 %r, %%%d" % (S_PER, 11 + sh),
                "s_mov_b32 %r, %%%d" % (S_NB, 12 + sh), "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        for n, (lo, hi) in sorted(g.constants().items()):
            head.append("s_mov_b32 s%d, %r" % (n, Lit(float(lo))))
            head.append("s_mov_b32 s%d, %r" % (n + 1, Lit(float(hi))))
        lines = head + P.text(label_suffix="%=")
        st = P.stats()
        out.append("// dyn, HS %d, mode %d: %d VGPRs, %d instructions" % (HS, MODE, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label", "probe"))))
        out.append("#define PROT_SV8_DYN_%d_%d \\" % (HS, MODE))
        for l in lines:
            if l.startswith(";"):
                continue
            out.append('\t"%s\\n\t" \\' % l)
        out.append('\t""')
        out.append("")
    # MODE_SWEEP (run kind only; edge blocks take the Hilbert core and are tested by the C++ code), for the CLI engine that runs
    # on these blocks (blksiz 8192: HS = 8).  Operands as prot_olasave_asm.inc's PROT_SV_SWEEP_*: %0 survivor cursor (out), %1
    # running max |x| (out), %2 in, %3 blocks, %4 LDS byte offset, %5 constants table, %6 bound, %7 gain, %8 thread id, %9 and %10 the
    # cursor and the running maximum on entry
    for HS in SWEEP_HS:
        g = Gen8k(HS, MODE_SWEEP, [o for o in opts if o not in ("stamps", "tick")], "run")
        P = g.build()
        nv[HS] = max(nv.get(HS, 0), g.n_vgpr)
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
        out.append("#define PROT_SV8_SWEEP_%d \\" % HS)
        for l in lines:
            if l.startswith(";"):
                continue
            out.append('\t"%s\\n\t" \\' % l)
        out.append('\t""')
        out.append("")
    for HS in sorted(nv):
        clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % i for i in SGPR_CLOBBER] + ['"v%d"' % i for i in range(0, nv[HS])]
        if "stamps" in opts:
            clob += ['"v121"']
        out.append("#define PROT_SV8_ASM_CLOBBERS_%d " % HS + ", ".join(clob))
        if HS in SWEEP_HS:
            out.append("#define PROT_SV8_SWEEP_CLOBBERS_%d PROT_SV8_ASM_CLOBBERS_%d, \"v120\"" % (HS, HS))
    out.append("")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "..", "phaserotate.lv2_amd", "csrc")
    if "--variant" in sys.argv:   # timing experiments: python3 gen_olasave8k.py --variant NAME --opts a,b  -> csrc/variants/
        name = sys.argv[sys.argv.index("--variant") + 1]
        opts = [o for o in sys.argv[sys.argv.index("--opts") + 1].split(",") if o] if "--opts" in sys.argv else []
        wrong = {"nolds", "nomem", "novalu", "cyclic", "nob1", "nob3"} & set(opts)
        if not wrong:
            for HS, MODE in VARIANTS:
                self_test(HS, MODE, nblk=2, opts=opts, verbose=False)
                self_test_dyn(HS, MODE, nb=3, xr=1, stride=1, opts=opts, verbose=False)
        os.makedirs(os.path.join(csrc, "variants"), exist_ok=True)
        dst = os.path.join(csrc, "variants", "prot_olasave8k_asm_%s.inc" % name)
        emit(dst, opts)
        print("wrote", os.path.normpath(dst), "(results WRONG by design)" if wrong else "(self-test passed)")
        sys.exit(0)
    for HS, MODE in VARIANTS:
        self_test(HS, MODE, nblk=2 if "--quick" in sys.argv else 3)
        self_test(HS, MODE, kind="core")
    for HS in SWEEP_HS:
        self_test(HS, MODE_SWEEP, nblk=2)
    for HS, MODE in VARIANTS:
        self_test_dyn(HS, MODE, nb=4 if "--quick" in sys.argv else 7, xr=(3 * HS) & 7, stride=2 if MODE else 1)
    if "--check" not in sys.argv:
        dst = os.path.join(csrc, "prot_olasave8k_asm.inc")
        emit(dst)
        print("wrote", os.path.normpath(dst))
