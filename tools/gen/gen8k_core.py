#!/usr/bin/env python3
"""Gen8k: the steady-state loop of olasave8k_kernel with a static run of blocks per workgroup (and the "core" kind of the
16384-point blocks); the constants of its LDS / SGPR plan; device_table() = the per-engine constants as the host lays them out.
The description of the block, the self-tests and the emitter are in gen_olasave8k.py (which re-exports everything here);
Gen8kDyn (blocks drawn from a counter: the shipped headline loop) is gen8k_dyn.py."""
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
