#!/usr/bin/env python3
"""Generator of the steady-state loop of olasave16k_kernel (phaserotate.lv2_amd/csrc/prot_olasave16k.hip): the overlap-save, table-free
polyphase block of gen_olasave8k.py on a 16384-POINT transform pair -- 1024 threads (16 waves), 16 points per thread, ONE
workgroup per CU (154 KiB of LDS) -- for the CLI's default engine at 88.2 / 96 kHz (blksiz 16384, cli/phase-rotate.cc:749-755:
8192 odd taps = 8 context rows of 1024 complex samples, 8 rows = 16384 frames new per block).  Until round 5 that engine's
Hilbert pass was two accumulating olapoly_kernel launches of two 2048-tap partitions each, through a workspace in HBM (17 bytes
per channel-frame by counter against the sweep's algorithmic 4); this is ONE launch that reads the stream once.

    python3 tools/gen/gen_olasave16k.py            -> csrc/prot_olasave16k_asm.inc  (+ self-test of every variant in the emulator)
    python3 tools/gen/gen_olasave16k.py --check    -> self-test only

The maths is tools/model/olasave_model16k.py: thread T = 256 e + 16 b + c keeps positions p = 1024 a + T; pass 1 over a; ONE
radix-4 layer over the two top bits e of T, done by the READER of the first LDS exchange (thread (k1, ke, c) reads the four
quarters of its column and takes u0 + (-i)^ke u1 + (-1)^ke u2 + (+i)^ke u3: three packed adds with swizzles, one code variant per
ke -- ke = T >> 8 is wave-uniform); the layer's twiddle W64^{b ke} from the 15 constants W32^k the pointwise stage keeps in
SGPRs anyway and ONE more, W64^1 (an odd exponent costs a second multiply); passes 2 and 3 over b and c as on 8192 points; the
inverse mirrors it.  Four workgroup barriers per block as there."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "model"))
from isa import Prog, Emu, V, VP, S, SP, Lit  # noqa: E402
import gen_olasave as G4  # noqa: E402
from gen_olasave import S_IN, S_OUT, S_NBLK, S_LDS, S_SV, S_ROT, S_C32, S_RR, S_W1, S_W3, S_W9, S_T, S_A0, MODE_ROT, MODE_HILBERT, MODE_SWEEP  # noqa: E402
from gen_olasave8k import Gen8k, S_TWA, S_TWB, KG  # noqa: E402

NTHR = 1024
CH = 272                            # slots (8 B) per chunk = one quarter (e / ke) of a k1 row: 16 x 17
ROWS = 4 * CH + 16                  # slots per k1 row of the exchange area
ROWB = 8 * NTHR                     # bytes per row of a block in the stream: 1024 complex samples
LDS_EX = 16 * ROWS * 8              # 141312
LDS_TWF, LDS_TWI = LDS_EX, LDS_EX + 8192       # the two twiddle tables of the middle passes (1024 entries each)
LDS_BYTES = LDS_EX + 16384          # 157696 of the CU's 163840
LDS_GC = 16 * ROWB                  # core blocks: the gathered frames behind the 16 staged rows
WROWS = 7                           # rows of the exchange area one base register reaches (DS offsets are 16 bits: 7 x 8832 < 65536)

S_W64 = SP(86)                      # W64^1 (the 8192-point loop keeps its radix-2 sign here)
S_KE = S(88)                        # e = ke = T >> 8 of the wave (0..3)
SGPR_CLOBBER = range(34, 100)


class Gen16k(Gen8k):
    def __init__(self, HS, MODE, opts=(), kind="run"):
        assert HS in (8, 12) and (16 - HS) % 2 == 0
        self.HS, self.MODE, self.NA = HS, MODE, 1
        self.opts = set(opts)
        self.kind = kind
        self.NT = 16 - HS
        self.R0 = self.NT // 2
        self.ODD, self.ROT = 0, 0
        self.BLK = 2 * NTHR * HS               # frames per block
        P = self.P = Prog()
        self.sw_rows, self.sw_label = [], 0
        self.adv = 4 * self.BLK
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
        self.TA1, self.TB1 = VP(n), VP(n + 2)
        n += 4
        self.RHO = [VP(n)]
        n += 2
        self.G_CAR, self.G_NX = V(n), V(n + 1)
        n += 2
        # write bases of thread T (its slot of rows 0.., 7.., 14..), read bases of its column (chunk 0 of the same rows), the group's
        # row / chunk / transposed-chunk / table addresses, the stream offset, the gather offset
        (self.A_W0, self.A_W7, self.A_W14, self.A_C0, self.A_C7, self.A_C14, self.A_R1, self.A_CH, self.A_X2, self.A_TL, self.VOFF,
         self.VG) = [V(n + i) for i in range(12)]
        n += 12
        self.TMP = []
        cap = 119
        while n + 1 <= cap:
            self.TMP.append(VP(n))
            n += 2
        self.n_vgpr = n
        assert len(self.TMP) >= 8, "temporaries: %d" % len(self.TMP)
        if MODE == MODE_SWEEP:
            self.V_RAW = V(120)
        self.TB = self.TMP[:4]
        self.TR = self.TMP[4:5]
        self.TT = self.TMP[5:9] if len(self.TMP) >= 9 else self.TMP[5:6] * 4
        self.nbf = len(set(t.n for t in self.TT))
        self.lab = 0

    # ------------------------------------------------------------ geometry
    def wslot(self, k1):
        """(base register, immediate) of thread T's slot in row k1 (its own chunk e)"""
        base = (self.A_W0, self.A_W7, self.A_W14)[k1 // WROWS]
        return base, 8 * ROWS * (k1 % WROWS)

    def cslot(self, k1, j):
        """(base register, immediate) of chunk j of thread T's column in row k1"""
        base = (self.A_C0, self.A_C7, self.A_C14)[k1 // WROWS]
        return base, 8 * ROWS * (k1 % WROWS) + 8 * CH * j

    def switch_ke(self, variants):
        """variants[ke]() for the wave's ke = 0..3: four copies of a region, each wave runs one.  Every copy must leave the same
        operations in flight (the generator's wait bookkeeping merges them) and contain no inserted wait that the others lack."""
        P = self.P
        labs = [self.label("L_ke%d" % k) for k in range(4)]
        done = self.label("L_ked")
        st0 = P.fork()
        ends = []
        for k in range(1, 4):
            P.s_cmp_eq_u32(S_KE, Lit(k))
            P.s_cbranch_scc1(labs[k])
        for k in range(4):
            if k:
                P.label(labs[k])
            P.restore(st0)
            variants[k]()
            ends.append(P.fork())
            if k < 3:
                P.s_branch(done)
        P.label(done)
        for e_ in ends[:-1]:
            P.join(e_)

    def cmul_c32n(self, d, a, k, conj, r, neg=False):
        """d = +-a * W32^k (conj: W32^-k), k = 0..15 (0: a sign or nothing); as Gen8k.cmul_c32 with the sign folded into the modifiers"""
        P = self.P
        if k == 0:
            return [lambda: P.pk("mul", d, [a, S_C32[8]], op_sel=[0, 0], op_sel_hi=[1, 0])] if neg else []   # S_C32[8] = (-1, -1)
        if k == 8:      # -i (conj: +i), negated: +i (conj: -i)
            if (not conj) != neg:
                return [lambda: P.pk("mul", d, [a, S_C32[8]], op_sel=[1, 0], op_sel_hi=[0, 0], neg_lo=[1, 0])]   # (a.hi, -a.lo) = -i a
            return [lambda: P.pk("mul", d, [a, S_C32[8]], op_sel=[1, 0], op_sel_hi=[0, 0], neg_hi=[1, 0])]       # (-a.hi, a.lo) = +i a
        w = S_C32[k]
        s = 1 if neg else 0
        if not conj:
            return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 1], op_sel_hi=[0, 0], neg_lo=[s, 0], neg_hi=[s, 0]),
                    lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 0, 0], op_sel_hi=[1, 1, 1], neg_lo=[s, 1, 0], neg_hi=[s, 0, 0])]
        return [lambda: P.pk("mul", r, [a, w], op_sel=[0, 1], op_sel_hi=[0, 0], neg_lo=[s, 0], neg_hi=[s, 1]),
                lambda: P.pk("fma", d, [a, w, r], op_sel=[1, 0, 0], op_sel_hi=[1, 1, 1], neg_lo=[s, 0, 0], neg_hi=[s, 0, 0])]

    def mul_w64(self, x, m, conj, r):
        """thunks for x *= W64^m (conj: W64^-m), m = 0..63: W32^{m >> 1} (a sign for the upper half) and W64^1 for an odd m"""
        m %= 64
        q, odd = m >> 1, m & 1
        ops = self.cmul_c32n(x, x, q % 16, conj, r, neg=q >= 16)
        if odd:
            P = self.P
            if not conj:
                ops += [lambda: P.pk("mul", r, [x, S_W64], op_sel=[0, 0], op_sel_hi=[0, 1]),
                        lambda: P.pk("fma", x, [x, S_W64, r], op_sel=[1, 1, 0], op_sel_hi=[1, 0, 1], neg_lo=[0, 1, 0])]
            else:
                ops += [lambda: P.pk("mul", r, [x, S_W64], op_sel=[0, 0], op_sel_hi=[0, 1], neg_hi=[0, 1]),
                        lambda: P.pk("fma", x, [x, S_W64, r], op_sel=[1, 1, 0], op_sel_hi=[1, 0, 1])]
        return ops

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
        P.v_add_u32(self.A_C0, S_LDS, b)                        # lds + 8 (T & 255): chunk 0 of a column
        P.v_add_u32(self.A_C7, Lit(8 * ROWS * WROWS), self.A_C0)
        P.v_add_u32(self.A_C14, Lit(8 * ROWS * 2 * WROWS), self.A_C0)
        P.v_mul_u24(m, Lit(8 * CH), a)                          # 2176 e
        P.v_add_u32(self.A_W0, self.A_C0, m)                    # the thread's own slot (chunk e) of rows 0..6
        P.v_add_u32(self.A_W7, self.A_C7, m)
        P.v_add_u32(self.A_W14, self.A_C14, m)
        P.v_readfirstlane(S_KE, a)
        P.s_nop(4)
        # passes 2 and 3: thread (k1, ke, c), k1 = (T >> 4) & 15, c = T & 15
        P.v_lshr(b, Lit(4), t)
        P.v_and(b, Lit(15), b)                                  # k1
        P.v_and(c, Lit(15), t)                                  # c
        P.v_mul_u24(b, Lit(8 * ROWS), b)
        P.v_add_u32(b, S_LDS, b)                                # lds + 8 ROWS k1
        P.v_lshl(d, Lit(3), c)                                  # 8 c
        P.v_add_u32(self.A_R1, b, d)                            # chunk 0 of row k1, column c
        P.v_add_u32(self.A_CH, self.A_R1, m)                    # the group's own chunk (= its quarter of the row), + c
        P.v_mul_u24(d, Lit(8 * KG), c)
        P.v_add_u32(self.A_X2, b, d)
        P.v_add_u32(self.A_X2, self.A_X2, m)                    # chunk + 17 c
        P.v_lshl(m, Lit(11), a)                                 # 2048 ke
        P.v_lshl(d, Lit(3), c)
        P.v_add_u32(m, m, d)
        P.v_add_u32(self.A_TL, S_LDS, m)
        P.v_add_u32(self.A_TL, Lit(LDS_TWF), self.A_TL)         # lds + tables + 8 (256 ke + c)   (+ 0 / 8192 + 128 k2 as immediates)
        # gather offsets: lane l of a wave fetches the frame in front of this wave's first sample of the row N[l] holds
        P.v_and(lane, Lit(63), t)
        P.v_lshr(wv, Lit(6), t)
        P.v_lshl(wv, Lit(9), wv)                                # 512 B per wave
        P.v_add_u32(a, Lit(NT), lane)                           # N[l] = row NT + l
        P.v_lshl(a, Lit(13), a)
        P.v_mov(b, Lit(HS))
        P.v_cmp_lt_u32(lane, b)                                 # lane < HS
        P.v_mov(b, Lit(0))
        P.v_cndmask(a, b, a)                                    # lanes without a row: offset 0 (a valid address)
        P.v_add_u32(self.VG, a, wv)
        if self.kind != "core":
            self.first_gather_offsets(vg0, lane, wv, [a, b, c])
        P.comment("---- prologue: twiddle registers, |B| factors, the two tables of the middle passes")
        for row, dst in ((0, self.TA1), (1, self.TB1), (4, self.RHO[0]), (2, tabs[0]), (3, tabs[1])):
            P.s_add_u32(S_A0.lo, S_SV.lo, Lit(ROWB * row))
            P.s_addc_u32(S_A0.hi, S_SV.hi, Lit(0))
            P.global_load_dwordx2(dst, self.VOFF, S_A0, 0)
        return tabs, t, vg0

    def first_gather_offsets(self, vg0, lane, wv, tmp):
        """vg0[lane] = byte offset (relative to one row BELOW the block) of the context row whose front frame that lane of G_CAR
        must hold for the first build_pq (lane HS - NT + k <-> carried row k), + 512 wave; other lanes: 0"""
        P, HS, NT = self.P, self.HS, self.NT
        x, y, z = tmp
        lo = HS - NT
        P.v_add_u32(x, Lit(1 - lo), lane)                       # row candidate + 1
        P.v_lshl(x, Lit(13), x)
        P.v_mov(z, Lit(ROWB))
        P.v_mov(y, Lit(lo))
        P.v_cmp_lt_u32(lane, y)
        P.v_cndmask(x, x, z)                                    # lanes below: the row below the block (offset ROWB - ROWB = 0 of the base used)
        P.v_mov(y, Lit(lo + NT - 1))
        P.v_cmp_lt_u32(y, lane)
        P.v_cndmask(x, x, z)                                    # lanes above: likewise
        P.v_add_u32(vg0, x, wv)

    def new_row_bytes(self, i):
        return ROWB * (self.NT + i), False

    def prologue_rows(self, tabs, t, vg0):
        P, HS, NT = self.P, self.HS, self.NT
        core = self.kind == "core"
        if not core:
            P.comment("---- prologue: the first block's rows (requested early)")
            for k in range(NT):
                self.load_row(self.K[HS + k], ROWB * k)
            P.s_sub_u32(S_A0.lo, S_IN.lo, Lit(ROWB))
            P.s_subb_u32(S_A0.hi, S_IN.hi, Lit(0))
            P.global_load_dword(self.G_CAR, vg0, S_A0, -4)
            self.load_new_rows(0)
        w = self.TMP[1].lo
        wt = self.TMP[1].hi
        P.v_add_u32(w, S_LDS, self.VOFF)
        P.v_add_u32(wt, Lit(LDS_TWF), w)
        P.pk("mul", tabs[1], [tabs[1], S_ROT], op_sel=[0, 1], op_sel_hi=[1, 1])
        P.ds_write_b64(wt, tabs[0], 0)
        P.ds_write_b64(wt, tabs[1], LDS_TWI - LDS_TWF)
        if core:
            P.comment("---- prologue: the staged block")
            w2 = self.TMP[2].hi
            P.v_add_u32(w2, Lit(8 * ROWB), w)                   # rows 8..15 (DS offsets are 16 bits)
            self.SRC = {}
            for r in range(16):
                self.SRC[r] = self.K[r] if r >= self.R0 else self.N[r]
                P.ds_read_b64(self.SRC[r], w if r < 8 else w2, ROWB * (r % 8))
            a4 = self.TMP[2].lo
            P.v_lshl(a4, Lit(2), t)
            P.v_add_u32(a4, S_LDS, a4)
            P.v_add_u32(a4, Lit(LDS_GC), a4)
            P.ds_read_b32(self.G_CAR, a4, 0)
            P.s_waitcnt(vmcnt=0, lgkmcnt=0)
            P.s_barrier()   # every wave has its rows: the exchange area may be written
        P.comment("---- prologue done")

    # ---- forward pass 1: rows -> k1, outer twiddle, first exchange
    def fwd_pass1(self):
        P = self.P
        v = self.D
        self.stage1(v, 1)
        self.run_ops(self.stage1_twiddle_ops(v, 1, self.scratch_pairs()))
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
        # the previous block's last exchange has been read by all four waves that share a column: its slots may be written
        self.stamp(2)
        P.s_barrier()
        self.stamp(3)
        n = 0
        for ka in range(4):
            base, imm = self.wslot(ka)
            P.ds_write_b64(base, v[4 * ka], imm)
        for la in range(1, 4):
            for ka in range(4):
                x = v[4 * ka + la]
                self.cmul(x, x, B[la], 1, sc[n % len(sc)])
                n += 1
                base, imm = self.wslot(ka + 4 * la)
                P.ds_write_b64(base, x, imm)

    def layer_reads(self, slot_of, order, DIR):
        """the reader's half of the radix-4 layer: D[k] <- the DFT-4 output number ke (forward) / e (inverse) of the four chunks of
        slot k, two k at a time with the next pair's reads in flight; then the layer's twiddle (forward only: the inverse applies it
        in front of its writes)"""
        P = self.P
        v = self.D
        two = len(self.TMP) >= 12      # (two pairs of values in flight, or one when the temporaries are few: HS = 12)
        sets = [self.TMP[0:6], self.TMP[6:12] if two else self.TMP[0:6]]

        def variant(ke):
            def run():
                def issue(j):
                    for i, k in enumerate(order[2 * j:2 * j + 2]):
                        for q in range(4):
                            base, imm = slot_of(k, q)
                            P.ds_read_b64(v[k] if q == 0 else sets[j % 2][3 * i + q - 1], base, imm)

                def fold(j):
                    for i, k in enumerate(order[2 * j:2 * j + 2]):
                        u1, u2, u3 = sets[j % 2][3 * i:3 * i + 3]
                        if ke in (0, 2):
                            self.add(v[k], v[k], u2)
                            self.add(u1, u1, u3)
                            (self.add if ke == 0 else self.sub)(v[k], v[k], u1)
                        else:
                            self.sub(v[k], v[k], u2)
                            self.sub(u1, u1, u3)
                            # forward: ke = 1: t0 - i t1, ke = 3: t0 + i t1; inverse (conjugate kernel): e = 1: t0 + i t1, e = 3: t0 - i t1
                            minus_i = (ke == 1) == (DIR > 0)
                            (self.add_mj if minus_i else self.sub_mj)(v[k], v[k], u1)
                if two:
                    issue(0)
                    for j in range(1, 8):
                        issue(j)
                        fold(j - 1)
                    fold(7)
                else:
                    for j in range(8):
                        issue(j)
                        fold(j)
                if DIR > 0 and ke:
                    sc = self.TMP[:]
                    lists = [self.mul_w64(v[b], b * ke, False, sc[(b - 1) % len(sc)]) for b in range(1, 16)]
                    for g in range(0, len(lists), len(sc)):
                        self.run_ops(self.interleave(lists[g:g + len(sc)]))
            return run
        P.settle(list(v) + self.TMP)
        self.no_weave = True
        self.switch_ke([variant(k) for k in range(4)])
        self.no_weave = False

    def ke_twiddle_inv(self, u):
        """the inverse layer's twiddle conj(W64^{b ke}) on the outputs b of the second inverse pass, in the waves with ke != 0"""
        P = self.P

        def variant(ke):
            def run():
                if not ke:
                    return
                sc = self.TMP[:]
                lists = [self.mul_w64(u[b], b * ke, True, sc[(b - 1) % len(sc)]) for b in range(1, 16)]
                for g in range(0, len(lists), len(sc)):
                    self.run_ops(self.interleave(lists[g:g + len(sc)]))
            return run
        P.settle(list(u) + self.TMP)
        self.no_weave = True
        self.switch_ke([variant(k) for k in range(4)])
        self.no_weave = False

    def epilogue_row(self, u, k):
        P, R0 = self.P, self.R0
        x = u[R0 + k]
        if self.MODE == MODE_SWEEP:
            self.sweep_row(self.K[R0 + k], x)
            return
        if self.MODE == MODE_ROT:
            P.pk("fma", x, [self.K[R0 + k], S_ROT, x], op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1])
        if self.kind == "core":
            P.ds_write_b64(self.A_OUT if k < 8 else self.A_OUT2, x, ROWB * (k % 8))   # (DS offsets are 16 bits)
        elif "nomem" not in self.opts:
            P.s_add_u32(S_A0.lo, S_OUT.lo, Lit(ROWB * k))
            P.s_addc_u32(S_A0.hi, S_OUT.hi, Lit(0))
            P.global_store_dwordx2(self.VOFF, x, S_A0, 0)

    def loop(self):
        P, HS = self.P, self.HS
        core = self.kind == "core"
        if not core:
            P.s_waitcnt(vmcnt=0)
            P.label("L_top")
            n_st = 0 if self.MODE == MODE_SWEEP else HS
            if self.MODE == MODE_SWEEP:
                P.s_waitcnt(lgkmcnt=0)
            else:
                P.s_waitcnt(vmcnt=n_st, lgkmcnt=0)
            P.reset_queues(vm_stores=n_st)
        P.comment("---- (p, q) pairs")
        self.build_pq()
        self.tap("pq", self.D)
        P.comment("---- forward pass 1 + exchange 1")
        self.fwd_pass1()
        P.s_waitcnt(lgkmcnt=0)
        P.s_barrier()
        P.comment("---- forward pass 2 (the radix-4 layer first)")
        order = [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
        # thread (k1, ke, c) reads value b of chunk q of its row at slot 16 b + c of the chunk
        self.layer_reads(lambda k, q: (self.A_R1, 128 * k + 8 * CH * q), order, 1)
        v = self.D
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
        P.comment("---- inverse pass 1")
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        self.twiddled_exchange8(v, -1, LDS_TWI - LDS_TWF, self.A_CH, KG * 8, self.A_X2, 8)
        P.comment("---- inverse pass 2, the radix-4 layer's twiddle, exchange 4")
        v = self.D
        self.tap("ip2in", self.D)
        self.stage1(v, -1)
        self.run_ops(self.stage1_twiddle_ops(v, -1, self.scratch_pairs()))
        bfl = self.stage2_lists(v, -1)
        self.run_ops(self.interleave(bfl) if self.nbf >= 4 else sum(bfl, []))
        u = self.natural(v)          # u[b] = output b of the pass
        self.tap("ip2out", u)
        self.ke_twiddle_inv(u)
        self.tap("ip2tw", u)
        for b in range(16):
            P.ds_write_b64(self.A_CH, u[b], 128 * b)   # into the group's own chunk: wave-local, no barrier in front
        P.s_waitcnt(lgkmcnt=0)
        P.s_barrier()
        # the reader's half of the inverse layer: thread T takes the DFT-4 output e of the four chunks of its column
        order = [4, 8, 12, 0, 5, 9, 13, 1, 6, 10, 14, 2, 7, 11, 15, 3]
        self.layer_reads(self.cslot, order, -1)
        self.tap("ip3in", self.D)
        P.comment("---- inverse pass 3")
        u = self.inv_pass3()
        self.tap("ip3out", u)
        P.comment("---- mix and store")
        if core:
            P.s_barrier()   # every wave has read its share of the last exchange: the area takes the output rows
            self.A_OUT, self.A_OUT2 = self.TMP[0].lo, self.TMP[0].hi
            P.v_add_u32(self.A_OUT, S_LDS, self.VOFF)
            P.v_add_u32(self.A_OUT2, Lit(8 * ROWB), self.A_OUT)
            for k in range(HS):
                self.epilogue_row(u, k)
            P.s_waitcnt(lgkmcnt=0)
            return
        for k in range(HS):
            self.epilogue_row(u, k)
        self.sweep_flush()
        P.s_add_u32(S_IN.lo, S_IN.lo, Lit(self.adv))
        P.s_addc_u32(S_IN.hi, S_IN.hi, Lit(0))
        if self.MODE != MODE_SWEEP:
            P.s_add_u32(S_OUT.lo, S_OUT.lo, Lit(self.adv))
            P.s_addc_u32(S_OUT.hi, S_OUT.hi, Lit(0))
        P.s_sub_u32(S_NBLK, S_NBLK, Lit(1))
        P.s_cmp_lg_u32(S_NBLK, Lit(0))
        P.s_cbranch_scc1("L_top")

    def twiddled_exchange8(self, v, DIR, tw_off, waddr, wstride, raddr, rstride):
        """Gen8k's, with the barrier's comment of this block: every wave has read the four quarters of its columns of the first exchange"""
        P = self.P

        def fetch(ka):
            for la in range(4):
                P.ds_read_b64(self.TB[la], self.A_TL, tw_off + 128 * (ka + 4 * la))
        bfl = [self.stage2_ops(v, ka, DIR, self.TT[0]) for ka in range(4)]
        r = self.TR[0]
        fetch(0)
        if DIR > 0:
            P.s_barrier()
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

    def constants(self):
        c = {}
        for k in range(1, 16):
            th = math.pi / 2 + math.pi * k / 16
            c[S_C32[k].n] = (math.cos(th), math.sin(th)) if k != 8 else (-1.0, -1.0)
        c[S_RR.n] = (G4.RR, -G4.RR)
        c[S_W1.n] = (G4.C1, -G4.S1)
        c[S_W3.n] = (G4.S1, -G4.C1)
        c[S_W9.n] = (-G4.C1, G4.S1)
        c[S_TWA.n] = (math.cos(2 * math.pi / 32768), -math.sin(2 * math.pi / 32768))
        c[S_TWB.n] = (math.cos(2 * math.pi * 4 / 32768), -math.sin(2 * math.pi * 4 / 32768))
        c[S_W64.n] = (math.cos(2 * math.pi / 64), -math.sin(2 * math.pi / 64))
        return c


# ---------------------------------------------------------------------------------------------------- self-test
def device_table(HS, fir=None):
    """the per-engine constants as prot_engine.hip (save_tables16k) lays them out: five rows of 1024 float2 -- W_N^{(T - 1/2)},
    W_N^{4 (T - 1/2)}, the half-shifted middle table, the plain middle table, the |B| factors of the rows next to DC -- and the gain
    mean|B| / 16384"""
    import olasave_model16k as M
    NT = 16 - HS
    if fir is None:
        fir = M.hilbert_fir(2048 * NT)
    gt = fir[1::2]
    rho = -M.tables(gt)
    mean = rho[1024:16384 - 1024].mean()
    tab = np.zeros(2 * 5 * 1024, np.float32)

    def put(idx, ang):
        tab[2 * idx], tab[2 * idx + 1] = math.cos(ang), math.sin(ang)
    for T in range(1024):
        put(T, -2 * math.pi * (T - 0.5) / 16384)
        put(1024 + T, -2 * math.pi * (T - 0.5) * 4 / 16384)
        ke, k2, c = T >> 8, (T >> 4) & 15, T & 15
        put(2 * 1024 + T, -2 * math.pi * ((c - 0.5) * ke / 1024 + (c - 0.5) * k2 / 256))
        put(3 * 1024 + T, -2 * math.pi * (k2 * ke / 1024 + c * k2 / 256))   # (inverse: the lane's digit is k2, the entry's row is c: transposed)
        k1, k2 = (T >> 4) & 15, T & 15          # pass-3 layout: the lane's low digit is k2
        q = k1 + 16 * ke + 64 * k2
        tab[2 * (4 * 1024 + T)] = rho[q] / mean
        tab[2 * (4 * 1024 + T) + 1] = rho[q + 15360] / mean
    return tab, np.float32(mean / 16384), gt


def self_test(HS, MODE, nblk=2, seed=1, verbose=True, opts=(), kind="run", thr2=0.30):
    g = Gen16k(HS, MODE, opts, kind)
    if kind == "core":
        nblk = 1
    P = g.build()
    NT, R0, BLK = g.NT, g.R0, g.BLK
    tab, gain, gt = device_table(HS)
    rng = np.random.default_rng(seed)
    n_frames = 32768 + BLK * nblk + 64
    x = (rng.random(n_frames, dtype=np.float32) - 0.5)
    a0 = 2
    out = np.zeros(BLK * nblk + 16, np.float32)
    emu = Emu(lds_bytes=LDS_BYTES, n_threads=NTHR)
    IN_BASE, OUT_BASE, SV_BASE = 0x10000000, 0x20000000, 0x30000000
    emu.map(IN_BASE, x.view(np.uint32))
    emu.map(OUT_BASE, out.view(np.uint32))
    emu.map(SV_BASE, tab.view(np.uint32))
    ldsf = emu.lds.view(np.float32)
    ca, sa = 0.6, -0.8
    NW = NTHR // 64
    if MODE == MODE_SWEEP:
        ca, sa = thr2, 1.0
        SURV_BASE, CAP = 0x40000000, 2 * BLK * nblk
        surv = np.zeros(NW * CAP, np.float32)
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
        for w in range(NW):
            a_ = SURV_BASE + 4 * CAP * w
            emu.s[w, S_OUT.n], emu.s[w, S_OUT.n + 1] = a_ & 0xffffffff, a_ >> 32
        emu.v[g.V_RAW.n] = np.zeros(NTHR, np.float32).view(np.uint32)
    tt = np.arange(NTHR)
    if kind == "core":
        for r in range(16):
            nloc = 1024 * r + tt
            a = a0 + 2 * nloc
            ldsf[(r * 1024 + tt) * 2], ldsf[(r * 1024 + tt) * 2 + 1] = x[a], x[a + 1]
        l, w = tt & 63, tt >> 6
        nloc = 1024 * np.minimum(l, 15) + 64 * w
        ldsf[LDS_GC // 4 + tt] = np.where(l < 16, x[a0 + 2 * nloc - 1], 0)
    emu.run(P)
    self_test.emu, self_test.gen = emu, g
    if kind == "core":
        for k in range(HS):
            out[2048 * k + 2 * tt], out[2048 * k + 2 * tt + 1] = ldsf[(k * 1024 + tt) * 2], ldsf[(k * 1024 + tt) * 2 + 1]
    xd = x.astype(np.float64)
    nz = (n_frames - a0) // 2 - 1
    z = xd[a0 - 1:a0 - 1 + 2 * nz:2] + 1j * xd[a0:a0 + 2 * nz:2]
    full = np.convolve(z, gt)
    Lg = len(gt)
    want = np.zeros(BLK * nblk)
    dry = np.zeros(BLK * nblk)
    for b in range(nblk):
        for k in range(HS):
            j = b * 1024 * HS + 1024 * (R0 + k) + np.arange(1024)
            y = full[j + Lg // 2]
            dq, dc = xd[a0 + 2 * j], xd[a0 + 2 * j + 1]
            pos = b * BLK + 2048 * k + 2 * np.arange(1024)
            if MODE == MODE_SWEEP:
                want[pos], want[pos + 1] = y.real, y.imag
                dry[pos], dry[pos + 1] = dq, dc
            elif MODE == MODE_ROT:
                want[pos], want[pos + 1] = ca * dq + sa * y.real, ca * dc + sa * y.imag
            else:
                want[pos], want[pos + 1] = sa * y.real, sa * y.imag
    if MODE == MODE_SWEEP:
        err = 0.0
        for w in range(NW):
            end = (int(emu.s[w, S_OUT.n]) | (int(emu.s[w, S_OUT.n + 1]) << 32)) - (SURV_BASE + 4 * CAP * w)
            got = surv[CAP * w:CAP * w + end // 4].reshape(-1, 2)
            exp = []
            for b in range(nblk):
                for k in range(HS):
                    for half in range(2):
                        pos = b * BLK + 2048 * k + 2 * (64 * w + np.arange(64)) + half
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
                    pos = b * BLK + 2048 * k + 2 * lanes
                    rexp = np.maximum(rexp, np.maximum(np.abs(dry[pos]), np.abs(dry[pos + 1])))
            assert np.array_equal(raw, rexp.astype(np.float32)), "running max |x|"
        kind = kind + " sweep"
    else:
        err = np.abs(out[:BLK * nblk] - want).max()
    st = P.stats()
    npk = sum(v for k, v in st.items() if k.startswith("pk_"))
    nval = sum(v for k, v in st.items() if k.startswith("v_"))
    if verbose:
        print("16k %s HS %2d mode %d: max err %.2e over %d blocks; VGPRs %d (temporaries %d pairs); loop+prologue: %d packed, %d other VALU, "
              "%d ds, %d vmem, %d waits, %d barriers" % (kind, HS, MODE, err, nblk, g.n_vgpr, len(g.TMP), npk, nval,
                                                        sum(v for k, v in st.items() if k.startswith("ds_")),
                                                        st.get("gload2", 0) + st.get("gload1", 0) + st.get("gstore2", 0), st.get("s_waitcnt", 0), st.get("s_barrier", 0)))
    assert err < 1e-6, err
    return g, P


VARIANTS = [(8, MODE_HILBERT), (8, MODE_ROT), (12, MODE_HILBERT), (12, MODE_ROT)]
SWEEP_HS = (8, 12)   # HS = 12: CLI blksiz 8192 (4096 odd taps = 4 context rows), 12 of 16 rows new where the 8192-point block has 8 of 16


def emit(path, opts=()):
    out = ["// GENERATED by tools/gen/gen_olasave16k.py -- do not edit.  The steady-state loop of olasave16k_kernel (16384-point blocks, 1024",
           "// threads) as gfx950 assembly with hand-allocated registers; the emulator (tools/gen/isa.py) has run every variant",
           "// against the float64 convolution (tests/test_gen_olasave.py).",
           "// operands: %0 in (aligned pair of local sample 0 of the first block), %1 out (first output frame), %2 blocks, %3 LDS byte",
           "// offset of the exchange area, %4 constants table, %5 ca, %6 sa * gain, %7 thread id",
           "#pragma once", ""]
    nv = {}

    def body(g, P, head, tail=()):
        for n, (lo, hi) in sorted(g.constants().items()):
            head.append("s_mov_b32 s%d, %r" % (n, Lit(float(lo))))
            head.append("s_mov_b32 s%d, %r" % (n + 1, Lit(float(hi))))
        lines = head + P.text(label_suffix="%=") + list(tail)
        for l in lines:
            if l.startswith(";"):
                continue
            out.append('\t"%s\\n\t" \\' % l)
        out.append('\t""')
        out.append("")
    for kind in ("run", "core"):
        for HS, MODE in VARIANTS:
            g = Gen16k(HS, MODE, opts, kind)
            P = g.build()
            nv[HS] = max(nv.get(HS, 0), g.n_vgpr)
            st = P.stats()
            out.append("// %s, HS %d, mode %d: %d VGPRs, %d instructions" % (kind, HS, MODE, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label", "probe"))))
            out.append("#define PROT_SV16_%s_%d_%d \\" % ("ASM" if kind == "run" else "CORE", HS, MODE))
            body(g, P, ["s_mov_b64 %r, %%0" % S_IN, "s_mov_b64 %r, %%1" % S_OUT, "s_mov_b32 %r, %%2" % S_NBLK, "s_mov_b32 %r, %%3" % S_LDS,
                        "s_mov_b64 %r, %%4" % S_SV, "s_mov_b32 %r, %%5" % S_ROT.lo, "s_mov_b32 %r, %%6" % S_ROT.hi, "v_mov_b32 %r, %%7" % g.TMP[0].lo,
                        "s_waitcnt vmcnt(0) lgkmcnt(0)"])
    # MODE_SWEEP: operands as prot_olasave8k_asm.inc's PROT_SV8_SWEEP_*
    for HS in SWEEP_HS:
        g = Gen16k(HS, MODE_SWEEP, opts, "run")
        P = g.build()
        nv[HS] = max(nv.get(HS, 0), g.n_vgpr)
        st = P.stats()
        out.append("// sweep, HS %d: %d VGPRs, %d instructions" % (HS, g.n_vgpr, sum(v for k, v in st.items() if k not in ("comment", "label", "probe"))))
        out.append("#define PROT_SV16_SWEEP_%d \\" % HS)
        body(g, P, ["s_mov_b64 %r, %%2" % S_IN, "s_mov_b64 %r, %%9" % S_OUT, "s_mov_b32 %r, %%3" % S_NBLK, "s_mov_b32 %r, %%4" % S_LDS,
                    "s_mov_b64 %r, %%5" % S_SV, "s_mov_b32 %r, %%6" % S_ROT.lo, "s_mov_b32 %r, %%7" % S_ROT.hi, "v_mov_b32 %r, %%8" % g.TMP[0].lo,
                    "v_mov_b32 %r, %%10" % g.V_RAW, "s_waitcnt vmcnt(0) lgkmcnt(0)"],
             ["s_mov_b64 %%0, %r" % S_OUT, "v_mov_b32 %%1, %r" % g.V_RAW])
    for HS in sorted(nv):
        clob = ['"memory"', '"vcc"', '"scc"'] + ['"s%d"' % i for i in SGPR_CLOBBER] + ['"v%d"' % i for i in range(0, nv[HS])]
        out.append("#define PROT_SV16_ASM_CLOBBERS_%d " % HS + ", ".join(clob))
        if HS in SWEEP_HS:
            out.append("#define PROT_SV16_SWEEP_CLOBBERS_%d PROT_SV16_ASM_CLOBBERS_%d, \"v120\"" % (HS, HS))
    out.append("")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "..", "phaserotate.lv2_amd", "csrc")
    for HS, MODE in VARIANTS:
        self_test(HS, MODE, nblk=2)
        self_test(HS, MODE, kind="core")
    for HS in SWEEP_HS:
        self_test(HS, MODE_SWEEP, nblk=2)
    if "--check" not in sys.argv:
        dst = os.path.join(csrc, "prot_olasave16k_asm.inc")
        emit(dst)
        print("wrote", os.path.normpath(dst))
