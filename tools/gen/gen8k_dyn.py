#!/usr/bin/env python3
"""Gen8kDyn: Gen8k's block with the workgroup's NEXT block drawn from a per-XCD counter (round 5: the shipped headline loop),
and dyn_claims() = its claim logic restated for the self-test.  Emitted and self-tested by gen_olasave8k.py."""
import math  # noqa: F401
import os
import sys

import numpy as np  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import Prog, Emu, V, VP, S, SP, Lit  # noqa: E402,F401
import gen_olasave as G4  # noqa: E402,F401
from gen_olasave import Gen, S_IN, S_OUT, S_NBLK, S_LDS, S_SV, S_ROT, S_C32, S_RR, S_W1, S_W3, S_W9, S_T, S_A0, MODE_ROT, MODE_HILBERT, MODE_SWEEP  # noqa: E402,F401
from gen8k_core import *  # noqa: E402,F401,F403
from gen8k_core import Gen8k  # noqa: E402

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
