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
from isa import Prog, Emu, V, VP, S, SP, Lit  # noqa: E402,F401
import gen_olasave as G4  # noqa: E402,F401
from gen_olasave import Gen, S_IN, S_OUT, S_NBLK, S_LDS, S_SV, S_ROT, S_C32, S_RR, S_W1, S_W3, S_W9, S_T, S_A0, MODE_ROT, MODE_HILBERT, MODE_SWEEP  # noqa: E402,F401
from gen8k_core import *  # noqa: E402,F401,F403   (Gen8k, the LDS / SGPR plan, device_table)
from gen8k_dyn import *  # noqa: E402,F401,F403    (Gen8kDyn, dyn_claims)


# ---------------------------------------------------------------------------------------------------- self-test
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
