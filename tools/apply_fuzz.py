#!/usr/bin/env python3
"""Randomised differential test of the short-chain block (prot_olasave512.hip) against the 256-thread block (PROT_TEST_NO_SHORT_CHAIN):
random engines (LV2 44.1 / 48 / 88.2 / 96 kHz, CLI blksiz 2048 / 4096), channel counts, lengths from 1 frame up, channel strides and
4-byte-aligned stream starts, stateless calls and stateful series of random pieces, steady rotation and per-frame ramps -- outputs
within 2e-6, nothing written beyond the requested frames.      python3 tools/apply_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad = 0
took = 0
for case in range(cases):
    kind, param = [(pkg.KIND_LV2, 44100), (pkg.KIND_LV2, 48000), (pkg.KIND_LV2, 88200), (pkg.KIND_LV2, 96000), (pkg.KIND_CLI, 2048), (pkg.KIND_CLI, 4096)][rng.integers(6)]
    chn = int(rng.integers(1, 5))
    n = int(rng.choice([1, 2, 3, 63, 64, 777, 1024, 4095, 4096, 5120, 5121, 20000, 100003, 300000]) if rng.random() < 0.5 else rng.integers(1, 400000))
    pad, lead = int(rng.integers(0, 9)), int(rng.integers(0, 4))          # channel stride n + pad; the streams start `lead` floats into the buffer
    stride = n + pad
    x = torch.rand(lead + chn * stride + 8, device=dev) - 0.5
    rot = rng.uniform(-1, 1, 2 * chn).astype(np.float32)
    stateful = rng.random() < 0.5
    ramp = rng.random() < 0.25
    pieces = [n]
    if stateful:
        cuts = sorted(set(int(v) for v in rng.integers(1, max(2, n), size=int(rng.integers(0, 6)))))
        pieces = [b - a for a, b in zip([0] + cuts, cuts + [n]) if b > a]
    outs = []
    for off_opt in (0, 1):
        pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, off_opt)
        e = pkg.Engine(kind, param, chn)
        y = torch.full_like(x, 7.0)
        pos = 0
        for m in pieces:
            xp, yp = x.data_ptr() + 4 * (lead + pos), y.data_ptr() + 4 * (lead + pos)
            if ramp:
                rs = torch.from_numpy(np.ascontiguousarray(np.stack([np.stack([np.cos(0.001 * (pos + np.arange(m)) + c), np.sin(0.001 * (pos + np.arange(m)) + c)], 1) for c in range(chn)]), np.float32)).to(dev)
                e.apply_dev_ramp(xp, stride, m, yp, stride, m, rs.data_ptr(), 0 if stateful else pkg.F_STATELESS)
            else:
                e.apply_dev(xp, stride, m, yp, stride, m, rot, 0 if stateful else pkg.F_STATELESS)
            pos += m
        e.sync()
        e.close()
        outs.append(y)
    pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, 0)
    d = float((outs[0] - outs[1]).abs().max())
    took += d > 0
    ok = d < 2e-6
    for o in outs:
        ok = ok and float((o[:lead] - 7.0).abs().max() if lead else 0.0) == 0.0
        for c in range(chn):
            tail = o[lead + c * stride + n:lead + (c + 1) * stride] if c + 1 < chn else o[lead + c * stride + n:]
            ok = ok and (tail.numel() == 0 or float((tail - 7.0).abs().max()) == 0.0)
    if not ok:
        bad += 1
        print("MISMATCH case %d: kind %d param %d chn %d n %d pad %d lead %d pieces %r ramp %d: max diff %.3e" % (case, kind, param, chn, n, pad, lead, pieces[:8], ramp, d))
print("%d cases (%d took the short-chain block), %d mismatches" % (cases, took, bad))
sys.exit(1 if bad else 0)
