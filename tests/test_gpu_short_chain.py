"""GPU parity of the SHORT-CHAIN block (round 6, csrc/prot_olasave512.hip): launches of fewer transform blocks than the device has
CUs take a 512-thread form of the 4096-point overlap-save block (8 points per thread, four radix-8 passes each way) -- another
fp32 factorisation of the same filter.  Against the 256-thread block on the same inputs (PROT_TEST_NO_SHORT_CHAIN: 2e-6) and
against the CPU oracle's run() (src/phaserotate.c:540-760 restated in oracle/): stateless and stateful, ragged lengths, history and
silence in reach, stereo with different rotations, odd strides, all three block geometries (8, 10, 12 new rows per block)."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

TOL = 2e-6


def _pair(pkg, make, call):
    """the same calls with and without the 512-thread block"""
    outs = []
    for off in (0, 1):
        pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, off)
        try:
            e = make()
            outs.append(call(e))
            e.sync()
            e.close()
        finally:
            pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, 0)
    return outs


@pytest.mark.parametrize("kind,param,chn,n,odd", [("lv2", 48000, 1, 5120 * 37 + 13, 0), ("lv2", 96000, 2, 256 * 1024, 0), ("lv2", 96000, 2, 100_003, 1),
                                                  ("lv2", 48000, 2, 1024, 0), ("lv2", 44100, 1, 777, 1), ("cli", 2048, 2, 6144 * 20 + 5, 0),
                                                  ("lv2", 48000, 4, 5120 * 40, 0)])
def test_short_chain_block_against_block256_and_oracle(pkg, kind, param, chn, n, odd):
    import torch
    dev = torch.device("cuda:0")
    stride = n + 2 + odd
    g = torch.Generator(device=dev).manual_seed(77 + chn)
    x = torch.rand(chn * stride, generator=g, device=dev) - 0.5
    rot = np.array([0.28, -0.96, -0.6, 0.8, 0.0, 1.0, 1.0, 0.0][:2 * chn], np.float32)
    K = pkg.KIND_LV2 if kind == "lv2" else pkg.KIND_CLI

    def call(e):
        y = torch.full_like(x, 7.0)
        e.apply_dev(x.data_ptr(), stride, n, y.data_ptr(), stride, n, rot, pkg.F_STATELESS)
        e.sync()
        return y
    y5, y2 = _pair(pkg, lambda: pkg.Engine(K, param, chn), call)
    d = float((y5 - y2).abs().max())
    assert 0.0 < d < TOL, d          # (another factorisation: equal to rounding, not to the bit -- if it is 0 the launch did not take the new block)
    for c in range(chn):             # nothing written beyond the requested frames
        assert float((y5[c * stride + n:(c + 1) * stride] - 7.0).abs().max()) == 0.0
    if kind == "lv2" and param in (48000, 96000):
        lat = {48000: 1792, 96000: 2560}[param]
        for c in range(min(chn, 2)):
            xs = x[c * stride:c * stride + n].cpu().numpy()[None, :]
            o = orc.Lv2Engine(param, 1)
            o.activate()
            ang = np.degrees(np.arctan2(-rot[2 * c + 1], rot[2 * c]))
            yo = o.run_stream(xs, float(ang), 1024)[0]
            skip = 2 * lat + 4096    # (Q10: the oracle instance ramps from 0 degrees on its first blocks)
            if n > skip + 64:
                got = y5[c * stride + skip:c * stride + n].cpu().numpy()
                assert np.abs(got - yo[skip:]).max() < TOL, c


@pytest.mark.parametrize("rate,chn,sizes", [(48000, 1, [1024] * 9 + [333, 4096, 64, 5120, 1000]), (96000, 2, [262144, 8192, 1024, 100_001, 4095])])
def test_short_chain_block_stateful_calls_carry_the_history(pkg, rate, chn, sizes):
    """stateful calls of every size (history in reach of the first block, silence in reach of the last, the next history written by the
    extra workgroups): call by call against the 256-thread block, the whole stream against one stateless call"""
    import torch
    dev = torch.device("cuda:0")
    n = sum(sizes)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((chn, n), generator=g, device=dev) - 0.5
    rot = np.array([0.6, 0.8, -0.8, 0.6][:2 * chn], np.float32)

    def call(e):
        y = torch.zeros_like(x)
        pos = 0
        for m in sizes:
            xs, ys = x[:, pos:], y[:, pos:]
            e.apply_dev(xs.data_ptr(), n, m, ys.data_ptr(), n, m, rot, 0)
            pos += m
        e.sync()
        return y
    y5, y2 = _pair(pkg, lambda: pkg.Engine(pkg.KIND_LV2, rate, chn), call)
    assert float((y5 - y2).abs().max()) < TOL
    e = pkg.Engine(pkg.KIND_LV2, rate, chn)
    y1 = torch.zeros_like(x)
    e.apply_dev(x.data_ptr(), n, n, y1.data_ptr(), n, n, rot, pkg.F_STATELESS)
    e.sync()
    e.close()
    assert float((y5 - y1).abs().max()) < TOL


def test_short_chain_block_randomised_differential():
    """tools/apply_fuzz.py (random engines, lengths from one frame, strides, 4-byte-aligned starts, stateful series, ramps): the
    512-thread block against the 256-thread block within 2e-6, nothing written beyond the requested frames -- sixty cases of one seed
    (750 of other seeds were run by hand in round 6)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "apply_fuzz.py"), "60", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout and "took the short-chain block" in r.stdout
