"""The generated loop of olasave_kernel (tools/gen/gen_olasave.py -> csrc/prot_olasave_asm.inc) on the CPU: every variant runs
in the instruction emulator (tools/gen/isa.py: packed-fp32 semantics, LDS, the s_waitcnt discipline, the DPP and SGPR hazards
found on the hardware) against the float64 convolution of the stream with the reference's odd taps, and the committed include
file is what the generator writes today."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "gen"))
sys.path.insert(0, os.path.join(ROOT, "tools", "model"))


@pytest.mark.parametrize("hs", [8, 10, 12])
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("kind", ["run", "core"])
def test_generated_block_matches_float64_convolution(hs, mode, kind):
    import gen_olasave as G
    G.self_test(hs, mode, 1, nblk=2, kind=kind, verbose=False)  # asserts max error < 1e-6 and a clean wait / hazard record


@pytest.mark.parametrize("thr2", [0.30, 0.75, 50.0])
def test_generated_sweep_block_lists_the_survivors(thr2):
    """MODE_SWEEP of the 4096-point loop (blksiz 2048's fused pass): the four waves' survivor lists and raw maxima against the
    float64 convolution, with one frame in seven surviving (every row appends), one in five hundred (most rows take the branch
    around the appends, some do not) and none"""
    import gen_olasave as G
    G.self_test(12, 3, 1, nblk=2, kind="run", verbose=False, thr2=thr2)


def test_model_of_the_table_free_form():
    """tools/model/olasave_model.py: the half-shifted transform pair with |B| = 1 outside the bins next to DC reproduces the
    convolution with g to 1e-7 for every FIR the engines use"""
    import numpy as np
    import olasave_model as M
    rng = np.random.default_rng(1)
    for n_taps, hs in [(3072, 10), (4096, 8), (2048, 12)]:
        g = M.hilbert_fir(n_taps)[1::2]
        B = M.tables(g)
        z = 0.3 * (rng.standard_normal(M.N) + 1j * rng.standard_normal(M.N))
        want = np.convolve(z, g)[len(g) // 2:len(g) // 2 + M.N]
        lo, hi = 256 * (16 - hs) // 2, M.N - 256 * (16 - hs) // 2
        assert np.abs(M.block(z, B, len(g), False, 1) - want)[lo:hi].max() < 1e-7


def test_committed_include_is_current(tmp_path):
    import gen_olasave as G
    dst = tmp_path / "asm.inc"
    G.emit(str(dst))
    cur = open(os.path.join(ROOT, "phaserotate.lv2_amd", "csrc", "prot_olasave_asm.inc")).read()
    assert open(dst).read() == cur, "run python3 tools/gen/gen_olasave.py"


@pytest.mark.parametrize("hs,mode,kind", [(13, 0, "run"), (8, 3, "run"), (8, 2, "core")])
def test_generated_8192_point_block_matches_float64_convolution(hs, mode, kind):
    """tools/gen/gen_olasave8k.py (round 4: 8192-point blocks, 512 threads, the radix-2 layer taken by the readers of the LDS
    exchanges): the LV2 48 kHz loop with its half-row rotation, the CLI blksiz 8192 sweep variant (survivor lists of the eight waves
    against the float64 convolution) and a staged block, in the emulator -- per-wave control flow, the LDS race detector, the
    lane-read and lane-mask hazards found on the hardware.  (All variants: python3 tools/gen/gen_olasave8k.py, ~10 minutes.)"""
    import gen_olasave8k as G8
    G8.self_test(hs, mode, nblk=2, kind=kind, verbose=False)
    if mode == 3:   # ... and with few survivors: rows that skip their appends
        G8.self_test(hs, mode, nblk=2, kind=kind, verbose=False, thr2=0.7)


def test_committed_8192_point_include_is_current(tmp_path):
    import gen_olasave8k as G8
    dst = tmp_path / "asm8k.inc"
    G8.emit(str(dst))
    cur = open(os.path.join(ROOT, "phaserotate.lv2_amd", "csrc", "prot_olasave8k_asm.inc")).read()
    assert open(dst).read() == cur, "run python3 tools/gen/gen_olasave8k.py"
