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


def test_model_of_the_short_chain_block():
    """tools/model/olasave512_model.py: the 512-thread block of prot_olasave512.hip thread by thread -- four radix-8 passes each way
    over the digits of n and k, the three exchanges' LDS slots, the constants as the host lays them out (tables()) -- against the
    convolution with g (and the radix-16 model) for the three block geometries"""
    import numpy as np
    import olasave_model as M
    import olasave512_model as M5
    rng = np.random.default_rng(3)
    for n_taps, hs in [(3072, 10), (4096, 8), (2048, 12)]:
        g = M.hilbert_fir(n_taps)[1::2]
        B = M.tables(g)
        tb = M5.tables(B)
        assert tb["fA"].shape == (7, 512) and tb["iB"].shape == (7, 64) and tb["rho"].shape == (512, 2)
        z = 0.3 * (rng.standard_normal(M.N) + 1j * rng.standard_normal(M.N))
        want = np.convolve(z, g)[len(g) // 2:len(g) // 2 + M.N]
        lo, hi = 256 * (16 - hs) // 2, M.N - 256 * (16 - hs) // 2
        y = M5.block(z, tb)
        assert np.abs(y - want)[lo:hi].max() < 1e-7
        assert np.abs(y - M.block(z, B, len(g), False, 1))[lo:hi].max() < 1e-12   # (the same approximation of |B|: equal to rounding)


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


@pytest.mark.parametrize("hs,mode,nb,xr,stride", [(13, 0, 6, 2, 3), (12, 2, 5, 7, 1), (8, 0, 3, 5, 2)])
def test_generated_dynamic_loop_transforms_the_blocks_its_tickets_name(hs, mode, nb, xr, stride):
    """Gen8kDyn (round 5: a workgroup claims its next block from a counter instead of walking a run): one workgroup in the emulator
    with the other workgroups' draws simulated by the stride of the counters -- the odd-context rotation at 48 kHz, regions that run
    out (the workgroup moves on to the next one, wraps around, and stops after eight), the scalar atomic's wait discipline, the LDS
    mailbox under the race detector.  Every claimed block against the float64 convolution, every other block untouched."""
    import gen_olasave8k as G8
    G8.self_test_dyn(hs, mode, nb=nb, xr=xr, stride=stride, verbose=False)


def test_dynamic_claim_model_covers_every_block_exactly_once():
    """the claim protocol restated in Python (dyn_claims): W workgroups drawing in any interleaving transform each block once"""
    import random
    import gen_olasave8k as G8
    for nb, W, seed in ((37, 5, 1), (8, 16, 2), (100, 64, 3), (3, 9, 4)):
        rnd = random.Random(seed)
        per = (nb + 7) // 8
        ctr = [0] * 8
        wgs = [{"xr": rnd.randrange(8), "tries": 8, "done": False} for _ in range(W)]
        got = []
        while not all(w["done"] for w in wgs):
            w = rnd.choice([w for w in wgs if not w["done"]])
            t = ctr[w["xr"]]
            ctr[w["xr"]] += 1
            while True:
                start = w["xr"] * per
                ln = 0 if start >= nb else min(per, nb - start)
                if t < ln:
                    got.append(start + t)
                    break
                w["tries"] -= 1
                if w["tries"] == 0:
                    w["done"] = True
                    break
                w["xr"] = (w["xr"] + 1) & 7
                t = ctr[w["xr"]]
                ctr[w["xr"]] += 1
        assert sorted(got) == list(range(nb)), (nb, W)
        assert G8.dyn_claims(nb, 3, 1)[0] == sorted(G8.dyn_claims(nb, 3, 1)[0], key=lambda b: ((b // per - 3) % 8, b))


def test_committed_8192_point_include_is_current(tmp_path):
    import gen_olasave8k as G8
    dst = tmp_path / "asm8k.inc"
    G8.emit(str(dst))
    cur = open(os.path.join(ROOT, "phaserotate.lv2_amd", "csrc", "prot_olasave8k_asm.inc")).read()
    assert open(dst).read() == cur, "run python3 tools/gen/gen_olasave8k.py"


@pytest.mark.parametrize("mode,kind", [(2, "run"), (0, "core"), (3, "run")])
def test_generated_16384_point_block_matches_float64_convolution(mode, kind):
    """tools/gen/gen_olasave16k.py (round 5: 16384-point blocks, 1024 threads, a radix-4 layer taken by the readers of the first and
    last LDS exchange -- four code variants by the wave's ke --, its twiddle W64^{b ke} from the W32 constants and one more): the CLI
    blksiz 16384 engine's Hilbert loop, a staged block with rotation, and the sweep variant (sixteen waves' survivor lists against
    the float64 convolution), in the emulator."""
    import gen_olasave16k as G16
    G16.self_test(8, mode, nblk=2, kind=kind, verbose=False)


def test_committed_16384_point_include_is_current(tmp_path):
    import gen_olasave16k as G16
    dst = tmp_path / "asm16k.inc"
    G16.emit(str(dst))
    cur = open(os.path.join(ROOT, "phaserotate.lv2_amd", "csrc", "prot_olasave16k_asm.inc")).read()
    assert open(dst).read() == cur, "run python3 tools/gen/gen_olasave16k.py"


def test_model_of_the_16384_point_block():
    """tools/model/olasave_model16k.py: the radix-4 layer's index arithmetic against numpy's FFT, the table-free block against the
    direct convolution with the 8192 odd taps of the CLI's default filter at 96 kHz"""
    import numpy as np
    import olasave_model16k as M
    rng = np.random.default_rng(3)
    w = rng.standard_normal(M.N) + 1j * rng.standard_normal(M.N)
    Z = np.fft.fft(w) * M.W(-np.arange(M.N), 2 * M.N)
    assert np.abs(M.fwd_shifted(w) - Z[M.kindex()]).max() < 1e-8
    g = M.hilbert_fir(16384)[1::2]
    B = M.tables(g)
    z = 0.3 * (rng.standard_normal(M.N) + 1j * rng.standard_normal(M.N))
    want = np.convolve(z, g)[len(g) // 2:len(g) // 2 + M.N]
    assert np.abs(M.block(z, B) - want)[1024 * 4:1024 * 12].max() < 1e-7
