"""GPU parity of the 360-angle minimum-peak sweep (the CLI's analysis pass, src/phase-rotate.cc:347-395 and 540-640 restated in
oracle/): peak tables against the oracle (1e-3 dB, angle index exact), pruned against unpruned and fused against streamed
bit for bit, shards against the single pass.  Tolerances as in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

import orc
from conftest import harmonic_stack, xorshift_uniform

pytestmark = pytest.mark.gpu

TOL = 2e-6
TOL_RAMP = 1e-5
TOL_DB = 1e-3


def db(x):
    return 20.0 * np.log10(np.maximum(np.asarray(x, np.float64), 1e-30))


def _signal(n, n_chn, seed):
    chans = []
    for c in range(n_chn):
        h = harmonic_stack(n, f0=110.0 * (1 + 0.31 * c))
        chans.append(0.85 * h + 0.1 * xorshift_uniform(n, seed=seed + c))
    return np.stack(chans, axis=1).astype(np.float32).reshape(-1)  # interleaved


@pytest.mark.parametrize("blksiz,n_chn,nblk_frac", [(2048, 1, 10.0), (2048, 2, 7.37), (1024, 1, 9.5),
                                                    (8192, 2, 4.2), (16384, 1, 3.5), (16384, 2, 30.3), (32768, 1, 6.5),
                                                    (4096, 2, 41.7)])
def test_sweep_full_table(pkg, blksiz, n_chn, nblk_frac):
    n = int(blksiz * nblk_frac)
    data = _signal(n, n_chn, seed=11)
    g = pkg.PhaseRotate(blksiz, n_chn)
    g.analyze_file(data, 0, 360, 1)
    o = orc.CliEngine(blksiz, n_chn)
    o.analyze_file(data, 0, 360, 1)
    pg, po = g.peaks(), o.peaks()
    assert np.abs(db(pg) - db(po)).max() < TOL_DB
    assert pg[:, 0].tolist() == po[:, 0].tolist()  # angle 0 is an exact abs-max (Q4)
    for c in range(n_chn):
        assert int(np.argmin(pg[c])) == int(np.argmin(po[c]))
    g.close()


def test_sweep_block_by_block_matches_oracle(pkg):
    """PhaseRotate::analyze call for call, incl. start=True on the first block (Q1),
    stride 24 (Q5) and a single-channel fine pass with negative angles."""
    P, n_chn, nblk = 2048, 2, 6
    data = _signal(P * nblk, n_chn, seed=21).reshape(nblk, P * n_chn)
    g = pkg.PhaseRotate(P, n_chn)
    o = orc.CliEngine(P, n_chn)
    for b in range(nblk):
        g.analyze(data[b], 0, 360, 24, -1, b == 0)
        o.analyze(data[b], 0, 360, 24, -1, b == 0)
    assert np.abs(db(g.peaks()) - db(o.peaks())).max() < TOL_DB
    for a in range(0, 360, 24):
        assert g.peak(-1, a) == pytest.approx(o.peaks()[:, a].max(), rel=1e-6)
    g.reset()
    o.reset()
    for b in range(nblk):
        g.analyze(data[b], -12, 13, 1, 1, b == 0)
        o.analyze(data[b], -12, 13, 1, 1, b == 0)
    pg, po = g.peaks(), o.peaks()
    assert np.abs(db(pg) - db(po)).max() < TOL_DB
    assert np.all(pg[0] == 0)  # channel 0 untouched by a chn=1 pass
    g.close()


def test_sweep_mid_stream_raw_peak_is_complete_call_by_call(pkg):
    """Angle 0 (the raw peak, calc_peak over the whole new block: cli/phase-rotate.cc:413-414) rides along in the sweep's
    loaders, which see x[t - P/2]: the last P/2 frames of a call that is not the stream's last must still be in
    peaks[c][0] when the call returns -- the loudest frame of every block sits in its second half here."""
    P, n_chn, nblk = 2048, 2, 5
    data = (0.05 * _signal(P * nblk, n_chn, seed=23)).reshape(nblk, P, n_chn).copy()
    for b in range(nblk):
        data[b, P - 3 - b, b % n_chn] = 0.1 + 0.15 * b      # in the last P/2 frames of block b
    g = pkg.PhaseRotate(P, n_chn)
    o = orc.CliEngine(P, n_chn)
    for b in range(nblk):
        g.analyze(data[b].reshape(-1), 0, 360, 24, -1, b == 0)
        o.analyze(data[b].reshape(-1), 0, 360, 24, -1, b == 0)
        pg, po = g.peaks(), o.peaks()
        assert pg[:, 0].tobytes() == po[:, 0].astype(np.float32).tobytes(), b
        assert pg[b % n_chn, 0] == np.float32(0.1 + 0.15 * b)
    # ... and the same through the angle-0-only route and a mixed sequence of the two
    g.reset()
    o.reset()
    for b in range(nblk):
        a1, st = (1, 24) if b % 2 else (360, 24)
        g.analyze(data[b].reshape(-1), 0, a1, st, -1, b == 0)
        o.analyze(data[b].reshape(-1), 0, a1, st, -1, b == 0)
        assert g.peaks()[:, 0].tobytes() == o.peaks()[:, 0].astype(np.float32).tobytes(), b
    g.close()


def test_sweep_first_block_quirk_q1(pkg):
    """An early impulse: dry contributions of t < blksiz/2 never enter a rotated peak."""
    P = 2048
    x = np.zeros(P * 3, np.float32)
    x[5] = 0.9
    g = pkg.PhaseRotate(P, 1)
    g.analyze_file(x, 0, 360, 1)
    pg = g.peaks()[0]
    o = orc.CliEngine(P, 1)
    o.analyze_file(x, 0, 360, 1)
    po = o.peaks()[0]
    assert pg[0] == np.float32(0.9)
    assert db(pg[2]) < -35.0  # ~ -40 dB at 1 deg
    assert np.abs(db(pg[1:]) - db(po[1:])).max() < 0.01
    g.close()


def test_sweep_empty_and_tiny_files(pkg):
    P = 2048
    g = pkg.PhaseRotate(P, 1)
    g.analyze_file(np.zeros(0, np.float32), 0, 360, 24)
    assert np.all(g.peaks() == 0)
    g.reset()
    x = xorshift_uniform(100, seed=2)
    g.analyze_file(x, 0, 360, 1)
    o = orc.CliEngine(P, 1)
    o.analyze_file(x, 0, 360, 1)
    assert np.abs(g.peaks() - o.peaks()).max() < 1e-6
    g.close()


def test_full_size_sweep_shards_equal_single_pass(pkg):
    """The multi-GPU sweep plumbing on ONE device: two engines sweep the two halves of a
    stereo stream (second one primed with the history in front of its shard), the tables
    are MAX-merged -- bit-identical to the single-engine table."""
    import torch
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    P, n_chn = 2048, 2
    nb = 3001
    n = nb * P - 777  # ragged last block
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.rand((n_chn, n), generator=g, device=dev) - 0.5) * 0.9
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    one.sync()
    # shard boundaries on whole device transform blocks (6144 frames = 3 blocks for this engine)
    plan = sweep_dist.plan_shards(n, P, 2, align=one.shard_align_blocks())
    assert all(b0 % one.shard_align_blocks() == 0 for b0, _ in plan)
    pk2 = torch.zeros((n_chn, 360), device=dev)
    for r, (b0, b1) in enumerate(plan):
        f0, nf = sweep_dist.shard_frames(n, P, plan, r)
        eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
        H = eng.info.history
        if r > 0:
            halo = x[:, f0 - H:f0].contiguous()
            torch.cuda.synchronize()
            keep = sweep_dist.prime_history(eng, halo)
        xs = x[:, f0:f0 + nf].contiguous()
        part = torch.zeros((n_chn, 360), device=dev)
        torch.cuda.synchronize()
        flags = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == len(plan) - 1 else 0)
        eng.sweep_dev(xs.data_ptr(), nf, nf, 0, 360, 1, -1, part.data_ptr(), flags)
        eng.sync()
        pk2 = torch.maximum(pk2, part)
        eng.close()
    assert torch.equal(pk1, pk2)
    one.close()


def _sweep_table(pkg, x, P, stride, prune, passes=((0, 360, 24), (100, 140, 1)), fused=False):
    """Peak table [n_chn][360] after a coarse and a fine pass over the interleaved file x, and the
    fraction of tile groups the sweep actually evaluated.  fused=False keeps the separate Hilbert and sweep
    passes (whose statistics count tiles of ALL frames) also where the engine would fuse them."""
    pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0 if prune else 1)
    pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, 0 if fused else 1)
    try:
        pr = pkg.PhaseRotate(P, x.shape[1])
        for a0, a1, st in passes:
            pr.analyze_file(x, a0, a1, st)
        tab = np.array([[pr.peak(c, a) for a in range(360)] for c in range(x.shape[1])], np.float32)
        ev, seen = pr.sweep_stats()
        _sweep_table.last_stats = pr.sweep_stats_ex()
        _sweep_table.last_fused = pr.sweep_stats_fused()
        pr.close()
    finally:
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
        pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, 0)
    return tab, (ev / seen if seen else None)  # the unpruned kernel keeps no statistics: seen == 0


@pytest.mark.parametrize("signal", ["noise+harmonics", "steady-tone", "silence-then-burst", "long-multichunk", "constant",
                                    "stationary-noise", "tone-then-noise", "bursts"])
def test_pruned_sweep_is_bit_identical_to_unpruned(pkg, signal):
    """The sweep tests every frame's r = sqrt(d^2 + h^2) against a lower bound of every final peak and
    evaluates only compacted tiles of survivors.  That must not change a single bit of the table --
    including when nothing can be skipped (a steady tone: every frame's r equals every angle's peak; a
    GROWING signal is no worst case on a parallel machine -- the waves working on the loud end publish
    their bound to those working on the quiet start), when the bound starts
    at zero (leading silence), across chunks (> 2^22 frames: later chunks start from the table of the
    earlier ones) and across passes.  The evaluated fraction is asserted too, both ways: a test in which
    nothing was skipped would compare the unpruned sweep with itself (that happened with the first,
    per-wave version of the bound); the steady tone must NOT be skippable and must go through the
    untested feed; long stationary noise -- the case the group-granular pruning of rounds 1-2 could not
    touch -- must stay below one half; and a wave that switched the test off on a tone must find its
    way back once the material turns quiet (two-way switch: "bursts" inside one launch, "tone-then-noise"
    across chunks)."""
    P = 2048
    rng = np.random.default_rng(11)
    if signal == "noise+harmonics":
        n = 600000
        x = np.stack([0.5 * harmonic_stack(n, 48000) + 0.1 * xorshift_uniform(n, seed=3), xorshift_uniform(n, seed=4)], 1)
    elif signal == "steady-tone":
        # constant envelope: r(t) = A for every frame and every angle's peak tends to A -- nothing is skippable
        n = 400000
        x = (0.7 * np.sin(2 * np.pi * 997.0 / 48000 * np.arange(n)))[:, None]
    elif signal == "tone-then-noise":
        # 2^22 frames of the steady tone (test switched off), then 3 x 2^22 of noise below it
        n = (1 << 22) * 4
        x = np.empty((n, 1), np.float32)
        g = 1 << 22
        x[:g, 0] = 0.7 * np.sin(2 * np.pi * 997.0 / 48000 * np.arange(g))
        x[g:, 0] = 0.5 * xorshift_uniform(n - g, seed=12)
    elif signal == "bursts":
        # ONE launch (a device chunk holds 2^22 frames; a workgroup walks it in steps of 1024 frames, grid-strided, so it
        # meets the material in file order): a quarter of steady tone -- all of it survives, the test goes off --, then
        # three quarters of quiet noise, where the wave must switch the test back on
        n, g = 1 << 22, 1 << 20
        x = np.empty((n, 1), np.float32)
        x[:g, 0] = 0.7 * np.sin(2 * np.pi * 997.0 / 48000 * np.arange(g))
        x[g:, 0] = 0.2 * xorshift_uniform(n - g, seed=30)
    elif signal == "silence-then-burst":
        n = 300000
        x = np.zeros((n, 2), np.float32)
        x[200000:200700, 0] = xorshift_uniform(700, seed=6)
        x[299990:, 1] = 0.9
    elif signal == "long-multichunk":
        # four device chunks (2^22 frames each): loud material in the first, 40 dB quieter afterwards --
        # the later chunks start from the bound the first one left in the table and are skipped -- plus
        # one late click louder than everything, which must still land in the table
        n = (1 << 22) * 3 + 12345
        x = (0.8 * harmonic_stack(n, 48000) + 0.05 * xorshift_uniform(n, seed=8))[:, None].astype(np.float32)
        x[1 << 22:] *= 0.01
        x[n - 3000:n - 2990] = 0.97
    elif signal == "stationary-noise":
        n = (1 << 22) * 3 + 777
        x = np.stack([0.9 * xorshift_uniform(n, seed=9), 0.9 * xorshift_uniform(n, seed=10)], 1)
    else:
        n = 100000
        x = np.full((n, 1), 0.25, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    a, frac_pruned = _sweep_table(pkg, x, P, 24, prune=True)
    st = dict(_sweep_table.last_stats)
    b, frac_plain = _sweep_table(pkg, x, P, 24, prune=False)
    assert a.tobytes() == b.tobytes()
    # FUSED (round 4): the Hilbert pass lists the frames above the bound of the whole call, the matrix-core kernel evaluates the
    # lists -- the same table bit for bit; chunks of 2^20 frames and more take that route, the others the separate passes
    f, _ = _sweep_table(pkg, x, P, 24, prune=True, fused=True)
    fu = dict(_sweep_table.last_fused)
    assert f.tobytes() == b.tobytes()
    assert (fu["tested"] > 0) == (x.shape[0] >= (1 << 20)), fu
    if fu["tested"]:
        print("\n[fused sweep, %s] %.1f %% of the tested frames listed" % (signal, 100.0 * fu["listed"] / fu["tested"]))
        assert fu["listed"] <= fu["tested"]
        if signal == "long-multichunk":
            assert fu["listed"] < 0.3 * fu["tested"], fu   # the quiet chunks are listed against the loud one's table
        if signal == "stationary-noise":
            assert fu["listed"] < 0.6 * fu["tested"], fu
    assert a.max() > 0            # every signal here has energy
    assert frac_plain is None and frac_pruned is not None  # the switch really selects the two kernels
    assert st["evaluated"] <= st["seen"] + 2 * 4 * 4096 and st["untested"] <= st["evaluated"]  # (+ zero-padded last tiles)
    print("\n[sweep pruning, %s] evaluated %.1f %% of the tiles, %.1f %% fed untested" %
          (signal, 100 * frac_pruned, 100.0 * st["untested"] / st["seen"]))
    # ... and the comparison is not vacuous: pruning must have skipped work where the signal allows it
    # (no expectation for "constant": its Hilbert branch is ~0, so the peaks near 90 degrees are ~0 and
    # little is skippable)
    if signal == "silence-then-burst":
        assert frac_pruned < 0.1, frac_pruned   # the silences cannot raise anything
    if signal == "long-multichunk":
        assert frac_pruned < 0.3, frac_pruned   # a part of the loud quarter is evaluated, the rest is not
    if signal == "stationary-noise":
        assert frac_pruned < 0.5, frac_pruned
        assert st["untested"] < 0.2 * st["seen"], st
    if signal == "growing":
        assert frac_pruned > 0.9, frac_pruned   # every frame holds a new record: nothing may be skipped
        assert st["untested"] > 0.5 * st["seen"], st  # ... and the test is switched off most of the time
    if signal == "bursts":
        assert 0.2 < frac_pruned < 0.85, frac_pruned  # (the tone quarter + the pauses that reach into the noise)
        assert 0.05 * st["seen"] < st["untested"] < 0.6 * st["seen"], st  # off in the tone, back on in the noise
    if signal == "growing-then-steady":
        assert frac_pruned < 0.6, frac_pruned   # the steady 3/4 are tested again (and mostly skipped)
        assert 0.1 * st["seen"] < st["untested"] < 0.4 * st["seen"], st


@pytest.mark.parametrize("stride,blocks", [(1, 12), (3, 4), (4, 3), (5, 3), (6, 2), (8, 2), (24, 1)])
def test_pruned_sweep_every_angle_block_layout(pkg, stride, blocks):
    """The sweep kernel spreads 32-angle blocks over four wave groups; 65..96 angles (3 blocks) leave one
    group without a block -- its waves must neither evaluate nor publish a bound --, 33..64 use two groups,
    <= 32 one.  Pruned and unpruned tables must be bit-identical in every layout, and equal to the oracle's
    within the gate."""
    from phaserotate_lv2_amd import sweep_dist
    P, n = 2048, 300000
    x = np.ascontiguousarray(np.stack([0.6 * harmonic_stack(n, 48000) + 0.05 * xorshift_uniform(n, seed=21),
                                       0.3 * xorshift_uniform(n, seed=22)], 1), np.float32)
    x[150000:, :] *= 0.05  # something to skip
    swept = [k for k in sweep_dist.angle_list(0, 360, stride) if k != 0]  # angle 0 is the raw peak, not swept
    assert (len(swept) + 31) // 32 == blocks
    passes = ((0, 360, stride),)
    a, frac = _sweep_table(pkg, x, P, stride, prune=True, passes=passes)
    b, _ = _sweep_table(pkg, x, P, stride, prune=False, passes=passes)
    assert a.tobytes() == b.tobytes()
    assert (a[:, swept] > 0).all()
    o = orc.CliEngine(P, 2)
    o.analyze_file(x, 0, 360, stride)
    ref = o.peaks()
    d_db = np.abs(20 * np.log10(a[:, swept] / ref[:, swept]))
    assert d_db.max() < 1e-3, d_db.max()


@pytest.mark.parametrize("n_chn,n,passes", [
    (3, 100000, ((0, 360, 24),)),                 # three channels: the scratch table's channel stride, 15 angles (one block)
    (1, 700, ((0, 360, 1),)),                     # less than one 1024-frame step, all angles
    (4, 5000, ((0, 1, 24),)),                     # angle 0 only: no sweep launch, the stand-alone abs-max kernel
    (2, 2048 * 3 + 1, ((0, 360, 8), (200, 236, 1))),  # a ragged last block, a second pass over a few angles without angle 0
    (5, 40000, ((1, 360, 7),)),                   # five channels, angle 0 NOT asked for (slot 0 must stay untouched)
    (33, 9000, ((0, 360, 1),)),                   # more channels than step counters per workgroup slot: fewer than 32 workgroups
    (64, 7000, ((0, 360, 3),)),                   #   per channel draw from as many counters as there are workgroups
    (1100, 2500, ((0, 360, 24),)),                # more channels than resident-workgroup slots: the scratch table in several launches
])
def test_sweep_table_small_and_many_channel_calls(pkg, n_chn, n, passes):
    """The sweep's per-launch scratch table and merge kernel (one value per angle, workgroup and wave; the raw peak of angle 0
    riding along in the loaders) at shapes the long-file tests do not reach: more channels than two, inputs shorter than
    one step, calls that ask for angle 0 alone or not at all.  Against the oracle (1e-3 dB, angle 0 exact)."""
    P = 2048
    x = np.ascontiguousarray(np.stack([0.5 * harmonic_stack(n, 48000) * (0.4 + 0.1 * c) + 0.02 * xorshift_uniform(n, seed=40 + c)
                                       for c in range(n_chn)], 1), np.float32)
    g = pkg.PhaseRotate(P, n_chn)
    o = orc.CliEngine(P, n_chn)
    asked = set()
    for a0, a1, st in passes:
        g.analyze_file(x, a0, a1, st)
        o.analyze_file(x, a0, a1, st)
        a = a0
        while a <= a1:           # the loop bounds of cli/phase-rotate.cc:409-428
            asked.add(((a % 360) + 360) % 360)
            a += st
            if a >= a1:
                break
    tab = np.array([[g.peak(c, a) for a in range(360)] for c in range(n_chn)], np.float32)
    ref = o.peaks()
    g.close()
    for a in range(360):
        if a not in asked:
            assert (tab[:, a] == 0).all(), a            # untouched slots stay untouched
    swept = sorted(a for a in asked if a != 0)
    if 0 in asked:
        assert tab[:, 0].tobytes() == ref[:, 0].astype(np.float32).tobytes()  # the raw peak: exact
        assert (tab[:, 0] > 0).all()
    if swept:
        assert (tab[:, swept] > 0).all()
        d_db = np.abs(20 * np.log10(tab[:, swept] / ref[:, swept]))
        assert d_db.max() < 1e-3, d_db.max()


@pytest.mark.parametrize("P,signal", [(16384, "harmonic"), (16384, "noise"), (32768, "harmonic"), (1024, "harmonic"), (1024, "noise"),
                                      (8192, "harmonic"), (8192, "noise"), (4096, "harmonic"), (4096, "noise")])
def test_fused_sweep_of_the_long_fir_engines_is_bit_identical(pkg, P, signal):
    """CLI blksiz 16384 (the reference's default at 96 kHz, cli/phase-rotate.cc:749-755) and 32768: the Hilbert branch takes two /
    four accumulating launches; on the fused route the LAST one tests the complete Hilbert value with its dry frame instead of
    storing it and lists the survivors (olapoly_kernel MODE_SWEEP); blksiz 1024: its one launch does; blksiz 8192 (the default at 44.1 /
    48 kHz): olasave8k_kernel<8, MODE_SWEEP>, the generated loop of the 8192-point blocks with the test in its epilogue.  Same table as the separate passes and as the unpruned
    kernel, bit for bit, over several device chunks (2^22 frames each), a coarse and a fine pass, stereo."""
    n = (1 << 22) * 2 + 54321
    if signal == "noise":
        x = np.stack([0.9 * xorshift_uniform(n, seed=21), 0.6 * xorshift_uniform(n, seed=22)], 1)
    else:
        x = np.stack([0.6 * harmonic_stack(n, 96000) + 0.03 * xorshift_uniform(n, seed=23),
                      0.5 * harmonic_stack(n, 96000)[::-1] + 0.03 * xorshift_uniform(n, seed=24)], 1)
        x[n // 3:n // 2] *= 0.1
    x = np.ascontiguousarray(x, np.float32)
    b, _ = _sweep_table(pkg, x, P, 24, prune=False)
    a, _ = _sweep_table(pkg, x, P, 24, prune=True)
    f, _ = _sweep_table(pkg, x, P, 24, prune=True, fused=True)
    fu = dict(_sweep_table.last_fused)
    assert a.tobytes() == b.tobytes()
    assert f.tobytes() == b.tobytes()
    assert fu["tested"] > 0 and 0 < fu["listed"] < 0.7 * fu["tested"], fu
    print("\n[fused sweep, blksiz %d, %s] %.1f %% of the tested frames listed" % (P, signal, 100.0 * fu["listed"] / fu["tested"]))


def test_fused_sweep_many_channels_and_single_channel_calls(pkg):
    """The fused route with few workgroups per channel (40 channels share the device: fewer step counters than the kernels'
    32 per channel are in use) and with single-channel calls on a multi-channel engine (chn >= 0: the lists, the bound and the
    table rows of ONE channel): both against the unpruned kernel, bit for bit."""
    n = (1 << 20) + 4321
    nc = 40
    x = np.stack([(0.2 + 0.02 * c) * xorshift_uniform(n, seed=300 + c) for c in range(nc)], 1).astype(np.float32)
    x[n // 2:n // 2 + 2000, 7] = 0.95 * np.sign(x[n // 2:n // 2 + 2000, 7])
    x = np.ascontiguousarray(x)
    b, _ = _sweep_table(pkg, x, 2048, 24, prune=False, passes=((0, 360, 3),))
    f, _ = _sweep_table(pkg, x, 2048, 24, prune=True, passes=((0, 360, 3),), fused=True)
    fu = dict(_sweep_table.last_fused)
    assert fu["tested"] == n // 2048 * 2048 * nc or fu["tested"] > 0, fu
    assert f.tobytes() == b.tobytes()
    # single-channel calls on a 3-channel engine
    x3 = np.ascontiguousarray(x[:, :3])
    tabs = {}
    for mode in ("unpruned", "fused"):
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1 if mode == "unpruned" else 0)
        try:
            pr = pkg.PhaseRotate(4096, 3)
            for c in (2, 0):
                pr.analyze_file(x3, 0, 360, 5, chn=c)
            tabs[mode] = np.array([[pr.peak(c, a) for a in range(360)] for c in range(3)], np.float32)
            if mode == "fused":
                assert pr.sweep_stats_fused()["tested"] > 0
            pr.close()
        finally:
            pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
    assert tabs["fused"].tobytes() == tabs["unpruned"].tobytes()
    assert tabs["fused"][1].max() == 0.0 and tabs["fused"][0].max() > 0 and tabs["fused"][2].max() > 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sweep_random_shapes_against_oracle_and_unpruned(pkg, seed):
    """Randomised: block size, channel count, length (1 frame ... 150 000), angle range and stride, signal shape per channel
    (noise, tone, growing tone, bursts in silence, level steps, DC), one or two passes.  The pruned table must equal the
    unpruned one bit for bit and the oracle's within 1e-3 dB, the raw peak (angle 0) exactly.  (tools/sweep_fuzz.py is the
    long-running pruned-vs-unpruned version of this.)"""
    rng = np.random.default_rng(seed)

    def signal(n, nc):
        t = np.arange(n)
        x = np.zeros((n, nc), np.float32)
        for c in range(nc):
            kind, amp = rng.integers(0, 6), rng.uniform(0.05, 0.9)
            if kind == 0:
                x[:, c] = amp * (rng.random(n) - 0.5)
            elif kind == 1:
                x[:, c] = amp * np.sin(2 * np.pi * rng.uniform(0.001, 0.2) * t)
            elif kind == 2:
                x[:, c] = amp * np.sin(2 * np.pi * rng.uniform(0.001, 0.05) * t) * np.linspace(0.1, 1, n)
            elif kind == 3:
                for _ in range(rng.integers(1, 6)):
                    a = rng.integers(0, n)
                    b = min(n, a + rng.integers(1, 5000))
                    x[a:b, c] = amp * (rng.random(b - a) - 0.5)
            elif kind == 4:
                lev = rng.uniform(0.01, 1.0, 8)
                x[:, c] = lev[np.minimum(t * 8 // max(n, 1), 7)] * 0.5 * np.sign(np.sin(2 * np.pi * 0.01 * t) + 1e-9) + 0.01 * (rng.random(n) - 0.5)
            else:
                x[:, c] = amp * 0.25
        return np.ascontiguousarray(x)

    for k in range(14):
        P = int(rng.choice([1024, 2048, 2048, 4096, 8192]))
        nc = int(rng.choice([1, 2, 2, 3]))
        n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 150000)]))
        st = int(rng.choice([1, 2, 3, 5, 8, 24]))
        a0 = int(rng.choice([0, 0, 1, rng.integers(0, 300)]))
        a1 = int(min(360, a0 + rng.choice([1, 40, 100, 360])))
        x = signal(n, nc)
        passes = ((a0, a1, st),) if k % 3 else ((a0, a1, st), (max(0, a0 - 7), min(360, a1 + 9), 1))
        a, _ = _sweep_table(pkg, x, P, st, prune=True, passes=passes)
        b, _ = _sweep_table(pkg, x, P, st, prune=False, passes=passes)
        assert a.tobytes() == b.tobytes(), (seed, k, P, nc, n, passes)
        o = orc.CliEngine(P, nc)
        for q0, q1, qs in passes:
            o.analyze_file(x, q0, q1, qs)
        ref = o.peaks().astype(np.float32)
        assert a[:, 0].tobytes() == ref[:, 0].tobytes(), (seed, k, "raw peak")
        nz = ref[:, 1:] > 1e-12
        assert ((a[:, 1:] > 0) == (ref[:, 1:] > 0)).all(), (seed, k, "touched slots")
        if nz.any():
            d_db = np.abs(20 * np.log10(a[:, 1:][nz] / ref[:, 1:][nz]))
            assert d_db.max() < 1e-3, (seed, k, P, nc, n, passes, float(d_db.max()))
