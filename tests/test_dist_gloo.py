"""CPU, world_size 2, gloo: frame-block sharding + MAX all-reduce of the peak
table reproduces the single-process table bit for bit (max is exact), and the
selection logic on the reduced table matches the oracle's restatement of main()."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import orc
from conftest import harmonic_stack, xorshift_uniform

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_shards(pkg):
    from phaserotate_lv2_amd import sweep_dist as sd
    assert sd.plan_shards(10 * 2048, 2048, 2, align=1) == [(0, 5), (5, 10)]
    assert sd.plan_shards(10 * 2048, 2048, 2, align=2) == [(0, 6), (6, 10)]  # pair-aligned boundaries
    assert sd.plan_shards(10 * 2048 + 1, 2048, 4, align=2) == [(0, 4), (4, 8), (8, 10), (10, 11)]
    assert sd.plan_shards(2048, 2048, 8, align=2) == [(0, 1)] + [(1, 1)] * 7
    assert sd.plan_shards(0, 2048, 2, align=2) == [(0, 0), (0, 0)]
    plan = sd.plan_shards(5 * 2048 + 100, 2048, 2, align=2)
    assert plan == [(0, 4), (4, 6)]
    assert sd.shard_frames(5 * 2048 + 100, 2048, plan, 1) == (4 * 2048, 1 * 2048 + 100)
    for n, w in ((345600000, 8), (12345678, 3)):
        pl = sd.plan_shards(n, 2048, w, align=3)
        assert pl[0][0] == 0 and pl[-1][1] == (n + 2047) // 2048
        assert all(a[1] == b[0] for a, b in zip(pl, pl[1:])) and all(b0 % 3 == 0 for b0, _ in pl)


def test_angle_shards_partition_the_angle_list(pkg):
    """sweep_dist.angle_shard: the ranks' (start, end, stride) triples visit every angle of the call's loop exactly once
    (cli/phase-rotate.cc:409-428: the loop visits `end` itself only when it starts there)."""
    from phaserotate_lv2_amd import sweep_dist as sd
    for ang in ((0, 360, 1), (0, 360, 24), (-12, 13, 1), (0, 0, 1), (5, 6, 1), (348, 372, 2), (-180, 180, 7)):
        visited = sd.angle_list(*ang)
        for world in (1, 2, 3, 5, 8, 16, 400):
            shares = [sd.angle_shard(ang[0], ang[1], ang[2], world, r) for r in range(world)]
            got = sorted(a for sh in shares if sh is not None for a in sd.angle_list(*sh))
            assert got == sorted(visited), (ang, world)
            assert shares[0] is not None


@pytest.mark.parametrize("n_chn,n_frames,mode", [(2, 2048 * 9 + 517, "one_phase"), (1, 2048 * 4, "one_phase"), (2, 2048 * 13 + 100, "two_phase")])
def test_sharded_sweep_world2_gloo(tmp_path, n_chn, n_frames, mode):
    """two_phase: sweep_dist.sweep_two_phase (seed -> MAX all-reduce -> main pass -> MAX all-reduce, round 5) over gloo with the
    oracle as each rank's engine: the merged table is the single pass's, bit for bit."""
    blksiz, world = 2048, 2
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world),
                               str(port), str(blksiz), str(n_chn), str(n_frames), str(tmp_path), mode])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    data = np.stack([0.8 * harmonic_stack(n_frames, f0=110.0 * (1 + 0.37 * c)) +
                     0.1 * xorshift_uniform(n_frames, seed=40 + c) for c in range(n_chn)], axis=1)
    ref = orc.CliEngine(blksiz, n_chn)
    ref.analyze_file(data.astype(np.float32), 0, 360, 1)
    want = ref.peaks()
    for r in range(world):
        got = np.load(os.path.join(tmp_path, "peaks_%d.npy" % r))
        assert np.array_equal(got, want)  # bit-exact: sharding cannot change a max


def test_selection_logic_matches_oracle_main(pkg):
    from phaserotate_lv2_amd import sweep_dist as sd
    P, n = 2048, 40000
    chans = [0.8 * harmonic_stack(n, f0=97.0 * (1 + 0.41 * c)) + 0.02 * xorshift_uniform(n, seed=c)
             for c in range(2)]
    data = np.stack(chans, axis=1).astype(np.float32)
    for link in (False, True):
        ma, fo, rz, rm, tbl = orc.cli_find_min(P, 2, data, stride=1, link=link)
        mins, p_min, r_zro = sd.coarse_candidates(tbl, 1, link)
        min_angle = [0, 0]
        for a, chs in mins.items():  # stride 1: last candidate wins (cli/phase-rotate.cc:859-865)
            for c in chs:
                min_angle[c] = a
        out, found = sd.finish_selection(min_angle, p_min)
        assert out == ma.tolist() and [int(f) for f in found] == fo.tolist()
        assert np.array_equal(r_zro, rz)
    # coarse candidates of the default stride are the ones the oracle refines
    ma24, fo24, _, _, tbl24 = orc.cli_find_min(P, 2, data, stride=24)
    mins, _, _ = sd.coarse_candidates(tbl24, 24, False)
    assert all(a % 24 == 0 for a in mins) and set(c for v in mins.values() for c in v) == {0, 1}
