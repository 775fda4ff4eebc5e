"""Multi-GPU readiness that can be shown on a one-GPU box:
 * the cfg4 workload at FULL size (60 min stereo 96 kHz = 345.6 M frames x 2 channels, generated on the device), swept
   as 8 shards on one device and MAX-merged, equals the single pass bit for bit;
 * prot_allreduce_peaks keeps ONE set of communicators for many all-reduces and destroys it with the last engine
   (a counting stand-in for librccl, loaded through the test hook prot_internal_set_rccl_library; the real library wants one rank per GPU).
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg4_full_size_sharded_sweep_equals_single_pass(pkg):
    import torch
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    P, n_chn, world = 2048, 2, 8
    n = 3600 * 96000  # BASELINE configs[3]
    x = torch.empty((n_chn, n), device=dev)
    g = torch.Generator(device=dev).manual_seed(44)
    L = 4 * 96000  # four seconds: a whole number of cycles of every harmonic below
    tt = np.arange(L, dtype=np.float64) / 96000.0
    u = torch.linspace(0.0, 1.0, n, dtype=torch.float64, device=dev)
    for c in range(n_chn):  # an asymmetric harmonic stack (a real minimum) with a slow swell (maxima in several shards) + noise
        f0 = 110.0 + 27.5 * c
        pat = sum((0.5 / k) * np.cos(2 * np.pi * k * f0 * tt + 0.3 * k) for k in range(1, 9))
        pat_d = torch.from_numpy(pat.astype(np.float32)).to(dev)
        x[c] = pat_d.repeat(n // L + 1)[:n]
        x[c] *= (0.6 + 0.3 * torch.sin(2 * np.pi * (3 + c) * u)).float()
        x[c] += 0.02 * (torch.rand(n, generator=g, device=dev) - 0.5)
    del u
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    one.sync()
    A = one.shard_align_blocks()
    plan = sweep_dist.plan_shards(n, P, world, align=A)
    assert len(plan) == world and all(b0 % A == 0 for b0, _ in plan)
    one.close()
    merged = torch.zeros((n_chn, 360), device=dev)
    best_shard = np.zeros((n_chn, 360), np.int64)
    for r in range(world):
        f0, nf = sweep_dist.shard_frames(n, P, plan, r)
        eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
        H = int(eng.info.history)
        if r > 0:
            halo = x[:, f0 - H:f0].contiguous()
            torch.cuda.synchronize()
            keep = sweep_dist.prime_history(eng, halo)  # noqa: F841 (kept alive until the sync below)
        xs = x[:, f0:f0 + nf]  # a view: rows n apart
        part = torch.zeros((n_chn, 360), device=dev)
        flags = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
        torch.cuda.synchronize()
        eng.sweep_dev(xs.data_ptr(), n, nf, 0, 360, 1, -1, part.data_ptr(), flags)
        eng.sync()
        best_shard[(part > merged).cpu().numpy()] = r
        merged = torch.maximum(merged, part)
        eng.close()
    assert torch.equal(pk1, merged)
    assert len(set(best_shard.reshape(-1).tolist())) > 1  # the table really is a merge of several shards
    pk = pk1.cpu().numpy()
    assert all(0 < int(pk[c].argmin()) < 360 for c in range(n_chn))  # a real minimum, not the raw-peak entry


@pytest.mark.parametrize("P,seconds", [(2048, 600), (16384, 600), (8192, 300), (1024, 300)])
def test_two_phase_sharded_sweep_equals_single_pass_and_lists_less(pkg, P, seconds):
    """Round 5: a shard that seeds its pruning bound from its own frames cannot know that the loud passage is
    somewhere else.  Two phases -- every shard's seeding pass alone (prot_sweep_seed_dev), the sampled tables MAX-merged, the main
    passes against the merged bound -- on eight shards of material whose level swells by 20 dB towards one end: the merged final
    table equals the single pass bit for bit (every fused route: blksiz 2048, 16384, 8192, 1024), the quiet shards list far fewer
    frames than they do alone, and through prot_cli_analyze_file_multi (eight engines on one device, host-merged tables)."""
    import torch
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    n_chn, world = 2, 8
    n = seconds * 96000 // P * P
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.empty((n_chn, n), device=dev)
    u = torch.linspace(0.0, 1.0, n, dtype=torch.float64, device=dev)
    L = 96000
    tt = np.arange(L, dtype=np.float64) / 96000.0
    for c in range(n_chn):
        pat = sum((0.5 / k) * np.cos(2 * np.pi * k * (110.0 + 11 * c) * tt + 0.3 * k) for k in range(1, 9))
        x[c] = torch.from_numpy(pat.astype(np.float32)).to(dev).repeat(n // L + 1)[:n]
        x[c] *= (0.08 + 0.8 * u ** 4).float()                      # quiet for most of the file, loud at the end
        x[c] += 0.01 * (torch.rand(n, generator=g, device=dev) - 0.5)
    del u
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    one.sync()
    plan = sweep_dist.plan_shards(n, P, world, align=one.shard_align_blocks())
    H = int(one.info.history)
    one.close()
    engines = [pkg.Engine(pkg.KIND_CLI, P, n_chn) for _ in range(world)]
    tabs = [torch.zeros((n_chn, 360), device=dev) for _ in range(world)]
    listed = {}
    for two_phase in (False, True, "one call"):
        shards, halos = [], []
        for r, e in enumerate(engines):
            f0, nf = sweep_dist.shard_frames(n, P, plan, r)
            fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
            shards.append((x[:, f0:f0 + nf], nf, fl))
            halos.append(x[:, f0 - H:f0].contiguous() if r > 0 else None)
        if two_phase != "one call":
            for r, e in enumerate(engines):
                e.reset()
                tabs[r].zero_()
            torch.cuda.synchronize()
            for r, e in enumerate(engines):
                if r > 0:
                    e.prime_dev(halos[r].data_ptr(), H, H)
        else:
            for t in tabs:
                t.fill_(7.0)                     # (prot_sweep_shard_seed_dev zeroes the table itself)
            torch.cuda.synchronize()
        if two_phase:
            for r, e in enumerate(engines):
                xs, nf, fl = shards[r]
                if two_phase == "one call":      # reset + table zeroed + history primed + seeding pass: prot_sweep_shard_seed_dev
                    e.sweep_shard_seed_dev(halos[r].data_ptr() if r > 0 else 0, H, H if r > 0 else 0, xs.data_ptr(), n, nf, 0, 360, 1,
                                           tabs[r].data_ptr(), fl, 8)
                else:
                    e.sweep_seed_dev(xs.data_ptr(), n, nf, 0, 360, 1, -1, tabs[r].data_ptr(), fl, 8)
                e.sync()
            seed = torch.stack(tabs).amax(0)     # (the first all-reduce)
            assert float(seed.max()) > 0
            for t in tabs:
                t.copy_(seed)
            torch.cuda.synchronize()
        for r, e in enumerate(engines):
            xs, nf, fl = shards[r]
            e.sweep_dev(xs.data_ptr(), n, nf, 0, 360, 1, -1, tabs[r].data_ptr(), fl)
            e.sync()
        merged = torch.stack(tabs).amax(0)       # (the second all-reduce)
        assert torch.equal(merged, pk1), two_phase
        listed[two_phase] = [e.sweep_stats_fused()["listed"] for e in engines]
    # the quiet shards (the first half of the file) list next to nothing against the global bound
    assert sum(listed[True][:4]) * 4 < sum(listed[False][:4]) + 4, listed
    assert listed["one call"] == listed[True], listed
    for e in engines:
        e.close()
    # ... and the host-file entry point: eight engines on one device against one engine
    nh = min(n, (1 << 22) * 2 + 4096 * 5)
    xh = np.ascontiguousarray(x[:, :nh].t().cpu().numpy())
    a = pkg.PhaseRotate(P, n_chn)
    a.analyze_file(xh, 0, 360, 1)
    want = a.peaks()
    eng8 = [pkg.PhaseRotate(P, n_chn) for _ in range(world)]
    pkg.analyze_file_multi(eng8, xh, 0, 360, 1)
    for e in eng8:
        assert np.array_equal(e.peaks(), want)
    assert int(want[0].argmin()) > 0


def test_sweep_calls_beyond_the_lists_32_bit_offsets_are_split(pkg):
    """the fused route's step descriptors address the survivor lists with 32-bit entry offsets; a resident call of more
    than 2^31 channel-frames is taken in two halves on a chunk boundary.  With the limit lowered by the test hook a modest call
    splits (recursively): the same table as the unsplit call, bit for bit, for two block sizes and with a ragged end."""
    import torch
    dev = torch.device("cuda:0")
    for P, n in ((2048, (1 << 23) + 12345), (16384, (1 << 23) + 999)):
        g = torch.Generator(device=dev).manual_seed(P)
        x = (torch.rand((2, n), generator=g, device=dev) - 0.5) * 0.9
        x[0] += 0.4 * torch.sin(torch.arange(n, device=dev) * 0.013)
        tabs = []
        for lg in (0, 22):
            pkg.set_test_option(pkg.TEST_FUSED_SPLIT_LOG2, lg)
            try:
                e = pkg.Engine(pkg.KIND_CLI, P, 2)
                pk = torch.zeros((2, 360), device=dev)
                torch.cuda.synchronize()
                e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
                e.sync()
                tabs.append(pk.clone())
                e.close()
            finally:
                pkg.set_test_option(pkg.TEST_FUSED_SPLIT_LOG2, 0)
        assert torch.equal(tabs[0], tabs[1]), P
        assert float(tabs[0].min()) > 0


STUB = r"""
#include <stddef.h>
static int n_init, n_destroy, n_start, n_end, n_allreduce, open_groups;
int ncclCommInitAll (void** comms, int n, const int* devs) { for (int i = 0; i < n; ++i) comms[i] = (void*)(size_t)(0x1000 + 16 * n_init + i); ++n_init; return 0; }
int ncclCommDestroy (void* c) { (void)c; ++n_destroy; return 0; }
int ncclGroupStart (void) { ++n_start; ++open_groups; return 0; }
int ncclGroupEnd (void) { ++n_end; --open_groups; return 0; }
int ncclAllReduce (const void* s, void* r, size_t count, int dt, int op, void* comm, void* stream) { (void)s; (void)r; (void)comm; (void)stream; ++n_allreduce; return (count == 720 && dt == 7 && op == 2 && open_groups == 1) ? 0 : 1; }
void stub_counts (int* o) { o[0] = n_init; o[1] = n_destroy; o[2] = n_start; o[3] = n_end; o[4] = n_allreduce; o[5] = open_groups; }
"""

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
stub = ctypes.CDLL(os.environ["TEST_RCCL_STUB"])
def counts():
    o = (ctypes.c_int * 6)(); stub.stub_counts(o); return list(o)
lib = pkg.load_library()
lib.prot_internal_set_rccl_library.argtypes = [ctypes.c_char_p]
assert lib.prot_internal_set_rccl_library(os.environ["TEST_RCCL_STUB"].encode()) == 0
pkg.set_test_option(pkg.TEST_ALLREDUCE, 2)  # the collective although both engines sit on one device
a, b = pkg.Engine(pkg.KIND_CLI, 2048, 2), pkg.Engine(pkg.KIND_CLI, 2048, 2)
arr = (ctypes.c_void_p * 2)(a._h, b._h)
for _ in range(25):
    assert lib.prot_allreduce_peaks(arr, 2) == 0
c1 = counts()
c = pkg.Engine(pkg.KIND_CLI, 2048, 2)  # a third engine, another device list (a, c)
arr2 = (ctypes.c_void_p * 2)(a._h, c._h)
for _ in range(3):
    assert lib.prot_allreduce_peaks(arr2, 2) == 0
c2 = counts()
a.close(); b.close()
c3 = counts()   # one engine still alive: communicators kept
c.close()
c4 = counts()   # last engine gone: communicators destroyed
print("COUNTS", [c1, c2, c3, c4])
"""


def test_allreduce_keeps_its_communicators(tmp_path):
    src = tmp_path / "rccl_stub.c"
    src.write_text(STUB)
    so = str(tmp_path / "librccl_stub.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, str(src)], check=True)
    env = dict(os.environ, TEST_RCCL_STUB=so)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("COUNTS")]
    assert r.returncode == 0 and line, r.stdout[-3000:]
    c1, c2, c3, c4 = eval(line[0][7:])
    # [init, destroy, group start, group end, all-reduce, open groups]
    assert c1 == [1, 0, 25, 25, 50, 0]   # ONE ncclCommInitAll for 25 all-reduces of two engines, groups balanced
    assert c2 == [1, 0, 28, 28, 56, 0]   # the same device list (0, 0) again: still the same communicators
    assert c3 == [1, 0, 28, 28, 56, 0]   # kept while an engine lives
    assert c4 == [1, 2, 28, 28, 56, 0]   # both destroyed with the last engine


CHILD_REAL = r"""
import ctypes, glob, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
# the RCCL that belongs to the HIP runtime of this process (torch's bundled pair; a C host gets /opt/rocm's pair by itself)
cand = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
assert cand, "no librccl beside torch"
lib.prot_internal_set_rccl_library.argtypes = [ctypes.c_char_p]
assert lib.prot_internal_set_rccl_library(cand[0].encode()) == 0
e = pkg.Engine(pkg.KIND_CLI, 2048, 2)
n = 2048 * 40
x = torch.rand((2, n), device="cuda:0") - 0.5
e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, 0, pkg.F_STREAM_START | pkg.F_STREAM_END, 0)
before = e.peaks().copy()
assert before.min() > 0
pkg.set_test_option(pkg.TEST_ALLREDUCE, 3)   # the collective for ONE engine: ncclCommInitAll (1 device), ncclAllReduce (MAX, in place)
arr = (ctypes.c_void_p * 1)(e._h)
for _ in range(3):
    rc = lib.prot_allreduce_peaks(arr, 1)
    assert rc == 0, (rc, lib.prot_last_error())
after = e.peaks().copy()
e.close()
print("REAL_RCCL", cand[0], bool((before == after).all()))
"""


def test_allreduce_through_the_real_rccl_with_one_rank(tmp_path):
    """VERDICT r4, missing 2: prot_allreduce_peaks' RCCL branch had only ever met a counting stand-in.  A one-GPU box cannot run a
    real multi-rank collective, but it can run the REAL library with one rank: dlopen, the five symbols, ncclCommInitAll on one
    device, the group calls, ncclAllReduce (float32, MAX, in place, on the engine's stream) with the enum values the library
    expects, and the communicator's destruction with the last engine -- the table comes back unchanged (PROT_TEST_ALLREDUCE = 3)."""
    r = subprocess.run([sys.executable, "-c", CHILD_REAL % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("REAL_RCCL")]
    assert r.returncode == 0 and line, r.stdout[-3000:]
    assert line[0].endswith("True"), line[0]


def test_file_analysis_double_buffered_equals_resident_pass(pkg):
    """prot_cli_analyze_file on a host buffer of several device chunks (upload and de-interleave of chunk k+1 under the
    sweep of chunk k: two staging pairs, a copy stream) against ONE resident prot_sweep_dev pass: the same table, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    P, n_chn = 2048, 2
    n = (1 << 22) * 3 + 54321
    rng = np.random.default_rng(7)
    x = (rng.random((n, n_chn), dtype=np.float32) - 0.5) * 0.8
    x[:, 0] += 0.3 * np.sin(np.arange(n) * 0.01).astype(np.float32)
    g = pkg.PhaseRotate(P, n_chn)
    g.analyze_file(x.reshape(-1), 0, 360, 8)
    tab = g.peaks()
    g.close()
    e = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
    pk = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    e.sweep_dev(xd.data_ptr(), n, n, 0, 360, 8, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    e.sync()
    assert np.array_equal(tab, pk.cpu().numpy())
    e.close()


@pytest.mark.parametrize("P,ang,world", [(2048, (0, 360, 1), 5), (16384, (0, 360, 24), 4), (2048, (-12, 13, 1), 8), (1024, (0, 360, 1), 3)])
def test_angle_sharded_sweep_equals_single_pass(pkg, P, ang, world):
    """BASELINE configs[3] calls the multi-GPU sweep "angle-sharded" (SURVEY 8e's first option): every rank sweeps the whole file
    for every world-th angle of the list, one MAX all-reduce merges the tables (sweep_dist.sweep_angle_sharded).  `world` engines
    on one device, the all-reduce replaced by a maximum over their tables: equal to the single pass bit for bit -- the fine
    pass, the coarse pass (every 24th angle), the search's +-12 degree pass around zero (negative angles, the raw peak at angle
    0 with one rank only), more ranks than some lists have angles."""
    import torch
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    n_chn = 2
    n = (1 << 21) // P * P + 3 * P
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.empty((n_chn, n), device=dev)
    tt = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
    for c in range(n_chn):
        sig = sum((0.5 / k) * torch.cos(2 * np.pi * k * (110.0 + 27.5 * c) * tt + 0.3 * k) for k in range(1, 17))
        x[c] = sig.float() * 0.6 + 0.02 * (torch.rand(n, generator=g, device=dev) - 0.5)
    fl = pkg.F_STREAM_START | pkg.F_STREAM_END
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    one.sweep_dev(x.data_ptr(), n, n, ang[0], ang[1], ang[2], -1, pk1.data_ptr(), fl)
    one.sync()
    one.close()
    visited = sweep_dist.angle_list(*ang)
    shares = [sweep_dist.angle_shard(ang[0], ang[1], ang[2], world, r) for r in range(world)]
    got = sorted(a for sh in shares if sh is not None for a in sweep_dist.angle_list(*sh))
    assert got == sorted(visited)                    # every angle of the call exactly once
    tabs = []
    for r in range(world):
        e = pkg.Engine(pkg.KIND_CLI, P, n_chn)
        t = torch.zeros((n_chn, 360), device=dev)
        sweep_dist.sweep_angle_sharded(e, x, t, fl, world, r, 0, ang, allreduce=lambda t_: t_)
        e.sync()
        tabs.append(t)
        e.close()
    merged = torch.stack(tabs).amax(0)
    assert torch.equal(merged, pk1)
    assert sum(int((t > 0).any()) for t in tabs) == sum(sh is not None for sh in shares)


@pytest.mark.parametrize("side_stream", [False, True])
def test_sweep_two_phase_with_default_arguments_on_a_real_engine(pkg, side_stream):
    """sweep_dist.sweep_two_phase's default stream was the engine's private non-blocking stream, while peaks.zero_() and
    the default all-reduce run on torch's current stream: nothing ordered them.  Default arguments now mean torch's CURRENT stream
    (on torch's default stream -- handle 0, which the C ABI reads as "the engine's own" -- the host orders the two).  Two shards of
    one file, each through sweep_two_phase(fresh=True) with NO stream argument, the all-reduce a torch operation on the current
    stream; repeated: the merged table equals the single pass bit for bit every time."""
    import torch
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    P, n_chn, world = 2048, 2, 2
    n = (1 << 23) // P * P
    g = torch.Generator(device=dev).manual_seed(21)
    tt = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
    x = torch.empty((n_chn, n), device=dev)
    for c in range(n_chn):
        sig = sum((0.5 / k) * torch.cos(2 * np.pi * k * (110.0 + 27.5 * c) * tt + 0.3 * k) for k in range(1, 17))
        x[c] = sig.float() * (0.2 + 0.5 * (tt / tt[-1]).float()) + 0.02 * (torch.rand(n, generator=g, device=dev) - 0.5)
    del tt
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    one.sync()
    plan = sweep_dist.plan_shards(n, P, world, align=one.shard_align_blocks())
    H = int(one.info.history)
    one.close()
    engines = [pkg.Engine(pkg.KIND_CLI, P, n_chn) for _ in range(world)]
    ts = torch.cuda.Stream(device=dev) if side_stream else torch.cuda.current_stream(dev)
    with torch.cuda.stream(ts):
        for rep in range(6):
            tabs = [torch.full((n_chn, 360), 3.0, device=dev) for _ in range(world)]  # (fresh=True zeroes them)
            other = {}

            def merge_with(r):
                # stands where dist.all_reduce would: a torch operation on the CURRENT stream that reads and writes the table
                def f(t):
                    if r in other:
                        t.copy_(torch.maximum(t, other[r]))
                    return t
                return f
            # phase one of shard 1 first, to have its seed table for shard 0's "all-reduce" (and vice versa): seeds only
            seeds = []
            for r, e in enumerate(engines):
                f0, nf = sweep_dist.shard_frames(n, P, plan, r)
                halo = x[:, f0 - H:f0].contiguous() if r > 0 else None
                fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
                t = torch.zeros((n_chn, 360), device=dev)
                e.sweep_shard_seed_dev(halo.data_ptr() if r > 0 else 0, H, H if r > 0 else 0, x[:, f0:].data_ptr(), n, nf, 0, 360, 1,
                                       t.data_ptr(), fl, 32, ts.cuda_stream)
                e.sync()
                seeds.append(t)
            for r, e in enumerate(engines):
                other[r] = seeds[1 - r]
            for r, e in enumerate(engines):
                f0, nf = sweep_dist.shard_frames(n, P, plan, r)
                halo = x[:, f0 - H:f0].contiguous() if r > 0 else None
                fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
                sweep_dist.sweep_two_phase(e, x[:, f0:f0 + nf], tabs[r], fl, world, allreduce=merge_with(r), halo=halo, fresh=True)
            merged = torch.maximum(tabs[0], tabs[1])   # on the current stream, like a caller's next torch operation
            for e in engines:
                e.sync()
            ts.synchronize()
            assert torch.equal(merged, pk1), rep
    for e in engines:
        e.close()


def test_call_on_a_callers_stream_waits_for_the_engines_own_stream(pkg):
    """include/prot_hip.h promises that calls on one engine are ordered on the device whatever streams they name;
    a call on the engine's OWN stream (stream = NULL) followed by a call on a caller's stream S was not (only caller -> other was).
    prime_dev(NULL) then sweep_dev(S), many times on fresh engines, against the same two calls on one stream: same tables -- and
    apply_dev(NULL) then apply_dev(S) on a stateful engine against the one-stream sequence, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    P, n_chn = 2048, 2
    n = (1 << 22)
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand((n_chn, n), generator=g, device=dev) - 0.5)
    x[:, :P * 64] *= 3.0    # what the history holds decides the first blocks' peaks
    s = torch.cuda.Stream(device=dev)
    th = -2.0 * np.pi * 33.0 / 360.0
    rot = np.array([np.cos(th), np.sin(th)] * n_chn, np.float32)
    e0 = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    H = int(e0.info.history)
    e0.close()
    ref_pk = ref_y = None
    for rep in range(8):
        for mixed in (False, True):
            e = pkg.Engine(pkg.KIND_CLI, P, n_chn)
            pk = torch.zeros((n_chn, 360), device=dev)
            y1 = torch.empty((n_chn, n // 2), device=dev)
            y2 = torch.empty((n_chn, n // 2), device=dev)
            torch.cuda.synchronize()
            first = 0 if mixed else s.cuda_stream
            # a long stateful call, then a call that continues from its history
            e.apply_dev(x.data_ptr(), n, n // 2, y1.data_ptr(), n // 2, n // 2, rot, 0, first)
            e.apply_dev(x[:, n // 2:].data_ptr(), n, n // 2, y2.data_ptr(), n // 2, n // 2, rot, 0, s.cuda_stream)
            e.sync()
            s.synchronize()
            e.reset()
            e.prime_dev(x[:, P * 64 - H:].data_ptr(), n, H, first)
            e.sweep_dev(x[:, P * 64:].data_ptr(), n, n // 2, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_END, s.cuda_stream)
            e.sync()
            s.synchronize()
            if ref_pk is None:
                ref_pk, ref_y = pk.clone(), y2.clone()
            assert torch.equal(pk, ref_pk), (rep, mixed)
            assert torch.equal(y2, ref_y), (rep, mixed)
            e.close()


@pytest.mark.parametrize("P", [16384, 2048])
def test_cfg4_full_size_at_the_references_block_size(pkg, P):
    """BASELINE configs[3] at FULL size (60 min stereo 96 kHz, the bench's harmonic material) at the block size the reference
    itself picks for a 96 kHz file, 16384 (cli/phase-rotate.cc:749-755), and at -f 2048: ONE resident pass ==
    eight two-phase frame-block shards (seed, merged seeds, main pass against the global bound) == eight angle shards, bit for
    bit; the pruned pass evaluates < 10 % of its tiles and equals the unpruned table; and the first 10 s of the same material
    against the CPU oracle: table within 1e-3 dB, minimum-peak angle index exact."""
    import torch
    import bench
    import orc
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda:0")
    n_chn, world = 2, 8
    n = 3600 * 96000
    x = bench._sweep_signal(torch, np, dev, n_chn, 0, n, 1234, "harmonic")
    fl_all = pkg.F_STREAM_START | pkg.F_STREAM_END
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), fl_all)
    one.sync()
    ev, seen = one.sweep_stats()
    assert seen > 0 and ev / seen < 0.1, (ev, seen)
    A = one.shard_align_blocks()
    H = int(one.info.history)
    plan = sweep_dist.plan_shards(n, P, world, align=A)
    # unpruned: the same table
    pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1)
    try:
        one.reset()
        pku = torch.zeros((n_chn, 360), device=dev)
        torch.cuda.synchronize()
        one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pku.data_ptr(), fl_all)
        one.sync()
    finally:
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
    assert torch.equal(pku, pk1)
    one.close()
    pk = pk1.cpu().numpy()
    assert all(0 < int(pk[c].argmin()) < 360 for c in range(n_chn))
    # eight frame-block shards in two phases, one engine per shard
    engines = [pkg.Engine(pkg.KIND_CLI, P, n_chn) for _ in range(world)]
    tabs = [torch.full((n_chn, 360), 9.0, device=dev) for _ in range(world)]
    shards = []
    torch.cuda.synchronize()
    for r, e in enumerate(engines):
        f0, nf = sweep_dist.shard_frames(n, P, plan, r)
        fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
        halo = x[:, f0 - H:f0].contiguous() if r > 0 else None
        shards.append((f0, nf, fl, halo))
        e.sweep_shard_seed_dev(halo.data_ptr() if r > 0 else 0, H, H if r > 0 else 0, x[:, f0:].data_ptr(), n, nf, 0, 360, 1,
                               tabs[r].data_ptr(), fl, 8)
        e.sync()
    seed = torch.stack(tabs).amax(0)             # (the first all-reduce)
    assert float(seed.max()) > 0 and bool((seed <= pk1).all())
    for t in tabs:
        t.copy_(seed)
    torch.cuda.synchronize()
    for r, e in enumerate(engines):
        f0, nf, fl, _ = shards[r]
        e.sweep_dev(x[:, f0:].data_ptr(), n, nf, 0, 360, 1, -1, tabs[r].data_ptr(), fl)
        e.sync()
    merged = torch.stack(tabs).amax(0)           # (the second all-reduce)
    assert torch.equal(merged, pk1)
    listed = sum(e.sweep_stats_fused()["listed"] for e in engines)
    tested = sum(e.sweep_stats_fused()["tested"] for e in engines)
    assert tested >= n * n_chn and listed < 0.1 * tested, (listed, tested)
    for e in engines:
        e.close()
    # eight angle shards (configs[3]'s wording): every rank the whole file, every eighth angle
    e = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    tabs = []
    for r in range(world):
        e.reset()
        t = torch.zeros((n_chn, 360), device=dev)
        torch.cuda.synchronize()
        sweep_dist.sweep_angle_sharded(e, x, t, fl_all, world, r, 0, (0, 360, 1), allreduce=lambda t_: t_)
        e.sync()
        tabs.append(t)
    assert torch.equal(torch.stack(tabs).amax(0), pk1)
    # the first 10 s against the oracle
    ns = 10 * 96000 // P * P
    e.reset()
    pk2 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    e.sweep_dev(x.data_ptr(), n, ns, 0, 360, 1, -1, pk2.data_ptr(), fl_all)
    e.sync()
    e.close()
    o = orc.CliEngine(P, n_chn)
    o.analyze_file(np.ascontiguousarray(x[:, :ns].cpu().numpy().T).reshape(-1), 0, 360, 1)
    pg, po = pk2.cpu().numpy(), o.peaks()
    db = np.abs(20 * np.log10(np.maximum(pg, 1e-30)) - 20 * np.log10(np.maximum(po, 1e-30))).max()
    assert db < 1e-3, db
    assert [int(pg[c].argmin()) for c in range(n_chn)] == [int(po[c].argmin()) for c in range(n_chn)]
