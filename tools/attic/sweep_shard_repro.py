import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
from phaserotate_lv2_amd import sweep_dist
dev = torch.device("cuda:0")
# poison the memory the library's allocations will reuse
for val in (1.0e30, float("nan")):
    big = torch.full((int(60e9) // 4,), val, device=dev); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
P, n_chn = 2048, 2
nb = 3001
n = nb * P - 777
g = torch.Generator(device=dev).manual_seed(2)
x = (torch.rand((n_chn, n), generator=g, device=dev) - 0.5) * 0.9
def single(unfused):
    pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, unfused)
    one = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk1 = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    one.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk1.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    one.sync()
    A = one.shard_align_blocks()
    one.close()
    return pk1, A
def shards(unfused, A):
    pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, unfused)
    plan = sweep_dist.plan_shards(n, P, 2, align=A)
    pk2 = torch.zeros((n_chn, 360), device=dev)
    parts = []
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
        parts.append(part.clone())
        pk2 = torch.maximum(pk2, part)
        eng.close()
    return pk2, parts
ref1, A = single(1)
ref2, rparts = shards(1, A)
print("unfused: single == shards:", bool(torch.equal(ref1, ref2)))
for rep in range(12):
    f1, _ = single(0)
    f2, parts = shards(0, A)
    print("rep", rep, "fused single == unfused single:", bool(torch.equal(f1, ref1)), " fused shards == unfused shards:", bool(torch.equal(f2, ref2)),
          " per shard:", [bool(torch.equal(a, b)) for a, b in zip(parts, rparts)])
    for name, a, b in (("single", f1, ref1), ("shard0", parts[0], rparts[0]), ("shard1", parts[1], rparts[1])):
        d = torch.nonzero(a != b)
        if d.numel():
            i = d[0].tolist()
            print("   ", name, "differs in", d.shape[0], "entries, first", i, "got", float(a[i[0], i[1]]), "want", float(b[i[0], i[1]]), "angles", sorted(set(d[:, 1].tolist()))[:12])
pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, 0)
