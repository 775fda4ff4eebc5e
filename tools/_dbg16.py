import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from __graft_entry__ import load_package
pkg = load_package()
from phaserotate_lv2_amd import sweep_dist
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
n_chn, total = 2, secs * 96000 // P * P
x = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, "harmonic")
eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
H = int(eng.info.history)
pk = torch.zeros((n_chn, 360), device=dev)
torch.cuda.synchronize()
eng.sweep_dev(x.data_ptr(), total, total, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
eng.sync()
single = pk.clone()
pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1)
eng.reset(); pk.zero_(); torch.cuda.synchronize()
eng.sweep_dev(x.data_ptr(), total, total, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
eng.sync()
pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
print("single pruned == unpruned:", bool(torch.equal(single, pk)))
plan = sweep_dist.plan_shards(total, P, 8, align=eng.shard_align_blocks())
for two_phase in (False, True):
    tabs = []
    seeds = []
    if two_phase:
        for r, (c0, c1) in enumerate(plan):
            xs = x[:, c0 * P:c1 * P]
            fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == 7 else 0)
            eng.reset(); pk.zero_(); torch.cuda.synchronize()
            if r > 0:
                hh = x[:, c0 * P - H:c0 * P].contiguous()
                eng.prime_dev(hh.data_ptr(), H, H)
            eng.sweep_seed_dev(xs.data_ptr(), total, xs.shape[1], 0, 360, 1, -1, pk.data_ptr(), fl, 8)
            eng.sync(); seeds.append(pk.clone())
        gseed = torch.stack(seeds).amax(0)
        print("seed <= single everywhere:", bool((gseed <= single).all()), "max seed/single", float((gseed / single).max()))
    for r, (c0, c1) in enumerate(plan):
        xs = x[:, c0 * P:c1 * P]
        fl = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == 7 else 0)
        eng.reset(); pk.zero_(); torch.cuda.synchronize()
        if r > 0:
            halo = x[:, c0 * P - H:c0 * P].contiguous()
            eng.prime_dev(halo.data_ptr(), H, H)
        if two_phase:
            pk.copy_(gseed); torch.cuda.synchronize()
        eng.sweep_dev(xs.data_ptr(), total, xs.shape[1], 0, 360, 1, -1, pk.data_ptr(), fl)
        eng.sync()
        tabs.append(pk.clone())
    merged = torch.stack(tabs).amax(0)
    d = torch.nonzero(merged != single)
    print("two_phase", two_phase, "merged == single:", bool(torch.equal(merged, single)), "differing entries", d.shape[0])
    for i in d[:6].tolist():
        print("   entry", i, "merged %.9g single %.9g" % (float(merged[i[0], i[1]]), float(single[i[0], i[1]])), "per shard", ["%.9g" % float(t[i[0], i[1]]) for t in tabs])
