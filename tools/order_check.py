# Discriminate "slow first measurement" from "poly genuinely slower": interleave, many repetitions, no discarding.
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
engs = {"poly": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0), "direct": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=pkg.CFG_DIRECT_FORM)}
def run(e, k):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(k):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / k
# cold first: exactly what bench.py does for the headline (3 warmup + 20 timed), on a fresh process
for name in ("poly", "direct", "poly", "direct", "direct", "poly"):
    w = run(engs[name], 3); t = run(engs[name], 20)
    print("%-7s warmup-avg %.4f ms  timed-avg %.4f ms" % (name, w, t), flush=True)
