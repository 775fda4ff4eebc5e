"""Interleaved A/B of the polyphase and direct forms of the 48 kHz and 96 kHz LV2 filters in ONE process:
R rounds, each round times every variant (20 launches after 3 untimed), order rotated per round.
Prints median / min / max per variant -- the spread is the run-to-run noise any comparison has to beat."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10
PF, DF = pkg.CFG_POLYPHASE, pkg.CFG_DIRECT_FORM
engs = {"poly48": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=PF), "direct48": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=DF),
        "poly96": pkg.Engine(pkg.KIND_LV2, 96000, 1, flags=PF), "direct96": pkg.Engine(pkg.KIND_LV2, 96000, 1, flags=DF),
        "poly_cli2048": pkg.Engine(pkg.KIND_CLI, 2048, 1, flags=PF), "direct_cli2048": pkg.Engine(pkg.KIND_CLI, 2048, 1, flags=DF),
        "poly_cli1024": pkg.Engine(pkg.KIND_CLI, 1024, 1, flags=PF), "direct_cli1024": pkg.Engine(pkg.KIND_CLI, 1024, 1, flags=DF)}
for k, e in engs.items():
    print("%-15s partitions %d hop %d" % (k, e.info.partitions, e.info.hop))
def run(e, k):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(k):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / k
for e in engs.values():
    run(e, 30)  # clock ramp, discarded for every variant alike
names = list(engs)
res = {k: [] for k in names}
for r in range(R):
    for name in names[r % len(names):] + names[:r % len(names)]:
        run(engs[name], 3)
        res[name].append(run(engs[name], 20))
for name in names:
    v = np.array(res[name])
    print("%-15s median %.4f  min %.4f  max %.4f  (n=%d)" % (name, np.median(v), v.min(), v.max(), len(v)))
