"""How the headline launch's duration moves with what the device did before it: groups of launches back to back from a fresh
process, then again after two seconds of idling (HIP events on the launch stream).  python3 tools/headline_drift.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda", 0)
n = 1 << 27
x = torch.rand(n, device=dev) - 0.5
y = torch.empty_like(x)
th = -2.0 * np.pi * 90.0 / 360.0
rot = np.array([np.cos(th), np.sin(th)], np.float32)
eng = pkg.Engine(pkg.KIND_LV2, 48000, 1)
ts = torch.cuda.Stream(device=dev)   # (a stream of the caller's: stream 0 would mean the engine's own, which torch events do not see)
torch.cuda.set_stream(ts)
st = ts.cuda_stream

def group(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

print("groups of 100 launches, us:", " ".join("%.0f" % (1e3 * group(100)) for _ in range(40)))
for idle in (0.5, 2.0, 5.0):
    time.sleep(idle)
    print("after %.1f s idle, groups of 20:" % idle, " ".join("%.0f" % (1e3 * group(20)) for _ in range(12)))
