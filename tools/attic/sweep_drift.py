"""cfg4 pass time (harmonic, blksiz 2048) against what the device did before: groups of ten passes back to back from a fresh
process, then groups of three after two seconds of idling.  python3 tools/sweep_drift.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda", 0)
P, n_chn = 2048, 2
total = 3600 * 96000 // P * P
x = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, "harmonic")
eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
peaks = torch.zeros((n_chn, 360), device=dev)
ts = torch.cuda.Stream(device=dev)   # (a stream of the caller's: stream 0 would mean the engine's own, which torch events do not see)
torch.cuda.set_stream(ts)
st = ts.cuda_stream
def one():
    eng.reset(); peaks.zero_()
    eng.sweep_dev(x.data_ptr(), total, total, 0, 360, 1, -1, peaks.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
out = []
for grp in range(60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): one()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 10)
print("groups of 10 passes, ms:", " ".join("%.2f" % v for v in out))
time.sleep(2.0)
out = []
for grp in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): one()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 3)
print("after 2 s idle, groups of 3:", " ".join("%.2f" % v for v in out))
