"""BASELINE config 3 in its own shape (LV2 stereo, 96 kHz filter, 256 x 1024-frame buffers per stateful launch): launch period by
events; run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration.  python3 tools/cfg3_probe.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
n_chn, n = 2, 256 * 1024
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = pkg.Engine(pkg.KIND_LV2, 96000, n_chn)
x = torch.rand((n_chn, n), device=dev) - 0.5
y = torch.empty_like(x)
rot = np.array([0.0, -1.0, 0.6, 0.8], np.float32)
step = lambda: eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, 0, st)
for _ in range(200):
    step()
torch.cuda.synchronize()
import time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record(ts)
for _ in range(launches):
    step()
e1.record(ts)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("launch period %.2f us by events; host time per call %.2f us" % (e0.elapsed_time(e1) / launches * 1e3, (t1 - t0) / launches * 1e6))
eng.close()
