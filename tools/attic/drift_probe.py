"""Diagnostic: does the headline launch time drift under sustained back-to-back launches?  Groups of 5 launches timed with
events on the launch stream, `groups` groups in a row without a pause; then the same after a 2 s pause.
usage: python tools/drift_probe.py [groups] [gen]   (gen: generate x with a seeded torch.Generator as bench.py does)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
n = 1 << 27
if len(sys.argv) > 2 and sys.argv[2] == "gen":
    gen = torch.Generator(device=dev).manual_seed(0x9E3779B9)
    x = torch.rand(n, generator=gen, device=dev, dtype=torch.float32) - 0.5
else:
    x = torch.rand(n, device=dev) - 0.5
y = torch.empty_like(x)
th = -2.0 * np.pi * 90.0 / 360.0
rot = np.array([np.cos(th), np.sin(th)], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1)
def run(tag):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    evs[0].record(ts)
    for g in range(groups):
        for _ in range(5):
            e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        evs[g + 1].record(ts)
    torch.cuda.synchronize()
    ms = [evs[g].elapsed_time(evs[g + 1]) / 5 for g in range(groups)]
    print(tag, " ".join("%.3f" % m for m in ms))
run("cold :")
run("again:")
time.sleep(2.0)
run("pause:")
