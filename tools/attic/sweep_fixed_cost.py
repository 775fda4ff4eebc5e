"""Fixed cost of one resident sweep pass (reset + sweep_dev over n frames, stereo, blksiz 2048) by size: what a shard pays however
small it is.  python3 tools/sweep_fixed_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
P, nc = 2048, 2
nmax = 345600000 // 8 // P * P
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand(nc * nmax, generator=g, device=dev) - 0.5) * 0.05
t = torch.arange(nmax, device=dev, dtype=torch.float32) / 96000.0
x[:nmax] += 0.5 * torch.sin(2 * np.pi * 110.0 * t); x[nmax:] += 0.4 * torch.sin(2 * np.pi * 137.5 * t)
eng = pkg.Engine(pkg.KIND_CLI, P, nc)
pk = torch.zeros(nc * 360, device=dev)
for n in (P * 4, P * 64, P * 1024, P * 4096, nmax // 4 // P * P, nmax):
    def one():
        eng.reset(); pk.zero_()
        eng.sweep_dev(x.data_ptr(), nmax, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
    for _ in range(3): one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(ts)
    for _ in range(10): one()
    e1.record(ts); torch.cuda.synchronize()
    print("n = %9d frames per channel: %.3f ms per pass by events, %.3f ms wall" % (n, e0.elapsed_time(e1) / 10, (time.perf_counter() - t0) * 100), flush=True)
eng.close()
