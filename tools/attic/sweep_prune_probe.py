"""Sweep time with and without pruning (exact either way) on three signals: the bench's noise, noise + harmonic stack,
and a monotonically growing signal (worst case: nothing can be skipped).  60 s of stereo 96 kHz per pass, 360 angles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
P, nc = 2048, 2
n = int(600 * 96000) // P * P   # 10 minutes
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.rand(nc * n, generator=g, device=dev) - 0.5
t = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
harm = sum(torch.cos(2 * np.pi * k * 110.0 * t + 0.3 * k) / k for k in range(1, 33)).float()
harm = 0.8 * harm / harm.abs().max()
sigs = {"uniform noise": noise,
        "harmonic stack + noise": torch.cat([harm + 0.05 * noise[:n], 0.7 * harm + 0.05 * noise[n:]]),
        "loud first second, then quiet": torch.where((torch.arange(nc * n, device=dev) % n) < 96000, noise, 1e-3 * noise),
        "growing noise": noise * torch.linspace(0.01, 1.0, n, device=dev).repeat(nc),
        "steady tone (worst case)": (0.7 * torch.sin(2 * np.pi * 997.0 * t)).float().repeat(nc)}
for name, x in sigs.items():
    res = {}
    for prune in (1, 0, 1, 0):
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1 - prune)
        e = pkg.Engine(pkg.KIND_CLI, P, nc)
        pk = torch.zeros(nc * 360, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
        torch.cuda.synchronize()
        res.setdefault(prune, []).append((time.perf_counter() - t0) * 1e3)
        res["tab%d" % prune] = pk.cpu().numpy().tobytes()
        ev, seen = e.sweep_stats()
        res["frac%d" % prune] = ev / max(seen, 1)
        res["ex%d" % prune] = e.sweep_stats_ex()
        e.close()
    print("%-28s pruned %s ms (evaluated %.1f %% of the groups)   unpruned %s ms (%.0f %%)   tables identical: %s" % (
        name, ["%.2f" % v for v in res[1]], 100 * res["frac1"], ["%.2f" % v for v in res[0]], 100 * res["frac0"],
        res["tab1"] == res["tab0"]), flush=True)
    ex = res["ex1"]
    print("    untested %.1f %% of the tiles, survivors %.1f %% of the tested frames" % (
        100.0 * ex["untested"] / max(ex["seen"], 1), 100.0 * ex["survivors"] / max(32 * (ex["seen"] - ex["untested"]), 1)), flush=True)
