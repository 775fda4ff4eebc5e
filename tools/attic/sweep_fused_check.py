"""Fused sweep (Hilbert pass + first test, lists, matrix-core evaluation) against the separate passes and the unpruned kernel:
tables bit for bit, times, how many frames were listed.  python tools/sweep_fused_check.py [seconds of stereo 96 kHz audio] [blksiz]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
nc = 2
n = int(sec * 96000) // P * P
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.rand(nc * n, generator=g, device=dev) - 0.5
t = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
harm = sum(torch.cos(2 * np.pi * k * 110.0 * t + 0.3 * k) / k for k in range(1, 33)).float()
harm = 0.8 * harm / harm.abs().max()
swell = (0.55 + 0.45 * torch.sin(2 * np.pi * t / 37.0)).float()
sigs = {"harmonic stack + swell + noise": torch.cat([harm * swell + 0.05 * noise[:n], 0.7 * harm * swell + 0.05 * noise[n:]]),
        "uniform noise": noise,
        "loud first second, then quiet": torch.where((torch.arange(nc * n, device=dev) % n) < 96000, noise, 1e-3 * noise),
        "steady tone (worst case)": (0.7 * torch.sin(2 * np.pi * 997.0 * t)).float().repeat(nc)}
for name, x in sigs.items():
    res = {}
    for mode in ("fused", "separate", "unpruned", "fused", "separate"):
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1 if mode == "unpruned" else 0)
        pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, 1 if mode == "separate" else 0)
        e = pkg.Engine(pkg.KIND_CLI, P, nc)
        pk = torch.zeros(nc * 360, device=dev)
        e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)   # (warm: buffers allocated)
        torch.cuda.synchronize(); e.reset(); pk.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((time.perf_counter() - t0) * 1e3)
        res["tab_" + mode] = pk.cpu().numpy().tobytes()
        if mode == "fused":
            f = e.sweep_stats_fused()
            res["listed"] = f["listed"] / max(1, f["tested"])
            ev, seen = e.sweep_stats()
            res["ev"] = ev / max(1, seen)
        e.close()
    pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0); pkg.set_test_option(pkg.TEST_SWEEP_UNFUSED, 0)
    print("%-32s fused %s ms (%.1f %% of the frames listed, %.0f %% of their tiles evaluated)  separate %s ms  unpruned %s ms  tables identical: fused/separate %s fused/unpruned %s" % (
        name, ["%.2f" % v for v in res["fused"]], 100 * res["listed"], 100 * res["ev"], ["%.2f" % v for v in res["separate"]], ["%.2f" % v for v in res["unpruned"]],
        res["tab_fused"] == res["tab_separate"], res["tab_fused"] == res["tab_unpruned"]), flush=True)
