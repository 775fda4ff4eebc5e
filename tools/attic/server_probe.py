#!/usr/bin/env python3
"""The run() server on the hardware: same bits as the launch route, latency, idle exit and restart, a device-wide synchronisation
beside it, two instances.  timeout 300 python3 tools/server_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
rng = np.random.default_rng(5)


def series(nchn, bs, n, rate=48000, angles=(90.0,)):
    xs = [(rng.random((nchn, bs), dtype=np.float32) - 0.5) for _ in range(n)]
    outs = {}
    for mode in (1, 0):
        pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, mode)
        g = pkg.Lv2Instance(rate, nchn)
        g.activate()
        o = [g.run(x, angles[i % len(angles)]).copy() for i, x in enumerate(xs)]
        outs[mode] = (np.stack(o), g.server_stats())
        g.close()
    pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, 0)
    same = outs[0][0].tobytes() == outs[1][0].tobytes()
    print("chn %d buf %4d rate %d angles %s: server == launch: %s; server stats %s (launch route %s)" %
          (nchn, bs, rate, angles, same, outs[0][1], outs[1][1]), flush=True)
    return same


ok = True
ok &= series(1, 1024, 50)
ok &= series(2, 1024, 50)
ok &= series(1, 64, 100)
ok &= series(2, 256, 60, rate=96000)
ok &= series(1, 1024, 60, angles=(90.0, 45.0, 45.0, 10.0, -120.0))   # ramps
ok &= series(1, 6000, 10)                                             # more than one block: the launch route
# latency
for nchn, bs in ((1, 1024), (2, 1024), (1, 64)):
    for mode in (0, 1):
        pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, mode)
        g = pkg.Lv2Instance(48000, nchn)
        g.activate()
        x = (rng.random((nchn, bs), dtype=np.float32) - 0.5)
        for _ in range(200):
            g.run(x, 90.0)
        ts = []
        for _ in range(2000):
            t0 = time.perf_counter(); g.run(x, 90.0); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print("chn %d buf %4d %s: median %.1f us  p10 %.1f  p99 %.1f  %s" % (nchn, bs, "launch" if mode else "server", np.median(ts),
              np.percentile(ts, 10), np.percentile(ts, 99), g.server_stats()), flush=True)
        g.close()
pkg.set_test_option(pkg.TEST_LV2_NO_SERVER, 0)
# idle exit and restart
g = pkg.Lv2Instance(48000, 1); g.activate()
x = (rng.random((1, 1024), dtype=np.float32) - 0.5)
g.run(x, 90.0)
time.sleep(0.6)
t0 = time.perf_counter(); g.run(x, 90.0); dt = time.perf_counter() - t0
print("after 0.6 s idle: run() took %.1f us, stats %s (a second start expected)" % (dt * 1e6, g.server_stats()), flush=True)
# a device-wide synchronisation beside a live server: ends when the server leaves by itself (<= 0.25 s)
import torch
g.run(x, 90.0)
t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("torch.cuda.synchronize() beside a live server: %.3f s" % dt, flush=True)
# other streams beside a live server: do their kernels run? (hardware-queue sharing)
g.run(x, 90.0)
worst = 0.0
for i in range(12):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        t0 = time.perf_counter()
        a = torch.zeros(1024, device="cuda") + 1
        st.synchronize()
        worst = max(worst, time.perf_counter() - t0)
    g.run(x, 90.0)
print("12 other streams beside a live server: slowest kernel + sync %.4f s" % worst, flush=True)
# two instances: the second takes the launch route
h = pkg.Lv2Instance(48000, 1); h.activate()
for _ in range(5):
    g.run(x, 90.0); h.run(x, 45.0)
print("two instances: %s / %s" % (g.server_stats(), h.server_stats()), flush=True)
g.close(); h.close()
print("ALL SAME" if ok else "MISMATCH")
