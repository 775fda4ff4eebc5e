"""Latency of the LV2 run() drop-in on host buffers: default (round 4: the resident run() server for calls of up to one transform
block per channel, a launch on the zero-copy host block beyond) vs staged (test option LV2_STAGED), per buffer size; the server
against the launch route: tools/server_probe.py.
Each variant in its own process; median of 2000 calls after 200 warm-up calls."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
pkg.set_test_option(pkg.TEST_LV2_STAGED, int(sys.argv[1]))
for nchn, bs in ((1, 64), (1, 256), (1, 1024), (2, 1024), (1, 8192)):
    g = pkg.Lv2Instance(48000, nchn); g.activate()
    x = (np.random.default_rng(3).random((nchn, bs), dtype=np.float32) - 0.5)
    for _ in range(200): g.run(x, 90.0)
    ts = []
    for _ in range(2000):
        t0 = time.perf_counter(); g.run(x, 90.0); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("  chn %%d buf %%5d: median %%.1f us  p10 %%.1f  p99 %%.1f" %% (nchn, bs, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 99)), flush=True)
    g.close()
''' % ROOT
for name, staged in (("default (zero-copy up to 4096 samples per call)", 0), ("staged", 1),
                     ("default (zero-copy up to 4096 samples per call)", 0), ("staged", 1)):
    print(name, flush=True)
    subprocess.run([sys.executable, "-c", CHILD, str(staged)])
