"""run() from C on ONE stereo instance (96 kHz and 48 kHz), 1024-frame buffers: mean us per call (round 6: relayed against polled doorbell)"""
import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
out = []
for rate, nc in ((96000, 2), (48000, 2), (48000, 4), (48000, 1)):
    g = pkg.Lv2Instance(rate, nc)
    g.activate()
    g.time_lv2_run(1024, 300)
    ts = sorted(g.time_lv2_run(1024, 500) for _ in range(6))
    out.append("%d x%d %.2f" % (rate // 1000, nc, ts[len(ts) // 2]))
    g.close()
print(os.environ.get("PROT_LIB_PATH", "shipped").split("_")[-1], " | ".join(out))
