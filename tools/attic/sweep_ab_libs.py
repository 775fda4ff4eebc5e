#!/usr/bin/env python3
"""cfg4 pass times (harmonic, noise; blksiz 2048 or $PROT_AB_BLKSIZ) of several libraries, interleaved in child processes on ONE box:
    python3 tools/sweep_ab_libs.py main nopipe ...      ("main" = the shipped library; others = lib/variants/libprot_hip_<tag>.so)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda", 0)
P, n_chn = int(os.environ.get("PROT_AB_BLKSIZ", "2048")), 2   # (PROT_AB_BLKSIZ=16384: the long-FIR route)
total = 3600 * 96000 // P * P
out = []
for kind in ("harmonic", "noise"):
    x = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, kind)
    eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    peaks = torch.zeros((n_chn, 360), device=dev)
    ts = torch.cuda.Stream(device=dev)   # (a stream of the caller's: stream 0 means the engine's own, which torch events do not bracket)
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    def one():
        eng.reset(); peaks.zero_()
        eng.sweep_dev(x.data_ptr(), total, total, 0, 360, 1, -1, peaks.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
    for _ in range(40): one()   # (the pass time settles over ~50 ms of continuous work: tools/sweep_drift.py)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): one()
    e1.record(); torch.cuda.synchronize()
    out.append("%%s %%.3f" %% (kind, e0.elapsed_time(e1) / 5))
    eng.close(); del x
print("RESULT " + " ".join(out))
''' % ROOT
tags = sys.argv[1:] or ["main"]
for rnd in range(3):
    for tag in tags:
        env = dict(os.environ)
        if tag != "main":
            env["PROT_LIB_PATH"] = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants", "libprot_hip_%s.so" % tag)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        print("%-10s %s" % (tag, [l for l in r.splitlines() if l.startswith("RESULT")][-1][7:] if "RESULT" in r else "FAILED"), flush=True)
