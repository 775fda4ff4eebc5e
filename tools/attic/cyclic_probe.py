#!/usr/bin/env python3
"""TIMING EXPERIMENT (round 4, last day): the headline launch with the resident workgroups walking the stream as ONE dense window
(workgroup b takes blocks b, b + 512, ...: generator option "cyclic" + -DPROT_SV8_CYCLIC; results wrong -- the context rows
are not reloaded) against the shipped run-per-workgroup order, complete and with arithmetic and LDS traffic removed.
    python3 tools/gen/gen_olasave8k.py --variant cyc --opts cyclic; ... --variant cycmem --opts cyclic,nolds,novalu; ... --variant runmem --opts nolds,novalu
    python3 tools/ab_build.py sv8:cyc:-DPROT_SV8_CYCLIC sv8:cycmem:-DPROT_SV8_CYCLIC sv8:runmem
    python3 tools/cyclic_probe.py           (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); st = ts.cuda_stream
n = 1 << 27
x = torch.rand(n, device=dev) - 0.5
y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
eng = pkg.Engine(pkg.KIND_LV2, 48000, 1)
def step(): eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
for _ in range(300): step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ts)
for _ in range(40): step()
e1.record(ts); torch.cuda.synchronize()
print("RESULT %%.4f" %% (e0.elapsed_time(e1) / 40))
''' % ROOT
blocks, per, grid = (1 << 27) // 13312, ((1 << 27) // 13312 - 8) // 512, -(-((1 << 27) // 13312 + 1) // 20)
norm = {"cyc": blocks / (grid * per), "cycmem": blocks / (grid * per)}
for rnd in range(3):
    for tag in ("main", "cyc", "runmem", "cycmem"):
        env = dict(os.environ)
        if tag != "main":
            env["PROT_LIB_PATH"] = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants", "libprot_hip_%s.so" % tag)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
        ms = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if not ms:
            print(tag, "FAILED", r.stderr[-300:]); continue
        f = norm.get(tag, 1.0)
        print("%-8s %.4f ms%s" % (tag, ms[0] * f, "  (measured %.4f on %d of %d blocks)" % (ms[0], grid * per, blocks) if f != 1.0 else ""), flush=True)
