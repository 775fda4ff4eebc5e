"""In-kernel clock of the generated loop (diagnostic libraries built with -DPROT_SV_CLOCK: tools/ab_build.py sv:<tag>:-DPROT_SV_CLOCK[,INC=<variant>]):
`python tools/sv_clock.py tag1 tag2 ...` -> shader clock = cycles / (100 MHz ticks) while the loop runs, cycles per block."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, os, ctypes
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
e = pkg.Engine(pkg.KIND_LV2, 48000, 1)
for _ in range(300):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS)
e.sync()
a = (ctypes.c_ulonglong * 4)(); lib.prot_debug_sv_clock(a); a0 = list(a)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS)
ev1.record(); torch.cuda.synchronize()
lib.prot_debug_sv_clock(a); d = [a[i] - a0[i] for i in range(4)]
print("RESULT ms/launch %%.4f  clock %%.3f GHz  cycles/block/wg %%.0f  runs %%d blocks %%d" %% (ev0.elapsed_time(ev1) / 50, d[0] / d[1] * 0.1, d[0] / d[2], d[3], d[2]))
""" % ROOT
for t in sys.argv[1:]:
    env = dict(os.environ)
    env["PROT_LIB_PATH"] = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants", "libprot_hip_%s.so" % t)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    print("%-14s %s" % (t, line[0][7:] if line else "FAILED " + p.stdout[-1500:]), flush=True)
