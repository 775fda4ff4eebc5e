"""Per-workgroup timeline of one headline launch of olasave_kernel (diagnostic library built with -DPROT_SV_CLOCK):
`python tools/sv_timeline.py <tag> [PROT_RANK_W=...]` -> when the workgroups start, enter and leave the generated loop, and end."""
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
nw = 1024
a = (ctypes.c_ulonglong * (4 * nw))(); lib.prot_debug_sv_wg(a, nw)
w = np.array(list(a), dtype=np.float64).reshape(nw, 4) * 0.01   # us
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
w -= t0
print("workgroups %%d; starts %%.1f..%%.1f us; kernel span %%.1f us" %% (len(w), w[:, 0].min(), w[:, 0].max(), w[:, 3].max()))
pre = w[:, 1] - w[:, 0]; loop = w[:, 2] - w[:, 1]; post = w[:, 3] - w[:, 2]
print("before the loop: median %%.1f max %%.1f us;  loop: min %%.1f median %%.1f max %%.1f us;  after: median %%.1f max %%.1f us" %% (
    np.median(pre), pre.max(), loop.min(), np.median(loop), loop.max(), np.median(post), post.max()))
ends = np.sort(w[:, 3])
print("ends (us): 1%%%% %%.1f  10%%%% %%.1f  50%%%% %%.1f  90%%%% %%.1f  99%%%% %%.1f  max %%.1f" %% tuple(ends[[int(len(ends) * f) for f in (0.01, 0.1, 0.5, 0.9, 0.99)]].tolist() + [ends[-1]]))
for r in range(4):
    sel = w[r * 256:(r + 1) * 256]
    if len(sel):
        print("  ids %%4d..%%4d: start %%.1f  loop %%.1f  end median %%.1f max %%.1f" %% (r * 256, r * 256 + len(sel) - 1, np.median(sel[:, 0]), np.median(sel[:, 2] - sel[:, 1]), np.median(sel[:, 3]), sel[:, 3].max()))
last = np.argsort(w[:, 3])[-5:]
print("last five:", [(int(i), round(float(w[i, 0]), 1), round(float(pre[i]), 1), round(float(loop[i]), 1), round(float(w[i, 3]), 1)) for i in last])
""" % ROOT
tag = sys.argv[1]
env = dict(os.environ)
for kv in sys.argv[2:]:
    k, _, v = kv.partition("=")
    env[k] = v
env["PROT_LIB_PATH"] = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants", "libprot_hip_%s.so" % tag)
p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(p.stdout[-3000:])
