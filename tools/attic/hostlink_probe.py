import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
n, nc, P = 1 << 24, 2, 8192
x = (np.random.default_rng(5).random(n * nc, dtype=np.float32) - 0.5)
out = np.ones_like(x)  # touched
def med(f, k=5):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[k // 2] * 1e3
xt = torch.from_numpy(x)
d = torch.empty(n * nc, device="cuda")
def h2d(): d.copy_(xt); torch.cuda.synchronize()
hb = torch.empty(n * nc)
def d2h(): hb.copy_(d); torch.cuda.synchronize()
xp = xt.pin_memory()
def h2d_pinned(): d.copy_(xp); torch.cuda.synchronize()
print("torch H2D pageable 128 MiB: %.2f ms;  pinned: %.2f ms;  D2H pageable: %.2f ms" % (med(h2d), med(h2d_pinned), med(d2h)))
pr = pkg.PhaseRotate(P, nc)
pr.apply_file(x[: P * nc * 4], [90, -90])
print("apply_file into a pre-touched output: %.2f ms;  into a fresh np.zeros_like each call: %.2f ms" % (
    med(lambda: pr.apply_file(x, [90, -90], out=out)), med(lambda: pr.apply_file(x, [90, -90]))))
def ana():
    pr.reset(); pr.analyze_file(x, 0, 360, 24); pr.peak_all(0)
print("analyze_file (15 angles) + peak read-back: %.2f ms" % med(ana))
def ana360():
    pr.reset(); pr.analyze_file(x, 0, 360, 1); pr.peak_all(0)
print("analyze_file (360 angles) + peak read-back: %.2f ms" % med(ana360))
# does the analysis really see the whole file?  a spike in the LAST block must show up in the raw peak (angle 0)
x2 = x.copy(); x2[-5] = 7.0
pr.reset(); pr.analyze_file(x2, 0, 360, 24)
print("raw peak with a 7.0 spike at the very end of the file: %.3f (expect 7.0)" % pr.peak_all(0))
