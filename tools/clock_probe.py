"""Debug: per-workgroup timeline of the p2 kernel (library built with -DPROT_CLOCKDBG=1, see tools/ab_build.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0)
for _ in range(3):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
ev1.record(); torch.cuda.synchronize()
d = y[:4 * 512].view(torch.int32).cpu().numpy().astype(np.int64).reshape(512, 4) & 0xffffffff
cyc, t0, t1, hw = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
base = t0.min()
dur = (t1 - t0) / 100.0
print("%s: %.4f ms/launch; span of all WGs %.1f us; WG duration min/med/max %.1f/%.1f/%.1f us; start spread %.1f us; clock min/med/max %.0f/%.0f/%.0f MHz" % (
    os.environ.get("PROT_LIB_PATH", "main")[-14:], ev0.elapsed_time(ev1) / 5, (t1.max() - base) / 100.0, dur.min(), np.median(dur), dur.max(),
    (t0.max() - base) / 100.0, (cyc / dur).min(), np.median(cyc / dur), (cyc / dur).max()))
xcc = (hw >> 28) & 0xf; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
for xc in range(8):
    m = xcc == xc
    if m.any():
        print("  xcc %d: %3d WGs, duration med %.1f max %.1f us, end max %.1f us, distinct (se,sh,cu) %d" % (
            xc, m.sum(), np.median(dur[m]), dur[m].max(), (t1[m].max() - base) / 100.0, len(set(zip(se[m], sh[m], cu[m])))))
simd = (hw >> 4) & 3; wslot = hw & 0xf
print("  by wave slot:", {int(w): round(float(np.median(dur[wslot == w])), 1) for w in sorted(set(wslot))})
print("  by se:", {int(w): round(float(np.median(dur[se == w])), 1) for w in sorted(set(se))})
print("  by cu:", {int(w): round(float(np.median(dur[cu == w])), 1) for w in sorted(set(cu))})
key = xcc * 4096 + se * 256 + sh * 16 + cu
pairs = {}
for i in range(512):
    pairs.setdefault(int(key[i]), []).append(i)
sizes = {}
for k, v in pairs.items():
    sizes[len(v)] = sizes.get(len(v), 0) + 1
print("  WGs per (xcc,se,sh,cu) key histogram:", sizes)
two = [(dur[v[0]], dur[v[1]], v[0], v[1]) for v in pairs.values() if len(v) == 2]
if two:
    a = np.array([[p[0], p[1]] for p in two])
    print("  pairs on one CU: corr of durations %.2f; mean |diff| %.1f us; example pairs (wg ids):" % (np.corrcoef(a[:, 0], a[:, 1])[0, 1], np.abs(a[:, 0] - a[:, 1]).mean()), [(p[2], p[3]) for p in two[:6]])
order = np.argsort(dur)
print("  slowest WGs:", [(int(i), round(float(dur[i]), 1), int(xcc[i]), int(se[i]), int(cu[i])) for i in order[-8:]])
print("  fastest WGs:", [(int(i), round(float(dur[i]), 1), int(xcc[i]), int(se[i]), int(cu[i])) for i in order[:8]])
print("  duration deciles:", [round(float(q), 1) for q in np.percentile(dur, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])])
