"""How a headline launch ENDS, by workgroup and by CU (timeline build: tools/ab_build.py sv8:tl:-DPROT_SV8_TIMELINE or the tick build;
PROT_LIB_PATH=.../libprot_hip_tl.so python3 tools/sv8_dyn_tail.py [out.json] [engine flags]): entry / end stamps of every workgroup
(100 MHz), the CU it ran on (HW_ID), percentiles of the end times by dispatch rank and by CU, and the share of workgroup-slot time
that stands empty between the first entry and the last end."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
if not hasattr(lib, "prot_debug_sv8_wg"):
    sys.exit("not a timeline library")
lib.prot_debug_sv8_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
e = pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=flags)
call = lambda: e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
for _ in range(300):
    call()
torch.cuda.synchronize()
reps = 20
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    call()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
wt = (C.c_ulonglong * (4 * 2048))()
lib.prot_debug_sv8_wg(wt, 2048)
w = np.array(list(wt), dtype=np.int64).reshape(2048, 4)
ok = (w[:, 0] > 0) & (w[:, 2] > w[:, 0]) & (w[:, 0] > w[:, 0].max() - 100000000)
bid = np.arange(2048)[ok]
w = w[ok]
t0 = w[:, 0].min()
ent, loop, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0
hw = w[:, 3] & 0xffffffff
xcc = (w[:, 3] >> 32) & 0xf
cu_key = xcc * 65536 + ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1)
nwg = len(w)
rk = (bid >= nwg // 2).astype(int)
T = end.max()
pc = lambda a: [round(float(v), 1) for v in np.percentile(a, [0, 10, 25, 50, 75, 90, 100])]
res = {"ms_per_launch_events": ms, "workgroups": nwg, "last_end_us": float(T), "end_percentiles_us": {}, "flags": flags}
print("launch by events %.4f ms; %d workgroups; last end %.1f us after the first entry" % (ms, nwg, T))
for k in (0, 1):
    m = rk == k
    if m.any():
        res["end_percentiles_us"]["rank %d" % k] = pc(end[m])
        print("  rank %d: entry median %.2f us, to the loop %.2f us, end 0/10/25/50/75/90/100 %% = %s" % (k, np.median(ent[m]), np.median(loop[m][loop[m] > 0] - ent[m][loop[m] > 0]) if (loop[m] > 0).any() else -1, pc(end[m])))
per_cu = {}
for i in range(nwg):
    per_cu.setdefault(int(cu_key[i]), []).append(i)
cu_end = np.array([max(end[i] for i in v) for v in per_cu.values()])
cu_first = np.array([min(end[i] for i in v) for v in per_cu.values() if len(v) == 2])
cu_gap = np.array([abs(end[v[0]] - end[v[1]]) for v in per_cu.values() if len(v) == 2])
res["cu_end_percentiles_us"] = pc(cu_end)
res["cu_first_end_percentiles_us"] = pc(cu_first)
res["cu_alone_us_percentiles"] = pc(cu_gap)
print("  CUs: %d (two workgroups on %d); a CU's LAST end: %s" % (len(per_cu), (np.array([len(v) for v in per_cu.values()]) == 2).sum(), pc(cu_end)))
print("  a CU's FIRST end: %s; time one workgroup works on alone: %s" % (pc(cu_first), pc(cu_gap)))
empty = (ent.sum() + (T - end).sum()) / (nwg * T)
cu_idle = (T - cu_end).sum() / (len(cu_end) * T)
res["slot_time_empty_share"] = float(empty)
res["cu_idle_share"] = float(cu_idle)
print("  workgroup-slot time standing empty: %.1f %% of slots x launch; CUs wholly idle (after their last end): %.1f %%; one workgroup alone: %.1f %% of CU time"
      % (100 * empty, 100 * cu_idle, 100 * cu_gap.sum() / (len(cu_end) * T)))
res["cu_alone_share"] = float(cu_gap.sum() / (len(cu_end) * T))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
