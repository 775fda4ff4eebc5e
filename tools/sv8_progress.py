"""Progress of every workgroup of one headline launch (olasave8k_kernel<13, 0>) through its run of blocks: ONE 100 MHz stamp per
block at the end of the generated loop ("tick" variant: python3 tools/gen/gen_olasave8k.py --variant tick --opts tick;
python3 tools/ab_build.py sv8:tick:-DPROT_SV8_STAMPS; PROT_LIB_PATH=.../libprot_hip_tick.so python3 tools/sv8_progress.py [out.json]).
Prints the chip's throughput (blocks finished per microsecond) over the launch in bins, by dispatch rank, and the time a block
takes a workgroup while its CU's other workgroup is still resident / after it has left."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
if not hasattr(lib, "prot_debug_sv8_prog"):
    sys.exit("not a tick library (PROT_LIB_PATH=.../libprot_hip_tick.so)")
lib.prot_debug_sv8_prog.argtypes = [C.POINTER(C.c_uint), C.c_int]
lib.prot_debug_sv8_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0)
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
pr = (C.c_uint * (64 * 2048))()
lib.prot_debug_sv8_prog(pr, 2048)
prog = np.array(list(pr), dtype=np.int64).reshape(2048, 64)
ok = (w[:, 0] > 0) & (w[:, 2] > w[:, 0]) & (w[:, 0] > w[:, 0].max() - 100000000)
bid = np.arange(2048)[ok]
w, prog = w[ok], prog[ok]
t0 = w[:, 0].min()
ent, end = (w[:, 0] - t0) / 100.0, (w[:, 2] - t0) / 100.0
hw = w[:, 3] & 0xffffffff
xcc = (w[:, 3] >> 32) & 0xf
cu_key = xcc * 65536 + ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1)   # (xcc, se, sh, cu)
rk = bid // 256
T = end.max()
print("tick build: %.4f ms per launch by events; %d workgroups, last end %.1f us" % (ms, len(w), T))
# block end times per workgroup (us after the first entry): lane k = the block with k blocks left
ends = []
for i in range(len(w)):
    lanes = prog[i]
    tt = []
    for k in range(63, 0, -1):
        if lanes[k]:
            d = (int(lanes[k]) - int(t0 & 0xffffffff)) & 0xffffffff
            if d < 100000000:
                tt.append(d / 100.0)
    ends.append(np.array(sorted(tt)))
nb = np.array([len(t) for t in ends])
print("  blocks per workgroup by rank: " + ", ".join("rank %d: %d..%d" % (k, nb[rk == k].min(), nb[rk == k].max()) for k in range(int(rk.max()) + 1)))
allb = np.concatenate(ends)
res = {"ms_per_launch": ms, "workgroups": len(w), "last_end_us": float(T), "bins": []}
binw = 20.0
print("  throughput over the launch (blocks finished per us; resident workgroups at the bin's middle):")
for b0 in np.arange(0, T, binw):
    m = (allb >= b0) & (allb < b0 + binw)
    by = [int(((ends_k >= b0) & (ends_k < b0 + binw)).sum()) for ends_k in (np.concatenate([ends[i] for i in range(len(w)) if rk[i] == k]) for k in range(int(rk.max()) + 1))]
    live = int(((ent <= b0 + binw / 2) & (end > b0 + binw / 2)).sum())
    res["bins"].append({"t_us": float(b0), "blocks": int(m.sum()), "by_rank": by, "resident": live})
    print("    %4.0f..%4.0f us: %5.1f blocks/us (rank 0 %4d, rank 1 %4d blocks)  %3d workgroups resident" % (b0, b0 + binw, m.sum() / binw, by[0], by[1] if len(by) > 1 else 0, live))
# block time of a workgroup while its CU neighbour is resident / after it has left
per_cu = {}
for i in range(len(w)):
    per_cu.setdefault(int(cu_key[i]), []).append(i)
both, solo = {0: [], 1: []}, {0: [], 1: []}
for k, ids in per_cu.items():
    if len(ids) != 2:
        continue
    for a, b in ((ids[0], ids[1]), (ids[1], ids[0])):
        d = np.diff(np.concatenate([[ent[a] + 2.5], ends[a]]))
        for j, t_end in enumerate(ends[a]):
            (both if t_end <= end[b] else solo)[int(rk[a])].append(d[j])
res["block_us"] = {}
for name, grp in (("both resident", both), ("alone on its CU", solo)):
    for k in (0, 1):
        if grp[k]:
            res["block_us"]["%s, rank %d" % (name, k)] = {"n": len(grp[k]), "median": float(np.median(grp[k])), "mean": float(np.mean(grp[k]))}
            print("  block time, %s, rank %d: median %.2f us, mean %.2f us (%d blocks)" % (name, k, np.median(grp[k]), np.mean(grp[k]), len(grp[k])))
print("  CUs with two workgroups: %d of %d" % (sum(1 for v in per_cu.values() if len(v) == 2), len(per_cu)))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
