"""Where a wave of the headline 8192-point kernel (olasave8k_kernel<13, 0>) spends a block: cycles parked at each of the four
workgroup barriers and at each wait, per wave and block, from shader-clock stamps inside the generated loop; and the per-workgroup
timeline of one launch (entry / end times by dispatch rank and XCD: how much of the launch the wave slots stand empty).

Needs the stamped variant library:
    python3 tools/gen/gen_olasave8k.py --variant stamps --opts stamps
    python3 tools/ab_build.py sv8:stamps:-DPROT_SV8_STAMPS
    PROT_LIB_PATH=phaserotate.lv2_amd/lib/variants/libprot_hip_stamps.so python3 tools/sv8_stamps.py [out.json]
(a stamp is an s_memtime, which returns through lgkmcnt and drains the LDS queue: the stamped loop is a few percent slower than
the shipped one -- the tool prints both times when PROT_REF_MS is given)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen"))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
if not hasattr(lib, "prot_debug_sv8_stamps"):
    sys.exit("not a stamped library (PROT_LIB_PATH=.../libprot_hip_stamps.so)")
lib.prot_debug_sv8_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
lib.prot_debug_sv8_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
from gen_olasave8k import Gen8k
NAMES = Gen8k.STAMP_NAMES

dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0)
call = lambda: e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
for _ in range(200):
    call()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.prot_debug_sv8_stamps(buf, 32, 1)
reps = 20   # (the slots keep the LAST launch: every wave overwrites its own)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    call()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
lib.prot_debug_sv8_stamps(buf, 32, 0)
v = np.array(list(buf), dtype=np.float64)
wave_blocks, waves = v[16], v[17]
tot = v[1:15].sum()
reps_ = 1
res = {"kernel": "olasave8k_kernel<13, 0>", "ms_per_launch": ms, "ref_ms_shipped": float(os.environ.get("PROT_REF_MS", 0)) or None,
       "wave_blocks_per_launch": wave_blocks, "waves_per_launch": waves}
if tot > 0:   # (a timeline-only library -- -DPROT_SV8_TIMELINE around the shipped loop -- has no stamps)
    res["variant"] = "stamped loop"
    res["cycles_per_wave_block"] = tot / wave_blocks
    res["intervals"] = []
    print("stamped build: %.4f ms per launch; %.0f waves, %.2f blocks per wave; %.0f cycles per wave-block (sum of the intervals)" % (
        ms, waves, wave_blocks / waves, tot / wave_blocks))
    wait = 0.0
    for i in range(1, 15):
        cyc = v[i] / wave_blocks
        kind = "barrier" if NAMES[i].startswith("barrier") else "wait" if NAMES[i].startswith("wait") else "work"
        if kind != "work":
            wait += cyc
        res["intervals"].append({"i": i, "what": NAMES[i], "kind": kind, "cycles": cyc, "share": v[i] / tot})
        print("  %2d %-82s %8.0f cycles %5.1f %%" % (i, NAMES[i], cyc, 100 * v[i] / tot))
    res["parked_at_barriers_and_vmcnt_share"] = wait * wave_blocks / tot
    print("  parked at the four barriers and the row wait: %.1f %% of a wave's block" % (100 * wait * wave_blocks / tot))
else:
    res["variant"] = "shipped loop, timeline only"
    print("timeline build (shipped loop): %.4f ms per launch" % ms)

# the timeline of the LAST launch
wt = (C.c_ulonglong * (4 * 2048))()
lib.prot_debug_sv8_wg(wt, 2048)
w = np.array(list(wt), dtype=np.int64).reshape(2048, 4)
ok = (w[:, 0] > 0) & (w[:, 2] > w[:, 0]) & (w[:, 0] > w[:, 0].max() - 100000000)   # (workgroups without a run return at once)
bid = np.arange(2048)[ok]
w = w[ok]
nwg = len(w)
t0 = w[:, 0].min()
ent, loop, end = (w[:, 0] - t0) / 100.0, (np.maximum(w[:, 1], w[:, 0]) - t0) / 100.0, (w[:, 2] - t0) / 100.0
xcc = (w[:, 3] >> 32) & 0xf
rk = bid // 256
T = end.max()
print("timeline of one launch: %d workgroups; last end %.1f us after the first entry (launch by events %.1f us)" % (nwg, T, ms * 1e3))
res["timeline"] = {"workgroups": nwg, "last_end_us": T, "by_rank": []}
for k in range(int(rk.max()) + 1):
    m = rk == k
    row = {"rank": k, "entry_us": [float(ent[m].min()), float(np.median(ent[m])), float(ent[m].max())],
           "prologue_us_median": float(np.median(loop[m] - ent[m])),
           "end_us": [float(end[m].min()), float(np.median(end[m])), float(end[m].max())]}
    res["timeline"]["by_rank"].append(row)
    print("  rank %d: entry %.1f / %.1f / %.1f us (min / median / max), to the loop %.1f us, end %.1f / %.1f / %.1f us" % (
        k, *row["entry_us"], row["prologue_us_median"], *row["end_us"]))
idle = (ent.sum() + (T - end).sum()) / (nwg * T)
res["timeline"]["slot_time_empty_share"] = idle
res["timeline"]["mean_busy_us"] = float((end - ent).mean())
print("  workgroup-slot time standing empty (before entry + after end): %.1f %% of slots x launch; mean busy time of a workgroup %.1f us" % (100 * idle, (end - ent).mean()))
cnt = np.zeros(8)
for xx in range(8):
    cnt[xx] = (xcc == xx).sum()
print("  workgroups per XCD:", cnt.astype(int).tolist(), "; blockIdx %% 8 == xcc for %d of %d" % (int(((bid % 8) == xcc).sum()), nwg))
dec = np.percentile(end, [0, 10, 25, 50, 75, 90, 100])
res["timeline"]["end_percentiles_us"] = dec.tolist()
print("  end-time percentiles 0/10/25/50/75/90/100: " + " ".join("%.0f" % d for d in dec))
byx = []
for xx in range(8):
    m = xcc == xx
    if m.any():
        byx.append({"xcc": xx, "n": int(m.sum()), "end_us_median": float(np.median(end[m])), "end_us_max": float(end[m].max())})
res["timeline"]["by_xcc"] = byx
print("  by XCD (median / max end): " + "  ".join("x%d %.0f/%.0f" % (b["xcc"], b["end_us_median"], b["end_us_max"]) for b in byx))
res["output_checksum"] = float(y.double().abs().sum().item())
print("  checksum of the output %.6f" % res["output_checksum"])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
