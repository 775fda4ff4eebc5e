"""Diagnostic: where a wave of the headline polyphase kernel spends its cycles.
Needs the stamped variant library: `python tools/ab_build.py stamps:-DPROT_STAMPS=1`, then
`PROT_LIB_PATH=phaserotate.lv2_amd/lib/variants/libprot_hip_stamps.so python tools/phase_stamps.py [case]`.
Prints the per-phase average shader cycles per wave-iteration (summed over all waves / wave-iterations)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
lib.prot_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
cases = {"poly": (pkg.KIND_LV2, 48000, 0), "cli": (pkg.KIND_CLI, 2048, 0), "cli1024": (pkg.KIND_CLI, 1024, 0),
         "poly_96k": (pkg.KIND_LV2, 96000, 4)}
name = sys.argv[1] if len(sys.argv) > 1 else "poly"
kind, param, fl = cases[name]
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(kind, param, 1, flags=fl)
for _ in range(30):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.prot_debug_stamps(buf, 32, 1)
reps = 10
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
ev1.record(); torch.cuda.synchronize()
lib.prot_debug_stamps(buf, 32, 0)
v = np.array(list(buf), dtype=np.float64)
wave_it, waves = v[16], v[19]
names = ["0 top: wait loads, DPP shift", "1 fwd pass 1 + twiddle + LDS store", "2 barrier A", "3 reads, pass 2, local exchange",
         "4 H prefetch + pass 3", "5 spectrum multiply", "6 issue next loads + dry", "7 inv pass 1, local xchg, pass 2, store",
         "8 barrier B", "9 reads, twiddle, inv pass 3", "10 OLA + rotate + stores"]
print("%s: %.4f ms per launch (stamped build); %d waves, %.1f iterations per wave; clock %.0f MHz" % (
    name, ev0.elapsed_time(ev1) / reps, waves / reps, wave_it / waves, v[17] / v[18] * 100.0))
print("  per wave: %.0f cycles in the loop+epilogue = %.1f us wall (launch %.1f us); loop phases sum %.0f cycles" % (
    v[17] / waves, v[18] / waves / 100.0, ev0.elapsed_time(ev1) / reps * 1e3, v[:11].sum() / waves))
print("  stealing (per launch): chunks stolen %.0f, failed claims %.0f, victim searches %.0f, searches that found nothing %.0f" % (v[20] / reps, v[21] / reps, v[22] / reps, v[23] / reps))
tot = v[:11].sum()
for i in range(11):
    print("  %-44s %8.0f cycles  %5.1f %%" % (names[i], v[i] / wave_it, 100.0 * v[i] / tot))
print("  %-44s %8.0f cycles per wave-iteration" % ("sum", tot / wave_it))

# residency census: how many workgroups were alive at once, and how they were spread over the CUs
nwg = int(waves / reps / 4)
wt = (C.c_ulonglong * (4 * nwg))()
lib.prot_debug_wg_times.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.prot_debug_wg_times(wt, nwg)
w = np.array(list(wt), dtype=np.int64).reshape(nwg, 4)
t0 = w[:, 0][w[:, 0] > 0].min()
ent, end, hw = (w[:, 0] - t0) / 100.0, (w[:, 2] - t0) / 100.0, w[:, 3]
loop = ent + 7.5
wg_clk = w[:, 1] / np.maximum(end - ent - 7.5, 1.0)  # MHz, roughly (loop cycles / loop wall time)
print("  workgroups %d; entry spread %.1f us (median %.1f); prologue median %.1f us; lifetime median %.1f us; last end %.1f us" % (
    nwg, ent.max(), np.median(ent), np.median(loop - ent), np.median(end - ent), end.max()))
ev = sorted([(x, 1) for x in ent] + [(x, -1) for x in end])
cur = mx = 0
for _, d in ev:
    cur += d; mx = max(mx, cur)
print("  max concurrent workgroups %d; late starters (entry > 20 us): %d" % (mx, int((ent > 20).sum())))
xcc = (hw >> 32) & 0xf; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 4096 + se * 256 + sh * 16 + cu
early = ent <= 20
cnt = {}
for kk in key[early]:
    cnt[int(kk)] = cnt.get(int(kk), 0) + 1
hist = {}
for v_ in cnt.values():
    hist[v_] = hist.get(v_, 0) + 1
print("  first-round workgroups per (xcc,se,sh,cu):", dict(sorted(hist.items())), "distinct CUs", len(cnt))
life = end - ent
print("  lifetime deciles (us):", [round(float(q), 1) for q in np.percentile(life, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])])
per = {}
for i in range(nwg):
    per.setdefault(int(key[i]), []).append(float(end[i]))
for n_on in sorted(set(len(v_) for v_ in per.values())):
    rows = np.array([sorted(v_) for v_ in per.values() if len(v_) == n_on])
    print("  CUs with %d workgroups (%d): mean end time of the 1st..last to finish (us):" % (n_on, len(rows)), [round(float(x), 1) for x in rows.mean(axis=0)],
          " max:", [round(float(x), 1) for x in rows.max(axis=0)])
cuend = {}
for i in range(nwg):
    cuend[int(key[i])] = max(cuend.get(int(key[i]), 0.0), float(end[i]))
ce = np.array(list(cuend.values())); ck = np.array(list(cuend.keys()))
print("  per-CU end time: deciles", [round(float(q), 1) for q in np.percentile(ce, [0, 10, 25, 50, 75, 90, 100])])
for xc in range(8):
    m = (ck // 4096) == xc
    if m.any():
        print("   xcc %d: %2d CUs, CU end mean %.1f max %.1f;  by se:" % (xc, m.sum(), ce[m].mean(), ce[m].max()),
              {int(s_): round(float(ce[m & (((ck // 256) % 16) == s_)].mean()), 1) for s_ in sorted(set((ck[m] // 256) % 16))})
xcc_of = (hw >> 32) & 0xf
print("  shader clock by xcc (MHz, median over workgroups):", {int(x): int(np.median(wg_clk[xcc_of == x])) for x in sorted(set(xcc_of))})
print("  loop cycles by xcc (k, median):", {int(x): int(np.median(w[:, 1][xcc_of == x]) / 1000) for x in sorted(set(xcc_of))})
bid = np.arange(nwg)
live = w[:, 2] > 0
print("  by blockIdx % 8 (= the XCD a workgroup is dealt to): xcc id, median end us by dispatch rank 0/1/2, max end")
for x in range(8):
    m = live & ((bid % 8) == x)
    rk = np.minimum(bid // 256, 2)
    print("   %d: xcc %s  end %s  max %.1f  loop kcycles %s" % (x, sorted(set(int(v_) for v_ in xcc_of[m])), [round(float(np.median(end[m & (rk == k_)])), 1) for k_ in range(3) if (m & (rk == k_)).any()],
          end[m].max(), [int(np.median(w[:, 1][m & (rk == k_)]) / 1000) for k_ in range(3) if (m & (rk == k_)).any()]))
order = np.argsort(end)
print("  blockIdx of the 12 last workgroups to finish:", [int(i) for i in order[-12:]], "first:", [int(i) for i in order[:12]])
print("  correlation(blockIdx, end) %.2f" % np.corrcoef(np.arange(nwg), end)[0, 1])

w2 = (C.c_ulonglong * (2 * nwg))()
if lib.prot_debug_wg_times(w2, -nwg) == 0:
    a = np.array(list(w2), dtype=np.int64).reshape(nwg, 2)
    own = a[:, 0] & 0xffffffff; stolen = a[:, 0] >> 32; thief_us = a[:, 1] / 100.0
    if stolen.sum() > 0:
        done = end - thief_us
        print("  owner phase ends (us): deciles", [round(float(q), 1) for q in np.percentile(done, [0, 10, 25, 50, 75, 90, 100])])
        print("  chunks claimed by owners: histogram", {int(k_): int((own == k_).sum()) for k_ in sorted(set(own))})
        print("  chunks taken elsewhere per workgroup: histogram", {int(k_): int((stolen == k_).sum()) for k_ in sorted(set(stolen))})
        print("  time as a thief (us): median %.1f max %.1f; per stolen chunk %.1f us" % (np.median(thief_us), thief_us.max(), thief_us.sum() / max(1, stolen.sum())))
        slow = np.argsort(done)[-20:]
        print("  the 20 slowest owners: claims", [int(own[i]) for i in slow], "owner-phase end", [round(float(done[i])) for i in slow])
