#!/usr/bin/env python3
"""profiles/<tag>_* from the raw rocprofv3 output of tools/profile_round.sh:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extra --no-sweep --no-cpu-baseline` (headline only)
  <tag>_kernel_stats_full.csv  the same of the full default bench (`python3 bench.py --no-cpu-baseline`)
  <tag>_bench.json         the JSON line that run printed
  <tag>_pmc.json           FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md section HBM) + WRITE_SIZE per launch of the headline kernel
  <tag>_pmc_sq.json        SQ counters per launch: headline kernel and sweep kernel
  traffic_latest.json      read by bench.py (roofline.traffic)"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
tag, out = sys.argv[1], sys.argv[2]
HEAD, SWEEP, SWEEP_PLAIN = "olasave8k_dyn_kernel<13, 0>", "sweep_eval_kernel<3>", "sweep_kernel<3, false>"
SWEEP_FUSED = "olasave_kernel<12, 3>"  # the Hilbert pass with the sweep's test in its epilogue (cfg4: 96 kHz, blksiz 2048)


def counters(d):
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def pick(agg, sub):
    res = {}
    for k, cs in agg.items():
        if sub in k:
            for c, v in cs.items():
                res[c] = {"launches": len(v), "mean": sum(v) / len(v)}
            res["_kernel"] = k[:160]
    return res


st = glob.glob(os.path.join(out, "stats", "**", "*_kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(PROF, tag + "_kernel_stats.csv"))
    for line in open(st[0]).read().splitlines()[:8]:
        print(line[:220])
# the timed region of that run = the LAST `steps` dispatches of the headline kernel (the pre-roll in front of them runs
# slower: clock ramp); their mean is what bench.py's kernel_ms must agree with
tr = glob.glob(os.path.join(out, "stats", "**", "*_kernel_trace.csv"), recursive=True)
if tr:
    rows = [r for r in csv.DictReader(open(tr[0])) if HEAD in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
    steps = 200
    if len(d) >= steps:
        last = d[-steps:]
        info = {"kernel": HEAD, "dispatches": len(d), "mean_ms_all": sum(d) / len(d), "timed_region": "last %d dispatches" % steps,
                "mean_ms_timed_region": sum(last) / steps, "min_ms": min(last), "max_ms": max(last),
                "mean_ms_first_50": sum(d[:50]) / 50}
        json.dump(info, open(os.path.join(PROF, tag + "_headline_trace.json"), "w"), indent=1)
        print(json.dumps(info))
sf = glob.glob(os.path.join(out, "stats_full", "**", "*_kernel_stats.csv"), recursive=True)
if sf:
    shutil.copy(sf[0], os.path.join(PROF, tag + "_kernel_stats_full.csv"))
bj = os.path.join(out, "bench_stats.json")
if os.path.exists(bj):
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(PROF, tag + "_bench_under_rocprof.json"), "w").write(lines[-1] + "\n")
fetch, write = pick(counters(os.path.join(out, "fetch")), HEAD), pick(counters(os.path.join(out, "write")), HEAD)
if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
    rd, wr = 2.0 * fetch["FETCH_SIZE"]["mean"] * 1024, write["WRITE_SIZE"]["mean"] * 1024
    t = {"kernel": fetch["_kernel"], "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
         "source": tag + "_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, KiB -> bytes)"}
    json.dump({"note": "separate --pmc passes; KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE exact",
               "FETCH_SIZE": fetch, "WRITE_SIZE": write, "per_launch": t}, open(os.path.join(PROF, tag + "_pmc.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(PROF, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(t))
sq = {"note": "separate --pmc passes (8 SQ slots each); per-launch means; SQ_* summed over the chip; *_CYCLES / WAIT / ACTIVE in quad-cycles"}
h = {}
for d in ("sq1", "sq2"):
    h.update(pick(counters(os.path.join(out, d)), HEAD))
s, s_plain, s_fused = {}, {}, {}
for d in ("sw1", "sw2"):
    s.update(pick(counters(os.path.join(out, d)), SWEEP))
    s_plain.update(pick(counters(os.path.join(out, d)), SWEEP_PLAIN))
    s_fused.update(pick(counters(os.path.join(out, d)), SWEEP_FUSED))
if h:
    if "SQ_ACTIVE_INST_ANY" in h and "SQ_WAVE_CYCLES" in h:
        h["active_fraction = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"] = h["SQ_ACTIVE_INST_ANY"]["mean"] / h["SQ_WAVE_CYCLES"]["mean"]
    # busy fractions of the launch (bench.py attaches them to roofline as builder-side figures): GRBM_GUI_ACTIVE is summed over
    # the 8 XCDs -> cycles of the launch; SQ_ACTIVE_INST_VALU counts one quad-cycle per VALU instruction, i.e. 4 cycles each
    # (packed fp32: measured 4, tools/ubench/pk_rate.hip; unpacked: 2 -- a slight over-statement) on 1024 SIMDs;
    # SQ_LDS_IDX_ACTIVE counts LDS-array cycles on 256 CUs
    need = ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
    if all(k in h for k in need):
        cyc = h["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        lat = {"kernel": h["_kernel"], "launch_cycles": cyc,
               "valu_busy": 4.0 * h["SQ_ACTIVE_INST_VALU"]["mean"] / (1024.0 * cyc),
               "lds_busy": h["SQ_LDS_IDX_ACTIVE"]["mean"] / (256.0 * cyc),
               "active_fraction": h["SQ_ACTIVE_INST_ANY"]["mean"] / h["SQ_WAVE_CYCLES"]["mean"],
               "source": tag + "_pmc_sq.json (rocprofv3 --pmc passes of python3 bench.py --no-extra ..., builder-side run; counters "
                               "perturb the clock: compare ratios, not times)"}
        json.dump(lat, open(os.path.join(PROF, "pmc_latest.json"), "w"), indent=1)
        print(json.dumps(lat))
    sq["headline"] = h
if s:
    sq["sweep"] = s
if s_plain:
    sq["sweep_unpruned"] = s_plain
if s_fused:
    sq["sweep_fused_hilbert_pass"] = s_fused
json.dump(sq, open(os.path.join(PROF, tag + "_pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else v.get("mean")) for k, v in h.items()}, indent=0)[:1500])
print(json.dumps({k: (v if not isinstance(v, dict) else v.get("mean")) for k, v in s.items()}, indent=0)[:1500])
