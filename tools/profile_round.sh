#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>
# rocprofv3 kernel stats of the default bench command, then separate --pmc passes (never combined with traces) for HBM
# traffic and the SQ counters of the headline kernel and of the sweep; raw output under gpurun_out/prof_<tag>/, the
# summaries that are meant to be judged are written to profiles/ by tools/summarize_round.py (run it there, copy back).
# Every rocprofv3 run sits under its own `timeout`: a counter set the hardware cannot schedule aborts and then hangs in the
# profiler's finalisation (round 5: 15 GPU-minutes lost to one such run).
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-extra --no-cpu-baseline"
S="--steps 4 --warmup 1 --preroll 2 --no-sweep"
# headline only (the same kernel instantiation also runs in the small LV2 run() launches of the full bench, which would pull its average down)
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -- $B --no-sweep --steps 200 --warmup 3 --preroll 200 > $OUT/bench_stats.json 2> $OUT/stats.err
# the full default bench (sweep and side measurements included)
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_full -- python3 bench.py --no-cpu-baseline > $OUT/bench_stats_full.json 2> $OUT/stats_full.err
timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- $B $S > /dev/null 2> $OUT/fetch.err
timeout 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -- $B $S > /dev/null 2> $OUT/write.err
timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/sq1 -- $B $S > /dev/null 2> $OUT/sq1.err
timeout 900 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/sq2 -- $B $S > /dev/null 2> $OUT/sq2.err
# the sweep (one pass of each material is enough: --sweep-seconds 600 keeps the counter runs short)
W="python3 bench.py --no-extra --no-cpu-baseline --steps 2 --warmup 1 --preroll 1 --sweep-seconds 600"
timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/sw1 -- $W > /dev/null 2> $OUT/sw1.err
timeout 900 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/sw2 -- $W > /dev/null 2> $OUT/sw2.err
# the sweep's HBM bytes by counter (harmonic and noise at blksiz 2048, harmonic at 16384): tools/sweep_traffic.py
for K in harmonic_2048 noise_2048 harmonic_16384 harmonic_8192; do
  timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/swt/fetch_$K -- python3 tools/sweep_traffic.py ${K%_*} ${K#*_} > $OUT/swt_fetch_$K.txt 2> $OUT/swt_fetch_$K.err
  timeout 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/swt/write_$K -- python3 tools/sweep_traffic.py ${K%_*} ${K#*_} > $OUT/swt_write_$K.txt 2> $OUT/swt_write_$K.err
done
python3 tools/sweep_traffic.py --summarize $OUT/swt $TAG
# the fp16 sample path (round 6): its launch alone with the radix-16 passes on the matrix core and on the vector pipe (tools/f16_prof.py)
for V in mfma no_mfma; do
  A=""; [ $V = no_mfma ] && A=no_mfma
  timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/f16/$V/stats -- python3 tools/f16_prof.py $A > /dev/null 2> $OUT/f16_${V}_stats.err
  timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA -d $OUT/f16/$V/a -- python3 tools/f16_prof.py $A > /dev/null 2> $OUT/f16_${V}_a.err
  timeout 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU -d $OUT/f16/$V/b -- python3 tools/f16_prof.py $A > /dev/null 2> $OUT/f16_${V}_b.err
done
python3 - $TAG $OUT <<'PY'
import collections, csv, glob, json, os, sys
tag, out = sys.argv[1], sys.argv[2]
res = {"note": "tools/f16_prof.py under rocprofv3 (30 launches of 2^27 frames, LV2 48 kHz fp16): kernel trace and two --pmc passes per kernel; per-launch means; SQ *_CYCLES / ACTIVE / WAIT in quad-cycles"}
for V in ("mfma", "no_mfma"):
    r = {}
    for part in "ab":
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, "f16", V, part, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "olapoly" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    r["kernel"] = row["Kernel_Name"][:120]
        r.update({k: sum(v) / len(v) for k, v in acc.items()})
    for f in glob.glob(os.path.join(out, "f16", V, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "olapoly" in row["Name"]:
                r["kernel_stats"] = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
    res[V] = r
json.dump(res, open(os.path.join("profiles", tag + "_f16_pmc.json"), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
rm -rf $OUT/f16
rocprofv3 -L 2>/dev/null | grep -i "mfma\|SQ_INSTS_VALU\b" | head -20 > $OUT/counters_available.txt
python3 tools/summarize_round.py $TAG $OUT
ls $OUT
