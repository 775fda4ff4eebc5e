#!/bin/bash
# quick SQ counter passes of the headline launch (on the GPU box): bash tools/pmc_quick.sh <outdir> [extra env assignments...]
OUT=${1:-gpurun_out/pmcq}; shift
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
B="python3 bench.py --no-extra --no-cpu-baseline --no-sweep --steps 4 --warmup 1 --preroll 2"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/a -- $B > /dev/null 2> $OUT/a.err
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/b -- $B > /dev/null 2> $OUT/b.err
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM -d $OUT/c -- $B > /dev/null 2> $OUT/c.err
python3 - $OUT <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "ola" not in k:
        continue
    print("KERNEL", k[:110])
    for c in sorted(cs):
        v = cs[c]
        print("  %-26s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
rm -rf $OUT/a $OUT/b $OUT/c
