#!/bin/bash
# memory-side counter passes of the headline launch (on the GPU box): bash tools/pmc_mem.sh <outdir>
# address translation (UTCL1 / UTCL2), the vector cache's pending stalls, the L2's stalls towards the fabric
OUT=${1:-gpurun_out/pmcm}; shift
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-extra --no-cpu-baseline --no-sweep --steps 4 --warmup 1 --preroll 2"
rocprofv3 -L 2>/dev/null | grep -o "\(TCP\|TCC\|TCA\|TA\|TD\|SQC\|GL2C\|UTCL2\|ATC\|MC\|VML2\|L2TLB\)_[A-Za-z_0-9]*" | sort -u > $OUT/mem_counters.txt
pass() {
  local name=$1; shift
  local have=""
  for c in "$@"; do if grep -qx "$c" $OUT/mem_counters.txt; then have="$have $c"; fi; done
  if [ -z "$have" ]; then echo "pass $name: none of [$*] exists" >> $OUT/log.txt; return; fi
  echo "pass $name:$have" >> $OUT/log.txt
  rocprofv3 --output-format csv --pmc $have -d $OUT/$name -- $B > /dev/null 2> $OUT/$name.err
}
pass a TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
pass b TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass c TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum
pass d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass e TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TCP_TCC_NC_READ_REQ_sum
pass f TCC_EA_RDREQ_LEVEL_sum TCC_EA_WRREQ_LEVEL_sum TCC_EA_ATOMIC_sum TCC_BUSY_sum
pass g TCC_EA_WR_UNCACHED_32B_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum TCC_EA_RDREQ_DRAM_sum
python3 - $OUT <<'PY'
import sys, glob, csv, collections, os, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    if "olasave" not in k:
        continue
    print("KERNEL", k[:110])
    res[k] = {}
    for c in sorted(cs):
        v = cs[c]
        res[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
        print("  %-34s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
json.dump(res, open(os.path.join(out, "pmc_mem.json"), "w"), indent=1)
PY
for n in a b c d e f g; do rm -rf $OUT/$n; done
