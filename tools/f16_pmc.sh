#!/bin/bash
# SQ counters of the fp16 headline-shape kernel (four workgroups per CU), two separate --pmc passes; prints per-launch means
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/f16p; mkdir -p $O
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq1 -- python3 tools/f16_launches.py 8 > /dev/null 2> $O/err1.txt
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sq2 -- python3 tools/f16_launches.py 8 > /dev/null 2> $O/err2.txt
python3 - <<PY > $O/f16_pmc.txt
import csv, glob, collections
for d in ("sq1","sq2"):
    agg=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/f16p/%s/**/*_counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "olapoly_kernel<prot::T16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
cat $O/f16_pmc.txt; rm -rf $O/sq1 $O/sq2
