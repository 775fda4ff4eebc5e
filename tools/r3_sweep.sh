#!/bin/bash
# sweep kernel times (rocprofv3 kernel trace of the probe on five signals)
mkdir -p gpurun_out/r3_sweep
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_sw -- python3 $GRAFT_REPO_ROOT/tools/sweep_prune_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r3_sweep/probe.txt 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_sw -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_sweep/kernel_stats.csv \;
find /tmp/prof_sw -name "*kernel_trace.csv" -exec cp {} gpurun_out/r3_sweep/kernel_trace.csv \;
grep -a "pruned" gpurun_out/r3_sweep/probe.txt | tail -12
