#!/bin/bash
# sweep kernel check: parity tests, then kernel times (rocprofv3 kernel trace of the probe on five signals)
mkdir -p gpurun_out/r3_sweep
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_sweep_semantics.py -q -m gpu -k "sweep or analyze or cli or shard or tie" -s 2>&1 | tail -60 > gpurun_out/r3_sweep/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_sw -- python3 $GRAFT_REPO_ROOT/tools/sweep_prune_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r3_sweep/probe.txt 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_sw -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_sweep/kernel_stats.csv \;
find /tmp/prof_sw -name "*kernel_trace.csv" -exec cp {} gpurun_out/r3_sweep/kernel_trace.csv \;
grep -a "evaluated\|passed\|failed" gpurun_out/r3_sweep/tests.txt | tail -20;
grep -a "pruned\|untested" gpurun_out/r3_sweep/probe.txt | tail -12
