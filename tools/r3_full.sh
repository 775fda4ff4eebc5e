#!/bin/bash
# full GPU suite + bench line
mkdir -p gpurun_out/r3_full
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r3_full/tests.txt
python bench.py > gpurun_out/r3_full/bench.json 2> gpurun_out/r3_full/bench.err
tail -5 gpurun_out/r3_full/tests.txt; tail -c 600 gpurun_out/r3_full/bench.err
