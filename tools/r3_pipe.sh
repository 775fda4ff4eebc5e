#!/bin/bash
mkdir -p gpurun_out/r3_pipe
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pipelined or cfg3 or stateful or chunk" 2>&1 | tail -15 > gpurun_out/r3_pipe/tests.txt
tail -6 gpurun_out/r3_pipe/tests.txt
timeout 300 python - > gpurun_out/r3_pipe/cfg3.txt 2>&1 <<'P'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n_chn, n = 2, 256 * 1024
x = torch.rand((n_chn, n), device=dev) - 0.5
ys = [torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)]
rot = np.array([0.0, -1.0, 0.6, 0.8], np.float32)
for flags in (0, pkg.CFG_PIPELINE, 0, pkg.CFG_PIPELINE):
    eng = pkg.Engine(pkg.KIND_LV2, 96000, n_chn, flags=flags)
    for k in range(200):
        eng.apply_dev(x.data_ptr(), n, n, ys[k % 3].data_ptr(), n, n, rot, 0, 0)
    eng.sync()
    t0 = time.perf_counter()
    N = 2000
    for k in range(N):
        eng.apply_dev(x.data_ptr(), n, n, ys[k % 3].data_ptr(), n, n, rot, 0, 0)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print("flags %2d: %.2f us per launch (wall incl. sync), host issue %.2f us per call" % (flags, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6), flush=True)
    eng.close()
P
cat gpurun_out/r3_pipe/cfg3.txt
