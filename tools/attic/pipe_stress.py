"""Pipelined (PROT_CFG_PIPELINE) against sequential stateful launches: which launches differ, where, how often.
python3 tools/pipe_stress.py [launches] [repeats] [per]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 48
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = int(sys.argv[3]) if len(sys.argv) > 3 else 256 * 1024
n_chn = 2
n = n_launch * per
g = torch.Generator(device=dev).manual_seed(5)
x = torch.rand((n_chn, n), generator=g, device=dev) - 0.5
rot = np.array([0.6, 0.8, -0.28, 0.96], np.float32)
def run(flags):
    e = pkg.Engine(pkg.KIND_LV2, 96000, n_chn, flags=flags)
    y = torch.zeros_like(x)
    for k in range(n_launch):
        e.apply_dev(x[:, k * per:].data_ptr(), n, per, y[:, k * per:].data_ptr(), n, per, rot, 0)
    e.sync()
    torch.cuda.synchronize()
    e.close()
    return y
ref = run(0)
for r in range(reps):
    y = run(pkg.CFG_PIPELINE)
    d = (y != ref)
    bad = []
    for k in range(n_launch):
        c = d[:, k * per:(k + 1) * per]
        if bool(c.any()):
            idx = torch.nonzero(c.any(0)).flatten()
            bad.append((k, int(c.sum()), int(idx[0]), int(idx[-1])))
    print("repeat %d: %d launches differ: %s" % (r, len(bad), bad[:12]), flush=True)
