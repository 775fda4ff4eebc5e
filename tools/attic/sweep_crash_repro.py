"""repro of an abort inside analyze_file seen in test_resident_file_passes_equal_streamed_ones[2048-2-8388608] (round 6)"""
import os, sys, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from __graft_entry__ import load_package
from conftest import harmonic_stack, xorshift_uniform
pkg = load_package()
n, n_chn, blksiz = int(sys.argv[1]), 2, 2048
x = np.ascontiguousarray(np.stack([0.5 * harmonic_stack(n, 48000) * (0.5 + 0.2 * c) + 0.03 * xorshift_uniform(n, seed=60 + c) for c in range(n_chn)], 1), np.float32)
y = np.ascontiguousarray(x[::-1] * 0.9)
for resident in (False, True):
    g = pkg.PhaseRotate(blksiz, n_chn)
    if resident:
        g.file_resident(True)
    for k, (data, (a0, a1, st)) in enumerate(((x, (0, 360, 24)), (x, (100, 140, 1)), (y, (0, 360, 8)), (x, (0, 360, 3)))):
        g.reset()
        print("analyze", resident, k, flush=True)
        g.analyze_file(data, a0, a1, st)
    g.reset()
    print("apply", resident, flush=True)
    g.apply_file(x, [37 + 5 * c for c in range(n_chn)])
    if resident:
        g.file_resident(False)
    g.close()
print("done")
