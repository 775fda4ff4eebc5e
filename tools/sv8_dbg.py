import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
rate = int(sys.argv[1])
dev = torch.device("cuda:0")
n = 1 << 24
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(n, generator=g, device=dev) - 0.5
rot = np.array([0.28, -0.96], np.float32)
e4 = pkg.Engine(pkg.KIND_LV2, rate, 1, flags=pkg.CFG_BLOCK4096)
y4 = torch.zeros_like(x)
e4.apply_dev(x.data_ptr(), n, n, y4.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
for edge in (1, 0):
    pkg.set_test_option(pkg.TEST_EDGE_PATH_ONLY, edge)
    e8 = pkg.Engine(pkg.KIND_LV2, rate, 1)
    pkg.set_test_option(pkg.TEST_EDGE_PATH_ONLY, 0)
    y8 = torch.zeros_like(x)
    e8.apply_dev(x.data_ptr(), n, n, y8.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    torch.cuda.synchronize()
    d = (y8 - y4).abs()
    d = torch.nan_to_num(d, nan=9.0)
    hs = int(e8.info.firlen); rows = 16 - hs // 1024; blk = 1024 * rows
    print("edge-only" if edge else "generated loop", "form", e8.info.form, "rows", rows, "max diff %.3e" % float(d.max()), "bad %d" % int((d > 2e-6).sum()))
    # per row of the first blocks
    for b in (0, 1, 2):
        r = [float(d[b * blk + 1024 * k: b * blk + 1024 * (k + 1)].max()) for k in range(rows)]
        print("  block", b, " ".join("%.0e" % v for v in r))
    # within the worst row: which threads
    k = 0
    seg = d[0:1024].cpu().numpy()
    print("  first row bad thread ids (pair index):", np.nonzero(seg.reshape(512, 2).max(1) > 2e-6)[0][:20], "count", int((seg.reshape(512,2).max(1) > 2e-6).sum()))
