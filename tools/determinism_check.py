import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
for kind, param in ((pkg.KIND_LV2, 48000), (pkg.KIND_CLI, 2048)):
    n = 1 << 27
    e = pkg.Engine(kind, param, 1)
    x = torch.rand(n, device=dev) - 0.5
    rot = np.array([0.6, -0.8], np.float32)
    torch.cuda.synchronize()  # x comes from torch's stream; the engine launches on its own
    ys = []
    for rep in range(3):
        y = torch.empty_like(x)
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS)
        e.sync(); ys.append(y)
    for i in (1, 2):
        d = (ys[i] != ys[0]).nonzero().flatten()
        print(kind, "rep", i, "mismatches", d.numel(), d[:10].tolist(), (ys[i]-ys[0]).abs().max().item())
    y2 = torch.empty_like(x); x2 = 2 * x; torch.cuda.synchronize()
    e.apply_dev(x2.data_ptr(), n, n, y2.data_ptr(), n, n, rot, pkg.F_STATELESS); e.sync()
    d = (y2 != 2 * ys[0]).nonzero().flatten()
    print(kind, "linearity mismatches", d.numel(), d[:10].tolist(), (y2 - 2*ys[0]).abs().max().item())
    if d.numel():
        i = d[0].item(); print(" at", i, y2[i].item(), 2*ys[0][i].item(), x[i-2000:i+1].abs().min().item())
    e.close()
