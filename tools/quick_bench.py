import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
for kind, param, fl, name in ((pkg.KIND_LV2, 48000, 1, "lv2-48k hop1024"), (pkg.KIND_LV2, 48000, 0, "lv2-48k p2"), (pkg.KIND_LV2, 96000, 0, "lv2-96k p2"), (pkg.KIND_CLI, 2048, 0, "cli-2048 hop2048")):
    n = 1 << 27
    e = pkg.Engine(kind, param, 1, flags=fl)
    x = (torch.rand(n, device=dev) - 0.5)
    y = torch.empty_like(x)
    rot = np.array([0.0, -1.0], np.float32)
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
    for _ in range(2):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    K = 5
    for _ in range(K):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / K
    print("%s: %.3f ms per 2^27 frames -> %.2f Gframes/s, %.2f TB/s algorithmic (%.1f%% of 8 TB/s)" % (
        name, ms, n / ms / 1e6, n * 8 / ms / 1e9, n * 8 / ms / 1e9 / 8 * 100), flush=True)
    e.close()
