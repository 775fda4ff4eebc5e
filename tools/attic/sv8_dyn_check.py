"""Dynamic blocks (olasave8k_dyn_kernel) against the static run plan (PROT_CFG_STATIC_RUNS): outputs bit for bit on several
shapes (mono / stereo, lengths that are not whole blocks, stateless and stateful calls), then the two timed interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
torch.manual_seed(3)
ok = True
for kind, param, nc, n, fl in ((pkg.KIND_LV2, 48000, 1, 1 << 27, pkg.F_STATELESS), (pkg.KIND_LV2, 48000, 2, (1 << 25) + 12346, pkg.F_STATELESS),
                              (pkg.KIND_LV2, 96000, 1, (1 << 26) + 77, pkg.F_STATELESS), (pkg.KIND_LV2, 48000, 1, (1 << 26) + 4098, 0),
                              (pkg.KIND_LV2, 192000, 3, (1 << 24) + 2, pkg.F_STATELESS), (pkg.KIND_CLI, 8192, 2, 1 << 25, pkg.F_STATELESS)):
    x = (torch.rand(nc * n, device=dev) - 0.5)
    ya, yb = torch.zeros_like(x), torch.zeros_like(x)
    rot = np.array([0.6, -0.8] * nc, np.float32)
    ea, eb = pkg.Engine(kind, param, nc, flags=0), pkg.Engine(kind, param, nc, flags=pkg.CFG_STATIC_RUNS)
    for rep in range(2):   # (twice: the second launch finds the counters as the first one left them)
        ea.apply_dev(x.data_ptr(), n, n, ya.data_ptr(), n, n, rot, fl, st)
        eb.apply_dev(x.data_ptr(), n, n, yb.data_ptr(), n, n, rot, fl, st)
        torch.cuda.synchronize()
        same = torch.equal(ya, yb)
        nz = int((ya != 0).sum().item())
        print("kind %d param %d chn %d n %d flags %d rep %d: %s (non-zero outputs %d of %d, max |a - b| %.3g)" % (
            kind, param, nc, n, fl, rep, "identical" if same else "DIFFERENT", nz, nc * n, float((ya - yb).abs().max().item())))
        ok = ok and same
    ea.close(); eb.close()
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
eng = {"dynamic": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0), "static": pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=pkg.CFG_STATIC_RUNS)}
for name, e in eng.items():
    for _ in range(200):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
for rnd in range(3):
    for name, e in eng.items():
        for _ in range(60):
            e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(40):
            e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 40
        print("round %d %-8s %.4f ms per 2^27 frames = %.3f of 8 TB/s" % (rnd, name, ms, 8.0 * n / (ms * 1e-3) / 8e12))
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
