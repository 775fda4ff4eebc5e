"""olasave8k_kernel against olasave_kernel (PROT_CFG_BLOCK4096) on one long stream: max difference, bit-identity of the generated
loop and the staged path, and ms per 2^27 frames of both, interleaved.  python tools/sv8_check.py [rate] [log2 frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 27
dev = torch.device("cuda:0")
n = 1 << lg
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(n, generator=g, device=dev) - 0.5
rot = np.array([0.28, -0.96], np.float32)
e8 = pkg.Engine(pkg.KIND_LV2, rate, 1)
e4 = pkg.Engine(pkg.KIND_LV2, rate, 1, flags=pkg.CFG_BLOCK4096)
print("forms", e8.info.form, e4.info.form)
y8, y4, y8c = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
e8.apply_dev(x.data_ptr(), n, n, y8.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
e4.apply_dev(x.data_ptr(), n, n, y4.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
d = (y8 - y4).abs()
print("8k vs 4k: max diff %.3e at %d; |y| max %.3f" % (float(d.max()), int(d.argmax()), float(y8.abs().max())))
bad = torch.nonzero(d > 2e-6).flatten()
if bad.numel():
    print("bad frames: %d, first %s last %s" % (bad.numel(), bad[:8].tolist(), bad[-8:].tolist()))
pkg.set_test_option(pkg.TEST_EDGE_PATH_ONLY, 1)
e8c = pkg.Engine(pkg.KIND_LV2, rate, 1)
pkg.set_test_option(pkg.TEST_EDGE_PATH_ONLY, 0)
m = min(n, 1 << 25)
e8c.apply_dev(x.data_ptr(), m, m, y8c.data_ptr(), m, m, rot, pkg.F_STATELESS, st)
y8b = torch.zeros_like(x)
e8.apply_dev(x.data_ptr(), m, m, y8b.data_ptr(), m, m, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
print("generated loop == staged path bit for bit (first 2^25):", bool(torch.equal(y8b[:m], y8c[:m])), " same as the full-length launch:", bool(torch.equal(y8b[:m - 65536], y8[:m - 65536])))
def timed(e, k=40):
    for _ in range(150):
        e.apply_dev(x.data_ptr(), n, n, y8.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(ts)
    for _ in range(k):
        e.apply_dev(x.data_ptr(), n, n, y8.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    b.record(ts); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for r in range(3):
    t8 = timed(e8); t4 = timed(e4)
    print("round %d: 8192-point %.4f ms (%.1f %% of 8 TB/s)   4096-point %.4f ms (%.1f %%)" % (r, t8, 8.0 * n / t8 / 8e9 * 100, t4, 8.0 * n / t4 / 8e9 * 100))
