import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
HS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kind, param = {10: (pkg.KIND_LV2, 48000), 8: (pkg.KIND_LV2, 96000), 12: (pkg.KIND_CLI, 2048), 14: (pkg.KIND_CLI, 1024)}[HS]
BLK = 512 * HS
n = BLK * 64
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(n, device=dev, generator=g) - 0.5).half()
rot = np.array([0.0, -1.0], np.float32)
e = pkg.Engine(kind, param, 1)
ys = []
for off in (1, 0):
    pkg.set_test_option(pkg.TEST_F16_NO_MFMA, off)
    y = torch.zeros_like(x)
    e.apply_dev_f16(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, 0)
    e.sync()
    ys.append(y.float().cpu().numpy())
pkg.set_test_option(pkg.TEST_F16_NO_MFMA, 0)
d = np.abs(ys[0] - ys[1])
print("max diff", d.max(), "rms", np.sqrt((d ** 2).mean()), "run_frames?", e.info.run_align)
pos = np.arange(n)
blk = pos // BLK
row = (pos % BLK) // 512
col = (pos % 512) // 2
print("by block (first 12):", [float("%.3g" % d[blk == b].max()) for b in range(12)])
print("by row:", [float("%.3g" % d[row == r].max()) for r in range(HS)])
print("by column group of 32:", [float("%.3g" % d[(col // 32) == c].max()) for c in range(8)])
print("by parity:", float(d[0::2].max()), float(d[1::2].max()))
