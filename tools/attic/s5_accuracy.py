"""max / rms error of the short-chain block against the CPU oracle's run() and against the 256-thread block (round 6: how many of the
block's constants are loaded and how many are products)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import orc
pkg = load_package()
dev = torch.device("cuda:0")
res = []
for rate, lat in ((48000, 1792), (96000, 2560)):
    n = 200000
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand(n, generator=g, device=dev) - 0.5
    rot = np.array([0.28, -0.96], np.float32)
    ys = []
    for off in (0, 1):
        pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, off)
        e = pkg.Engine(pkg.KIND_LV2, rate, 1)
        y = torch.zeros_like(x)
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS)
        e.sync(); e.close()
        ys.append(y.cpu().numpy().astype(np.float64))
    pkg.set_test_option(pkg.TEST_NO_SHORT_CHAIN, 0)
    o = orc.Lv2Engine(rate, 1); o.activate()
    ang = np.degrees(np.arctan2(-rot[1], rot[0]))
    yo = o.run_stream(x.cpu().numpy()[None, :], float(ang), 1024)[0].astype(np.float64)
    skip = 2 * lat + 4096
    d5, d2 = ys[0][skip:] - yo[skip:], ys[1][skip:] - yo[skip:]
    res.append("%d kHz: 512-thread vs oracle max %.2e rms %.2e | 256-thread vs oracle max %.2e rms %.2e | 512 vs 256 max %.2e" % (
        rate // 1000, np.abs(d5).max(), np.sqrt((d5 ** 2).mean()), np.abs(d2).max(), np.sqrt((d2 ** 2).mean()), np.abs(ys[0] - ys[1]).max()))
print(os.environ.get("PROT_LIB_PATH", "shipped").split("_")[-1], " || ".join(res))
