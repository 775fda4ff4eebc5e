"""fp16 sample path, matrix-core passes (prot_f16mfma.hip) against the packed-half butterflies (olapoly_kernel<T16>, test option
PROT_TEST_F16_NO_MFMA), interleaved in ONE process on one box: ms per 2^27 frames, median of 5 x 40 launches per round."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5).half(); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
cases = {"lv2_48k": (pkg.KIND_LV2, 48000), "lv2_96k": (pkg.KIND_LV2, 96000), "cli_2048": (pkg.KIND_CLI, 2048), "cli_1024": (pkg.KIND_CLI, 1024)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["lv2_48k", "cli_2048"]
for name in names:
    kind, param = cases[name]
    e = pkg.Engine(kind, param, 1)
    call = lambda: e.apply_dev_f16(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, st)
    res = {0: [], 1: []}
    for rnd in range(3):
        for off in (0, 1):
            pkg.set_test_option(pkg.TEST_F16_NO_MFMA, off)
            for _ in range(150):
                call()
            torch.cuda.synchronize()
            tt = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    call()
                e1.record(); torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1) / 40)
            res[off].append(sorted(tt)[2])
    pkg.set_test_option(pkg.TEST_F16_NO_MFMA, 0)
    f = lambda v: " | ".join("%.4f" % t for t in v)
    print("%-9s matrix core %s   (%.3f of 4 B/frame at 8 TB/s)    vector pipe %s   (%.3f)" % (name, f(res[0]), 4.0 * n / (min(res[0]) * 1e-3) / 8e12, f(res[1]), 4.0 * n / (min(res[1]) * 1e-3) / 8e12))
    e.close()
