"""the fp16 sample path's launch alone (LV2 48 kHz, 2^27 frames, 30 launches): something for rocprofv3 to look at
(python3 tools/f16_prof.py [no_mfma]: the vector-pipe kernel instead)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5).half(); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "no_mfma":
    pkg.set_test_option(pkg.TEST_F16_NO_MFMA, 1)
e = pkg.Engine(pkg.KIND_LV2, 48000, 1)
for _ in range(30):
    e.apply_dev_f16(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, 0)
e.sync()
