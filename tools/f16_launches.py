"""20 launches of the fp16 sample path at the headline shape (for rocprofv3 --pmc passes: `rocprofv3 --pmc ... -- python3 tools/f16_launches.py`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5).half(); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    e.apply_dev_f16(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, st)
torch.cuda.synchronize()
print("done")
