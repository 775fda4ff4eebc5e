"""Diagnostic: per-XCD timeline of the headline polyphase kernel, launch by launch (needs the timeline build:
`python tools/ab_build.py stamps2:-DPROT_STAMPS=2`, PROT_LIB_PATH=.../libprot_hip_stamps2.so).
Prints, for each of N launches in ONE process, the median / max end time (us after the first workgroup's entry) of the
workgroups dealt to each XCD (blockIdx % 8), by dispatch rank -- to tell a per-XCD property (same pattern every launch and
every process) from an address-dependent one (same within a process, different across processes) from noise."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
lib.prot_debug_wg_times.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
realloc = len(sys.argv) > 2 and sys.argv[2] == "realloc"
dev = torch.device("cuda:0")
n = 1 << 27
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(pkg.KIND_LV2, 48000, 1, flags=0)
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
for _ in range(10):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
nwg = 768
keep = []
for l in range(n_launch):
    if realloc:  # other physical pages behind the same shapes
        keep.append((x, y))
        x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    ev1.record(); torch.cuda.synchronize()
    wt = (C.c_ulonglong * (4 * nwg))()
    lib.prot_debug_wg_times(wt, nwg)
    w = np.array(list(wt), dtype=np.int64).reshape(nwg, 4)
    live = w[:, 2] > 0
    t0 = w[:, 0][live].min()
    end = (w[:, 2] - t0) / 100.0
    bid = np.arange(nwg); rk = np.minimum(bid // 256, 2)
    xcc = (w[:, 3] >> 32) & 0xf
    row = []
    for xx in range(8):
        m = live & ((bid % 8) == xx)
        row.append("%d[x%d]:%s/%3.0f" % (xx, int(np.median(xcc[m])), "/".join("%3.0f" % np.median(end[m & (rk == k)]) for k in range(3) if (m & (rk == k)).any()), end[m].max()))
    print("launch %d (x at %#x): %.1f us by events, last end %.0f us | %s" % (l, x.data_ptr(), ev0.elapsed_time(ev1) * 1e3, end[live].max(), "  ".join(row)))
