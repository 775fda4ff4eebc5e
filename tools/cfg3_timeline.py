#!/usr/bin/env python3
"""Where a cfg3 launch goes (LV2 stereo, 96 kHz filter, 256 x 1024-frame buffers per stateful launch: 128 workgroups of one 4096-point
block each): the launch period by events, and per workgroup -- from a -DPROT_SV_CLOCK variant (python3 tools/ab_build.py
sv:clk:-DPROT_SV_CLOCK,-DPROT_SV_CLOCK_NOSUM) -- kernel entry, generated-loop entry, loop exit, kernel exit in 100 MHz ticks.
    PROT_LIB_PATH=.../variants/libprot_hip_clk.so python3 tools/cfg3_timeline.py [n_buffers]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
n_chn, n = 2, (int(sys.argv[1]) if len(sys.argv) > 1 else 256) * 1024
eng = pkg.Engine(pkg.KIND_LV2, 96000, n_chn)
x = torch.rand((n_chn, n), device=dev) - 0.5
y = torch.empty_like(x)
rot = np.array([0.0, -1.0, 0.6, 0.8], np.float32)
step = lambda: eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, 0, st)
for _ in range(300):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L = 1000
e0.record(ts)
for _ in range(L):
    step()
e1.record(ts)
torch.cuda.synchronize()
print("launch period %.2f us by events" % (e0.elapsed_time(e1) / L * 1e3))
if hasattr(lib, "prot_debug_sv_wg"):
    nw = 512
    a = (ctypes.c_ulonglong * (4 * nw))()
    lib.prot_debug_sv_wg(a, nw)
    w = np.array(list(a), dtype=np.float64).reshape(nw, 4) * 0.01
    w = w[w[:, 0] > 0]
    if len(w):
        t0 = w[:, 0].min()
        print("workgroups stamped %d (x-index only: both channels write the same slot); entry %.2f..%.2f us after the first" % (len(w), 0.0, (w[:, 0] - t0).max()))
        lp = w[w[:, 1] > 0]
        if len(lp):
            print("  loop workgroups %d: entry->loop %.2f (median) loop %.2f loop end->kernel exit %.2f; whole %.2f us" % (
                len(lp), np.median(lp[:, 1] - lp[:, 0]), np.median(lp[:, 2] - lp[:, 1]), np.median(lp[:, 3] - lp[:, 2]), np.median(lp[:, 3] - lp[:, 0])))
        sg = w[w[:, 1] == 0]
        if len(sg):
            print("  staged workgroups %d: whole %.2f us (median), max %.2f" % (len(sg), np.median(sg[:, 3] - sg[:, 0]), (sg[:, 3] - sg[:, 0]).max()))
        print("  first entry -> last exit %.2f us" % (w[:, 3].max() - t0))
if hasattr(lib, "prot_debug_s5_wg"):   # the 512-thread block (a -DPROT_S5_CLOCK variant: python3 tools/ab_build.py s5:clk:-DPROT_S5_CLOCK)
    nw = 256
    a = (ctypes.c_ulonglong * (8 * nw))()
    lib.prot_debug_s5_wg(a, nw)
    w = np.array(list(a), dtype=np.float64).reshape(nw, 8)[:, [0, 1, 3, 4]] * 0.01   # entry, rows + constants arrived, transform done, exit
    w = w[w[:, 0] > 0]
    if len(w):
        d = np.diff(w, axis=1)
        print("512-thread block, %d workgroups: entry->rows+constants %.2f, transform %.2f, stores+exit %.2f; whole %.2f us (median), max %.2f; first entry -> last exit %.2f" % (
            len(w), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(w[:, 3] - w[:, 0]), (w[:, 3] - w[:, 0]).max(), w[:, 3].max() - w[:, 0].min()))
        slow = np.argsort(w[:, 3] - w[:, 0])[-3:]
        print("  slowest:", [(int(i), [round(float(x), 2) for x in d[i]]) for i in slow])
eng.close()
