"""Does what a process did before change the time of a headline launch or of a cfg4 sweep pass?  Both are timed twice in a fresh
process, then after ONE of: file (bench_file_host), pinned / pageable (H2D copies), streams (three torch streams at once), events
(two streams and the caller's chained by event waits), apply_file / analyze_file / engine8192 -- and timed twice again.
Round 4's finding: only `events` (and the bench sections that wait across streams) moved anything -- the sweep pass by +8 %, the
headline not at all -- and the cause was prot_engine_reset's host wait per pass, not a kernel (the reset is stream-ordered now).
    python3 tools/process_state_probe.py events"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); st = ts.cuda_stream
n = 1 << 27
x = torch.rand(n, device=dev) - 0.5
y = torch.empty_like(x)
th = -2.0 * np.pi * 90.0 / 360.0
rot = np.array([np.cos(th), np.sin(th)], np.float32)
P, n_chn = 2048, 2
total = 3600 * 96000 // P * P
xs = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, "noise")
peaks = torch.zeros((n_chn, 360), device=dev)

def head():
    eng = pkg.Engine(pkg.KIND_LV2, 48000, 1)
    for _ in range(300):
        eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(40):
        eng.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    e1.record(ts); torch.cuda.synchronize()
    eng.close()
    return e0.elapsed_time(e1) / 40 * 1e3

def sweep():
    eng = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    def one():
        eng.reset()
        eng.sweep_dev(xs.data_ptr(), total, total, 0, 360, 1, -1, peaks.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
    for _ in range(40): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    eng.close()
    return dt

def show(tag):
    print("%-40s headline %.1f us   noise sweep %.3f ms" % (tag, head(), sweep()), flush=True)

show("fresh"); show("fresh again")
which = sys.argv[1] if len(sys.argv) > 1 else "file"
if which == "file":
    bench.bench_file_host(pkg, np, 0); show("after bench_file_host")
elif which == "pinned":
    h = torch.empty(1 << 26, pin_memory=True); d = torch.empty(1 << 26, device=dev)
    for _ in range(5): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); del h, d; show("after pinned H2D copies")
elif which == "pageable":
    h = torch.empty(1 << 26); d = torch.empty(1 << 26, device=dev)
    for _ in range(5): d.copy_(h)
    torch.cuda.synchronize(); del h, d; show("after pageable H2D copies")
elif which in ("apply_file", "analyze_file", "engine8192"):
    nn, nc2, P2 = 1 << 22, 2, 8192
    xh = (np.random.default_rng(5).random(nn * nc2, dtype=np.float32) - 0.5)
    pr = pkg.PhaseRotate(P2, nc2)
    if which == "apply_file":
        for _ in range(3): pr.apply_file(xh, [90, -90])
    elif which == "analyze_file":
        for _ in range(3):
            pr.reset(); pr.analyze_file(xh, 0, 360, 24); pr.peak_all(0)
    pr.close(); show("after " + which)
elif which == "events":
    ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
    a = torch.empty(1 << 24, device=dev)
    for _ in range(50):
        ev = torch.cuda.Event()
        with torch.cuda.stream(ss[0]): a.add_(1.0); ev.record(ss[0])
        ss[1].wait_event(ev)
        with torch.cuda.stream(ss[1]): a.add_(1.0)
        ev2 = torch.cuda.Event(); ev2.record(ss[1]); ts.wait_event(ev2)
    torch.cuda.synchronize(); del ss; show("after cross-stream event waits")
elif which == "streams":
    ss = [torch.cuda.Stream(device=dev) for _ in range(3)]
    a = torch.empty(1 << 24, device=dev)
    for _ in range(50):
        for s_ in ss:
            with torch.cuda.stream(s_): a.add_(1.0)
    torch.cuda.synchronize(); del ss; show("after three concurrent streams")
show("once more")
