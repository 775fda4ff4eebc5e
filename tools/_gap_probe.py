import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
tiny = torch.zeros(64, device=dev)
big = torch.zeros(1 << 26, device=dev)
def chain(tag):
    for _ in range(200): tiny.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(2000): tiny.add_(1.0)
    e1.record(ts); torch.cuda.synchronize()
    a = e0.elapsed_time(e1) / 2000 * 1e3
    e0.record(ts)
    for _ in range(200): big.add_(1.0)
    e1.record(ts); torch.cuda.synchronize()
    b = e0.elapsed_time(e1) / 200 * 1e3
    print("%-10s tiny kernel chain %.2f us per kernel;  256 MiB add_ %.1f us per kernel" % (tag, a, b), flush=True)
chain("fresh"); chain("fresh")
mode = sys.argv[1] if len(sys.argv) > 1 else "chain3"
ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
a = torch.empty(1 << 24, device=dev)
for _ in range(50):
    if mode == "chain3":      # ss0 -> ss1 -> ts
        ev = torch.cuda.Event()
        with torch.cuda.stream(ss[0]): a.add_(1.0); ev.record(ss[0])
        ss[1].wait_event(ev)
        with torch.cuda.stream(ss[1]): a.add_(1.0)
        ev2 = torch.cuda.Event(); ev2.record(ss[1]); ts.wait_event(ev2)
    elif mode == "others":    # ss0 -> ss1, the timed stream not involved
        ev = torch.cuda.Event()
        with torch.cuda.stream(ss[0]): a.add_(1.0); ev.record(ss[0])
        ss[1].wait_event(ev)
        with torch.cuda.stream(ss[1]): a.add_(1.0)
    elif mode == "into_ts":   # ss0 -> ts
        ev = torch.cuda.Event()
        with torch.cuda.stream(ss[0]): a.add_(1.0); ev.record(ss[0])
        ts.wait_event(ev)
    elif mode == "from_ts":   # ts -> ss0
        ev = torch.cuda.Event(); a.add_(1.0); ev.record(ts)
        ss[0].wait_event(ev)
        with torch.cuda.stream(ss[0]): a.add_(1.0)
torch.cuda.synchronize()
chain("after " + mode); chain("after " + mode)
