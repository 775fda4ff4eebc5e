"""Randomised differential test of the sweep: pruned against unpruned tables (bit for bit) over random sizes, channel counts,
angle ranges / strides, block sizes, signal shapes (bursts, silences, steps, tones, noise), chunked and one-shot calls.
python3 tools/sweep_fuzz.py [cases] [seed] [big]     (big: every case long enough for the fused route -- 2^20 frames and more --,
block sizes up to 32768)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

def signal(n, nc):
    t = np.arange(n)
    x = np.zeros((n, nc), np.float32)
    for c in range(nc):
        kind = rng.integers(0, 6)
        amp = rng.uniform(0.05, 0.9)
        if kind == 0:
            x[:, c] = amp * (rng.random(n) - 0.5)
        elif kind == 1:
            x[:, c] = amp * np.sin(2 * np.pi * rng.uniform(0.001, 0.2) * t)
        elif kind == 2:
            x[:, c] = amp * np.sin(2 * np.pi * rng.uniform(0.001, 0.05) * t) * np.linspace(0.1, 1, n)
        elif kind == 3:  # bursts in silence
            for _ in range(rng.integers(1, 6)):
                a = rng.integers(0, n); b = min(n, a + rng.integers(1, 5000))
                x[a:b, c] = amp * (rng.random(b - a) - 0.5)
        elif kind == 4:  # steps of level
            lev = rng.uniform(0.01, 1.0, 8)
            seg = np.minimum((t * 8 // max(n, 1)), 7)
            x[:, c] = lev[seg] * 0.5 * np.sign(np.sin(2 * np.pi * 0.01 * t)) + 0.01 * (rng.random(n) - 0.5)
        else:
            x[:, c] = amp * 0.25
    return np.ascontiguousarray(x)

bad = 0
for k in range(cases):
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    P = int(rng.choice([1024, 2048, 2048, 4096, 8192, 16384, 32768] if big else [1024, 2048, 2048, 4096, 8192]))
    nc = int(rng.choice([1, 1, 2, 2, 3, 4]))
    n = int(rng.integers(1 << 20, 5 << 20)) if big else int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 200000), rng.integers(200000, 3000000)]))
    st = int(rng.choice([1, 1, 2, 3, 5, 8, 24]))
    a0 = int(rng.choice([0, 0, 1, rng.integers(0, 300)]))
    a1 = int(min(360, a0 + rng.choice([1, 40, 100, 360])))
    x = signal(n, nc)
    tabs = []
    for prune in (1, 0):
        pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 1 - prune)
        g = pkg.PhaseRotate(P, nc)
        if rng.random() < 0.5 or prune == 0 and False:
            pass
        g.analyze_file(x, a0, a1, st)
        if k % 3 == 0:  # a second pass over the same file: the table of the first one is the bound now
            g.analyze_file(x, max(0, a0 - 7), min(360, a1 + 9), 1)
        tabs.append(np.array([[g.peak(c, a) for a in range(360)] for c in range(nc)], np.float32))
        g.close()
    ok = tabs[0].tobytes() == tabs[1].tobytes()
    if not ok:
        bad += 1
        d = np.argwhere(tabs[0] != tabs[1])
        print("MISMATCH case %d: P %d nc %d n %d angles %d..%d step %d: %d entries differ, first %s" % (k, P, nc, n, a0, a1, st, len(d), d[:3].tolist()), flush=True)
pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, 0)
print("%d cases, %d mismatches" % (cases, bad))
