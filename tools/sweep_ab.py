"""Interleaved A/B of the sweep on ONE box: library before the pruning change vs now (pruning on / off), two signals.
Each measurement in its own process (a library is loaded once per process); R rounds, order rotated."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
P, nc = 2048, 2
n = int(600 * 96000) // P * P
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.rand(nc * n, generator=g, device=dev) - 0.5
t = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
harm = sum(torch.cos(2 * np.pi * k * 110.0 * t + 0.3 * k) / k for k in range(1, 33)).float()
harm = 0.8 * harm / harm.abs().max()
sigs = {"noise": noise, "harm": torch.cat([harm + 0.05 * noise[:n], 0.7 * harm + 0.05 * noise[n:]])}
out = []
for name, x in sigs.items():
    e = pkg.Engine(pkg.KIND_CLI, P, nc)
    pk = torch.zeros(nc * 360, device=dev)
    ts_ = []
    for rep in range(6):
        e.reset(); pk.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
        torch.cuda.synchronize()
        ts_.append((time.perf_counter() - t0) * 1e3)
    out.append("%%s %%.3f" %% (name, sorted(ts_[1:])[2]))
    e.close()
print("RESULT " + " ".join(out))
''' % ROOT
V = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants")
variants = [("before", {"PROT_LIB_PATH": os.path.join(V, "libprot_hip_before.so")}),
            ("now, prune off", {"PROT_SWEEP_PRUNE": "0"}), ("now, prune on", {})]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = {v[0]: {"noise": [], "harm": []} for v in variants}
for r in range(R):
    for name, env in variants[r % 3:] + variants[:r % 3]:
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, "-c", CHILD], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(name, "FAILED", p.stdout[-1500:]); continue
        f = line[0].split()
        res[name]["noise"].append(float(f[2])); res[name]["harm"].append(float(f[4]))
for name, _ in variants:
    for sig in ("noise", "harm"):
        v = np.array(res[name][sig])
        print("%-16s %-6s median %.3f ms  (min %.3f  max %.3f, n=%d)" % (name, sig, np.median(v), v.min(), v.max(), len(v)))
