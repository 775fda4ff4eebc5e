"""Same-box interleaved A/B of two libraries on the sweep, four signals, pruning on; plus the unpruned time once.
usage: sweep_ab2.py <tagA>=<libA.so or 'main'> <tagB>=<libB.so> [rounds]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
pkg.set_test_option(pkg.TEST_SWEEP_UNPRUNED, int(os.environ.get("SWEEP_AB_UNPRUNED", "0")))
dev = torch.device("cuda:0")
P, nc = 2048, 2
n = int(600 * 96000) // P * P
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.rand(nc * n, generator=g, device=dev) - 0.5
t = torch.arange(n, device=dev, dtype=torch.float64) / 96000.0
harm = sum(torch.cos(2 * np.pi * k * 110.0 * t + 0.3 * k) / k for k in range(1, 33)).float()
harm = 0.8 * harm / harm.abs().max()
sigs = {"noise": noise, "harm": torch.cat([harm + 0.05 * noise[:n], 0.7 * harm + 0.05 * noise[n:]]),
        "quiet": torch.where((torch.arange(nc * n, device=dev) %% n) < 96000, noise, 1e-3 * noise),
        "grow": noise * torch.linspace(0.01, 1.0, n, device=dev).repeat(nc)}
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
args = sys.argv[1:]
R = int(args[-1]) if args and args[-1].isdigit() else 4
variants = []
for a in args:
    if "=" in a:
        tag, lib = a.split("=", 1)
        variants.append((tag, {} if lib == "main" else {"PROT_LIB_PATH": os.path.abspath(lib)}))
variants.append(("unpruned(main)", {"SWEEP_AB_UNPRUNED": "1"}))
res = {v[0]: {} for v in variants}
for r in range(R):
    for name, env in variants[r % len(variants):] + variants[:r % len(variants)]:
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, "-c", CHILD], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(name, "FAILED", p.stdout[-800:]); continue
        f = line[0].split()[1:]
        for k, v in zip(f[0::2], f[1::2]):
            res[name].setdefault(k, []).append(float(v))
for name, _ in variants:
    print("%-16s " % name + "  ".join("%s %.3f (%.3f-%.3f)" % (k, np.median(v), min(v), max(v)) for k, v in res[name].items()))
