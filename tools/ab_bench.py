"""Time variant libraries (tools/ab_build.py) against each other on ONE box, interleaved:
`python tools/ab_bench.py [--rounds 3] [--cases p2,cli] tagA tagB ...` ("main" = the shipped library)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
cases = {"b4k": (pkg.KIND_LV2, 48000, 32), "b4k_96k": (pkg.KIND_LV2, 96000, 32), "hop1024": (pkg.KIND_LV2, 48000, 1), "p2": (pkg.KIND_LV2, 48000, 2), "p2_96k": (pkg.KIND_LV2, 96000, 2),
         "poly": (pkg.KIND_LV2, 48000, 0), "poly_96k": (pkg.KIND_LV2, 96000, 0), "poly_192k": (pkg.KIND_LV2, 192000, 0),
         "polyf_48k": (pkg.KIND_LV2, 48000, 4), "polyf_96k": (pkg.KIND_LV2, 96000, 4), "polyf_cli2048": (pkg.KIND_CLI, 2048, 4),
         "cli": (pkg.KIND_CLI, 2048, 0), "cli1024": (pkg.KIND_CLI, 1024, 0), "cli_direct": (pkg.KIND_CLI, 2048, 2), "cli4096": (pkg.KIND_CLI, 4096, 0), "cli8192": (pkg.KIND_CLI, 8192, 0),
         "cli8192_direct": (pkg.KIND_CLI, 8192, 2), "cli32768": (pkg.KIND_CLI, 32768, 0), "cli32768_direct": (pkg.KIND_CLI, 32768, 2)}
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
out = []
xh, yh = x.half(), torch.empty(n, device=dev, dtype=torch.float16)
for name in sys.argv[1].split(","):
    if name.startswith("f16_"):  # fp16 sample path of the named case
        kind, param, fl = cases[name[4:]]
        e = pkg.Engine(kind, param, 1, flags=fl)
        call = lambda: e.apply_dev_f16(xh.data_ptr(), n, n, yh.data_ptr(), n, n, rot, st)
    else:
        kind, param, fl = cases[name]
        e = pkg.Engine(kind, param, 1, flags=fl)
        call = lambda: e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    for _ in range(150):   # (the clocks settle over ~150 launches after any idle time)
        call()
    torch.cuda.synchronize()
    ts_ = []
    for rep in range(5):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(40):
            call()
        ev1.record(); torch.cuda.synchronize()
        ts_.append(ev0.elapsed_time(ev1) / 40)
    out.append("%%s %%.4f" %% (name, sorted(ts_)[2]))
    e.close()
print("RESULT " + " ".join(out))
''' % ROOT
args = sys.argv[1:]
rounds, cases = 3, "p2"
while args and args[0].startswith("--"):
    if args[0] == "--rounds": rounds = int(args[1])
    elif args[0] == "--cases": cases = args[1]
    args = args[2:]
tags = args or ["main"]
res = {t: [] for t in tags}
for r in range(rounds):
    for t in tags:
        env = dict(os.environ)
        lib_tag = t
        if "@" in t:  # "<lib tag>@NAME=VAL[,NAME=VAL]": environment switches of one library (e.g. main@PROT_NO_SAVE=1)
            lib_tag, _, ev = t.partition("@")
            for kv in ev.split(","):
                k, _, v = kv.partition("=")
                env[k] = v
        if lib_tag != "main":
            env["PROT_LIB_PATH"] = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "variants", "libprot_hip_%s.so" % lib_tag)
        p = subprocess.run([sys.executable, "-c", CHILD, cases], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(t, "FAILED", p.stdout[-2000:]); continue
        res[t].append(line[0][7:])
for t in tags:
    print("%-16s %s" % (t, " | ".join(res[t])), flush=True)
