#!/usr/bin/env python3
"""Where a workgroup of sweep_eval_kernel spends its time (round 5): a shard pass of the two-phase sweep (tools/shard_trace.py's
set-up) on the diagnostic build `python3 tools/ab_build.py sw:evst:-DPROT_EVAL_STAMPS`, whose kernel stamps the 100 MHz clock at
entry, after its prologue, the first claim, the first stage, each of the first twelve steps, the loop's end, the rings' rest
and the end.
    PROT_LIB_PATH=phaserotate.lv2_amd/lib/variants/libprot_hip_evst.so python3 tools/eval_stamps.py [blksiz [shards [which [kind]]]]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    a = sys.argv[1:]
    P, world, which, kind = int(a[0]) if a else 2048, int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 3, a[3] if len(a) > 3 else "harmonic"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import shard_trace
    from __graft_entry__ import load_package
    pkg = load_package()
    lib = pkg.load_library()
    lib.prot_debug_eval_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.prot_debug_eval_stamps(None, 1)
    shard_trace.run(P, world, which, kind)
    buf = np.zeros(4096 * 24, np.uint64)
    assert lib.prot_debug_eval_stamps(buf.ctypes.data, 0) == 0
    st = buf.reshape(4096, 24).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    n = st[:, 19]
    seg = {"entry (after the launch's first workgroup)": us(st[:, 0]), "prologue": (st[:, 1] - st[:, 0]) / 100.0, "first claim": (st[:, 2] - st[:, 1]) / 100.0,
           "first stage": (st[:, 3] - st[:, 2]) / 100.0, "loop": (st[:, 16] - st[:, 3]) / 100.0, "rest of the rings": (st[:, 17] - st[:, 16]) / 100.0,
           "statistics + final reduction": (st[:, 18] - st[:, 17]) / 100.0, "end (after the launch's first workgroup)": us(st[:, 18]), "steps": n.astype(float)}
    out = {"workgroups": int(len(st)), "blksiz": P, "shards": world, "kind": kind}
    for k, v in seg.items():
        out[k] = {"mean": round(float(v.mean()), 2), "p50": round(float(np.median(v)), 2), "p95": round(float(np.percentile(v, 95)), 2), "max": round(float(v.max()), 2)}
    per = []
    for s in range(12):
        m = n > s
        if m.sum() == 0:
            break
        prev = st[m, 3] if s == 0 else st[m, 3 + s]
        d = (st[m, 4 + s] - prev) / 100.0
        per.append({"step": s, "workgroups": int(m.sum()), "mean_us": round(float(d.mean()), 2), "p95_us": round(float(np.percentile(d, 95)), 2)})
    out["per_step"] = per
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
