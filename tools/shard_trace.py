#!/usr/bin/env python3
"""Timeline of ONE shard pass of cfg4 (shard 3 of 8 of the bench's harmonic material, blksiz 2048 or argv[1]): every kernel with
its start offset, duration and the gap in front of it.
    rocprofv3 --output-format csv --kernel-trace -d gpurun_out/shtr -- python3 tools/shard_trace.py [blksiz [shards [which [harmonic|noise]]]]
(shards = 1, which = 0: the whole file)
    python3 tools/shard_trace.py --summarize gpurun_out/shtr"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(P, world=8, which=3, kind="harmonic"):
    import numpy as np, torch
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    from phaserotate_lv2_amd import sweep_dist
    dev = torch.device("cuda", 0)
    n_chn, total = 2, 3600 * 96000 // P * P
    eng = pkg.Engine(pkg.KIND_CLI, P, n_chn, device=0)
    H = int(eng.info.history)
    plan8 = sweep_dist.plan_shards(total, P, world, align=eng.shard_align_blocks())
    c0, c1 = plan8[which]
    two_phase = world > 1 and os.environ.get("SHARD_ONE_PHASE") != "1"
    peaks = torch.zeros((n_chn, 360), device=dev)
    gseed = torch.zeros((n_chn, 360), device=dev)
    ts = torch.cuda.Stream(device=dev)   # (a stream of the caller's, so that peaks.zero_() is ordered with the engine's kernels)
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    if two_phase:
        # the table the first all-reduce leaves: the seeds of ALL shards of the whole file (as bench.py's predicted_8gpu computes it)
        xa = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, kind)
        for r, (b0, b1) in enumerate(plan8):
            xs = xa[:, b0 * P:b1 * P]
            fl0 = (pkg.F_STREAM_START if r == 0 else 0) | (pkg.F_STREAM_END if r == world - 1 else 0)
            eng.reset()
            peaks.zero_()
            if r > 0:
                hh = xa[:, b0 * P - H:b0 * P].contiguous()
                eng.prime_dev(hh.data_ptr(), H, H, st)
            eng.sweep_seed_dev(xs.data_ptr(), xa.shape[1], xs.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl0, max(8, 64 // world), st)
            torch.cuda.synchronize()
            gseed = torch.maximum(gseed, peaks)
        x = xa[:, c0 * P:c1 * P].contiguous()
        halo = xa[:, c0 * P - H:c0 * P].contiguous() if which > 0 else None
        del xa
    else:
        x = bench._sweep_signal(torch, np, dev, n_chn, c0 * P, (c1 - c0) * P, 1234, kind)
        halo = bench._sweep_signal(torch, np, dev, n_chn, c0 * P - H, H, 4321, kind).contiguous() if which > 0 else None
    import time
    for k in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset()
        peaks.zero_()
        if halo is not None:
            sweep_dist.prime_history(eng, halo, st)
        fl = (pkg.F_STREAM_START if which == 0 else 0) | (pkg.F_STREAM_END if which == world - 1 else 0)
        if two_phase:
            eng.sweep_seed_dev(x.data_ptr(), x.shape[1], x.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl, max(8, 64 // world), st)
            peaks.copy_(gseed)
        eng.sweep_dev(x.data_ptr(), x.shape[1], x.shape[1], 0, 360, 1, -1, peaks.data_ptr(), fl, st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("pass %d: host calls %.1f us, to idle %.1f us" % (k, (t1 - t0) * 1e6, (time.perf_counter() - t0) * 1e6))
    eng.close()


def summarize(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last pass = from the last zero_regions (prot_engine_reset) on
    last = max(i for i, r in enumerate(rows) if "zero_regions" in r["Kernel_Name"] or "FillFunctor" in r["Kernel_Name"])
    while last > 0 and int(rows[last]["Start_Timestamp"]) - int(rows[last - 1]["End_Timestamp"]) < 30000 and \
            ("zero_regions" in rows[last - 1]["Kernel_Name"] or "FillFunctor" in rows[last - 1]["Kernel_Name"]):
        last -= 1
    t0, prev = int(rows[last]["Start_Timestamp"]), None
    for r in rows[last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, ((s - prev) / 1e3) if prev else 0.0, (e - s) / 1e3,
                                                  r["Kernel_Name"].replace("void prot::", "").split("(")[0][:70]))
        prev = e
    print("pass: %.1f us from the first kernel's start to the last one's end" % ((prev - t0) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        a = sys.argv[1:]
        run(int(a[0]) if len(a) > 0 else 2048, int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 3, a[3] if len(a) > 3 else "harmonic")
