#!/usr/bin/env python3
"""HBM traffic of the cfg4 sweep by counter.  On the GPU box, under rocprofv3 (one counter per pass, no trace domains):

    rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/swt/fetch_harmonic -- python3 tools/sweep_traffic.py harmonic 2048
    rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/swt/write_harmonic -- python3 tools/sweep_traffic.py harmonic 2048

runs PASSES complete resident passes (reset, zeroed table, 360 angles) over SECONDS of stereo 96 kHz material and prints how many
channel-frames that was; `python3 tools/sweep_traffic.py --summarize <dir> <tag>` adds the counters of every dispatch of the
engine's own kernels (the Hilbert pass, the list plan, the matrix-core evaluation, the merge; NOT the kernels that generate the
signal) and divides: bytes per channel-frame next to SURVEY 8d's algorithmic 4.  FETCH_SIZE is doubled (gfx950: counted in 64 B
units where the tool assumes 32, MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact; both in KiB."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PASSES, SECONDS = 3, 3600.0  # (the bench's cfg4: 60 min)
OURS = ("olasave_kernel", "olasave8k_kernel", "olasave8k_dyn_kernel", "olasave16k_kernel", "sweep_seed", "olapoly_kernel", "sweep_kernel", "sweep_eval_kernel", "sweep_plan_kernel", "sweep_merge", "sweep_reset", "hist_",
        "ola4096", "zero_regions")


def run(kind, P):
    import numpy as np, torch
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    dev = torch.device("cuda", 0)
    n_chn, total = 2, int(SECONDS * 96000) // P * P
    x = bench._sweep_signal(torch, np, dev, n_chn, 0, total, 1234, kind)
    eng = pkg.Engine(pkg.KIND_CLI, P, n_chn, device=0)
    peaks = torch.zeros((n_chn, 360), device=dev)
    ts = torch.cuda.Stream(device=dev)   # (a stream of the caller's: peaks.zero_() below must be ORDERED with the engine's kernels; on
    torch.cuda.set_stream(ts)            # stream 0 -- the engine's own -- it was only the reset's host wait that kept them apart)
    st = ts.cuda_stream
    torch.cuda.synchronize()
    for _ in range(PASSES):
        eng.reset()
        peaks.zero_()
        eng.sweep_dev(x.data_ptr(), total, total, 0, 360, 1, -1, peaks.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END, st)
    eng.sync()
    torch.cuda.synchronize()
    fu = eng.sweep_stats_fused()
    print(json.dumps({"kind": kind, "blksiz": P, "passes": PASSES, "channel_frames_per_pass": total * n_chn, "fused": fu}))
    eng.close()


def summarize(d, tag):
    res = {"note": "separate --pmc passes of tools/sweep_traffic.py; KiB -> bytes; FETCH_SIZE x 2 on gfx950; every dispatch of the engine's "
                   "kernels in PASSES complete passes, divided by the channel-frames of those passes", "passes": PASSES, "seconds": SECONDS}
    for sub in sorted(glob.glob(os.path.join(d, "fetch_*"))):
        name = os.path.basename(sub)[len("fetch_"):]
        kind, P = name.rsplit("_", 1)
        cf = PASSES * (int(SECONDS * 96000) // int(P) * int(P)) * 2
        per = collections.defaultdict(lambda: [0.0, 0.0, 0])
        for which, col, mul in (("fetch_", 0, 2.0), ("write_", 1, 1.0)):
            for f in glob.glob(os.path.join(d, which + name, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if any(o in k for o in OURS):
                        short = k.split("(")[0].replace("void prot::", "")[:60]
                        per[short][col] += mul * 1024.0 * float(r["Counter_Value"])
                        per[short][2] += 1 if col == 0 else 0
        rd, wr = sum(v[0] for v in per.values()), sum(v[1] for v in per.values())
        res[name] = {"channel_frames": cf, "read_bytes_per_channel_frame": rd / cf, "write_bytes_per_channel_frame": wr / cf,
                     "hbm_bytes_per_channel_frame": (rd + wr) / cf, "algorithmic_bytes_per_channel_frame": 4.0,
                     "ratio_to_algorithmic": (rd + wr) / cf / 4.0,
                     "by_kernel": {k: {"read_per_cf": v[0] / cf, "write_per_cf": v[1] / cf, "dispatches": v[2]} for k, v in sorted(per.items())}}
    json.dump(res, open(os.path.join(ROOT, "profiles", tag + "_sweep_traffic.json"), "w"), indent=1)
    json.dump(res, open(os.path.join(ROOT, "profiles", "sweep_traffic_latest.json"), "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "by_kernel"}) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], int(sys.argv[2]))
