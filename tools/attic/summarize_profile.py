#!/usr/bin/env python3
"""Turn the rocprofv3 output that a gpurun call left under gpurun_out/ into the
small, tracked summaries under profiles/:

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc.json             FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/traffic_latest.json        HBM bytes per launch of the dominant kernel (read by bench.py)

usage: tools/summarize_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [kernel-substring]
FETCH_SIZE is doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950
(the counter tallies 128-B fabric requests at 64 B); WRITE_SIZE is exact.
Both counters are in KiB.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PROF = os.path.join(ROOT, "profiles")


def counters(d, name):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in agg.items()}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    os.makedirs(PROF, exist_ok=True)
    st = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(st, os.path.join(PROF, tag + "_kernel_stats.csv"))
    print(open(st).read()[:1500])
    if len(sys.argv) >= 5:
        fetch, write = counters(sys.argv[3], "FETCH_SIZE"), counters(sys.argv[4], "WRITE_SIZE")
        sub = sys.argv[5] if len(sys.argv) > 5 else "ola4096_kernel<4"
        out = {"note": "separate --pmc passes; KiB; FETCH_SIZE must be doubled on gfx950 "
                       "(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact",
               "FETCH_SIZE": fetch, "WRITE_SIZE": write}
        json.dump(out, open(os.path.join(PROF, tag + "_pmc.json"), "w"), indent=1)
        fk = [k for k in fetch if sub in k]
        wk = [k for k in write if sub in k]
        if fk and wk:
            rd = 2.0 * fetch[fk[0]]["mean_KiB"] * 1024
            wr = write[wk[0]]["mean_KiB"] * 1024
            t = {"kernel": fk[0], "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                 "source": tag + "_pmc.json (FETCH_SIZE x2 + WRITE_SIZE)"}
            json.dump(t, open(os.path.join(PROF, "traffic_latest.json"), "w"), indent=1)
            print(json.dumps(t, indent=1))


if __name__ == "__main__":
    main()
