import os, sys
sys.path.insert(0, os.getcwd())
import bench, json, numpy as np
from __graft_entry__ import load_package
pkg = load_package()
r = bench.bench_run_host(pkg, np, 0)
print(os.environ.get("PROT_LIB_PATH", "shipped").split("_")[-1], "us_per_run %.2f worst %.1f paced-blocking %.2f deferred %.2f two %.2f four %.2f mono+stereo %.2f" % (
    r["us_per_run"], r["worst_single_call_us"], r["rt_mode"]["blocking_us_per_run"], r["rt_mode"]["us_per_run"],
    r["instances_in_turn"]["two_mono_48k"]["us_per_run"], r["instances_in_turn"]["four_mono_48k"]["us_per_run"], r["instances_in_turn"]["mono_48k_and_stereo_96k"]["us_per_run"]))
