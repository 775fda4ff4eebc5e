#!/usr/bin/env python3
"""bench.py's lv2_run_host section alone (run() from C through the resident server: blocking, instances in turn, deferred)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, json, numpy as np
from __graft_entry__ import load_package
pkg = load_package()
r = bench.bench_run_host(pkg, np, 0)
print(json.dumps({k: r[k] for k in r if k != "instances_in_turn"}))
print(json.dumps(r.get("instances_in_turn")))
