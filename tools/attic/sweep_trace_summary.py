"""Per-pass sweep kernel time from a rocprofv3 kernel trace of tools/sweep_prune_probe.py (5 signals x {pruned, plain} x 2)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3_sweep/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sw = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "sweep_kernel" in r["Kernel_Name"]]
names = ["uniform noise", "harmonic stack + noise", "loud first second", "growing noise", "steady tone"]
for k in range(0, len(sw), 4):
    ch = sw[k:k + 4]
    print("%-24s pruned %s us   plain %s us" % (names[k // 4] if k // 4 < len(names) else "?",
          ["%.0f" % d for n, d in ch if "true" in n], ["%.0f" % d for n, d in ch if "false" in n]))
