import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split at the elementwise kernels of the probe
idx = [i for i, r in enumerate(rows) if "elementwise" in r["Kernel_Name"] and "add" in r["Kernel_Name"].lower()]
cut0, cut1 = (idx[0], idx[-1]) if idx else (len(rows), len(rows))
for tag, part in (("fresh", rows[:cut0]), ("after", rows[cut1 + 1:])):
    part = [r for r in part if "sweep" in r["Kernel_Name"] or "olasave" in r["Kernel_Name"]]
    part = part[-90:]   # the last passes of the phase
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part)
    span = int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(part, part[1:])]
    by = {}
    for r in part:
        k = r["Kernel_Name"].split("(")[0][-40:]
        by.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(tag, "kernels", len(part), "busy %.1f us" % (busy / 1e3), "span %.1f us" % (span / 1e3), "mean gap %.2f us" % (sum(gaps) / len(gaps) / 1e3), "max gap %.1f" % (max(gaps) / 1e3))
    for k, v in by.items():
        print("    %-42s n %3d  mean %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
