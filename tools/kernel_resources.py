#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/LDS/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phaserotate.lv2_amd", "csrc")
flags = sys.argv[1:] or ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]
src = os.path.join(CSRC, "prot_kernels.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + flags,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in r.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()[:60]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-62s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for k, v in rows.items():
    print("%-62s %5d %5d %5d %7d %6d %4d" % (k, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1),
                                              v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", -1)))
