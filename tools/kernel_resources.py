#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / scratch / LDS / occupancy of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage over the
device units of phaserotate.lv2_amd/build.py, with their build flags).  python3 tools/kernel_resources.py [unit.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phaserotate.lv2_amd", "csrc")
UNITS = [("prot_kernels.hip", ["-ffast-math", "-fno-slp-vectorize"]),
         ("prot_olapoly.hip", ["-ffast-math", "-fno-slp-vectorize"]),
         ("prot_sweep.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
         ("prot_kernels_f16.hip", ["-ffast-math"]),
         ("prot_f16mfma.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
         ("prot_olasave512.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=14"]),
         ("prot_olasave.hip", ["-ffast-math", "-fno-slp-vectorize"]),
         ("prot_olasave8k.hip", ["-ffast-math", "-fno-slp-vectorize"]),
         ("prot_olasave16k.hip", ["-ffast-math", "-fno-slp-vectorize"])]
want = set(sys.argv[1:])
print("%-78s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for unit, flags in UNITS:
    if want and unit not in want:
        continue
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(CSRC, unit), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur, rows = None, {}
    for line in r.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = re.sub(r"\(prot::.*$", "", cur).replace("void prot::", "").replace("prot::", "")[:76]
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print("# " + unit)
    for k, v in rows.items():
        print("%-78s %5d %5d %5d %7d %6d %4d" % (k, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1),
                                                  v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", -1)))
