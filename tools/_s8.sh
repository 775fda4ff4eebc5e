mkdir -p gpurun_out/s8
V=phaserotate.lv2_amd/lib/variants
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/s8/a.txt; cat gpurun_out/s8/a.txt
PROT_POLY_WPS=3 PROT_WG_PER_CU=6 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/s8/b.txt; cat gpurun_out/s8/b.txt
for r in 1 2; do for w in 3 6 9 12; do echo "WG_PER_CU=$w"; PROT_POLY_WPS=3 PROT_WG_PER_CU=$w timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024 tun 2>&1 | tail -1; done; done > gpurun_out/s8/ab.txt; cat gpurun_out/s8/ab.txt
for w in 2 4 6 8; do echo "WPS2 WG_PER_CU=$w"; PROT_WG_PER_CU=$w timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli8192 main 2>&1 | tail -1; done > gpurun_out/s8/ab2.txt; cat gpurun_out/s8/ab2.txt
