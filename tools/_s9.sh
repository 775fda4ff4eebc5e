V=phaserotate.lv2_amd/lib/variants
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/s9_a.txt; cat gpurun_out/s9_a.txt
