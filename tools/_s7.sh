mkdir -p gpurun_out/s7
V=phaserotate.lv2_amd/lib/variants
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s7/stamps_poly_3.txt 2>&1; cat gpurun_out/s7/stamps_poly_3.txt
PROT_POLY_WPS=2 PROT_WG_PER_CU=2 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s7/stamps_poly_2.txt 2>&1; cat gpurun_out/s7/stamps_poly_2.txt
