mkdir -p gpurun_out/s6
V=phaserotate.lv2_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s6/pytest.txt 2>&1; tail -2 gpurun_out/s6/pytest.txt
PROT_LIB_PATH=$V/libprot_hip_tun.so PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s6/pytest_wps3.txt 2>&1; tail -2 gpurun_out/s6/pytest_wps3.txt
for r in 1 2 3; do
timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k,cli8192 r1 main >> gpurun_out/s6/ab_wps2.txt 2>&1
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k tun >> gpurun_out/s6/ab_wps3.txt 2>&1
done
cat gpurun_out/s6/ab_wps2.txt; echo; cat gpurun_out/s6/ab_wps3.txt
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s6/stamps_poly_3.txt 2>&1; cat gpurun_out/s6/stamps_poly_3.txt
