mkdir -p gpurun_out/s5
V=phaserotate.lv2_amd/lib/variants
PROT_LIB_PATH=$V/libprot_hip_wst.so PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s5/pytest_wst.txt 2>&1; tail -3 gpurun_out/s5/pytest_wst.txt
for r in 1 2 3; do
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024 tun wst >> gpurun_out/s5/ab_wps3.txt 2>&1
timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024 tun wst >> gpurun_out/s5/ab_wps2.txt 2>&1
done
cat gpurun_out/s5/ab_wps3.txt; echo; cat gpurun_out/s5/ab_wps2.txt
