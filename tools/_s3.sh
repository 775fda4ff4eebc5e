mkdir -p gpurun_out/s3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s3/pytest.txt 2>&1; tail -3 gpurun_out/s3/pytest.txt
V=phaserotate.lv2_amd/lib/variants
PROT_LIB_PATH=$V/libprot_hip_tun.so PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s3/pytest_wps3.txt 2>&1; tail -3 gpurun_out/s3/pytest_wps3.txt
for r in 1 2; do
timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k r1 main tun >> gpurun_out/s3/ab.txt 2>&1
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k tun >> gpurun_out/s3/ab_wps3.txt 2>&1
done
cat gpurun_out/s3/ab.txt gpurun_out/s3/ab_wps3.txt
