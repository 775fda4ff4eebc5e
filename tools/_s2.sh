mkdir -p gpurun_out/s2
V=phaserotate.lv2_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s2/pytest.txt 2>&1; tail -3 gpurun_out/s2/pytest.txt
timeout 900 python tools/ab_bench.py --rounds 3 --cases poly,cli,cli8192,poly_192k,f16_poly r1 main > gpurun_out/s2/ab.txt 2>&1; cat gpurun_out/s2/ab.txt
PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s2/stamps_poly.txt 2>&1; cat gpurun_out/s2/stamps_poly.txt
