mkdir -p gpurun_out/ab1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -m gpu > gpurun_out/ab1/pytest.txt 2>&1; tail -3 gpurun_out/ab1/pytest.txt
timeout 900 python tools/ab_bench.py --rounds 4 --cases poly,cli,poly_96k,cli8192 base47 main nowbs notm > gpurun_out/ab1/ab.txt 2>&1; cat gpurun_out/ab1/ab.txt
