mkdir -p gpurun_out/r3b
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3b/pytest.txt 2>&1; tail -15 gpurun_out/r3b/pytest.txt
timeout 900 python tools/ab_bench.py --rounds 3 --cases poly,poly_96k,cli,cli1024 main main@PROT_NO_SAVE=1 > gpurun_out/r3b/ab.txt 2>&1; cat gpurun_out/r3b/ab.txt
