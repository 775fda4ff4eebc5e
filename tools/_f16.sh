timeout 600 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for w in "1.5,1.1,0.8,0.6" "1,1,1,1" "1.4,1.1,0.9,0.6" "1.6,1.1,0.8,0.5" "1.3,1.1,0.9,0.7"; do echo "W16=$w"; PROT_RANK_W16=$w timeout 300 python tools/ab_bench.py --rounds 3 --cases f16_poly,f16_cli,poly main 2>&1 | tail -1; done > gpurun_out/f16.txt 2>&1; cat gpurun_out/f16.txt
