mkdir -p gpurun_out/ab3
timeout 900 python tools/ab_bench.py --rounds 3 --cases poly main stamps2 base47 > gpurun_out/ab3/ab.txt 2>&1; cat gpurun_out/ab3/ab.txt
