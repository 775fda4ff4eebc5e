timeout 900 python tools/ab_bench.py --rounds 3 --cases poly main stream comp valu lds novalu nolds > gpurun_out/abl_r02.txt 2>&1; cat gpurun_out/abl_r02.txt
