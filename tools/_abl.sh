timeout 900 python tools/ab_bench.py --rounds 3 --cases poly,cli main cached > gpurun_out/abl_r02d.txt 2>&1; cat gpurun_out/abl_r02d.txt
