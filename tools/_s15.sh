timeout 600 python tools/ab_bench.py --rounds 3 --cases poly_96k,polyf_96k,cli4096,poly_192k,cli32768 main 2>&1 | tail -1
