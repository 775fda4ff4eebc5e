# GPU regression: all -m gpu tests + an interleaved A/B of the shipped library against the round-1 build
mkdir -p gpurun_out/chk
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/chk/pytest.txt 2>&1; tail -3 gpurun_out/chk/pytest.txt
timeout 600 python tools/ab_bench.py --rounds 3 --cases poly,cli,cli1024,poly_96k,cli8192,f16_poly main > gpurun_out/chk/ab.txt 2>&1; cat gpurun_out/chk/ab.txt
