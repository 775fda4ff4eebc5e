mkdir -p gpurun_out/s19
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_apply or long_fir or fused or block_size" > gpurun_out/s19/t1.txt 2>&1
tail -12 gpurun_out/s19/t1.txt
