mkdir -p gpurun_out/s23
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s23/tests.txt 2>&1
tail -4 gpurun_out/s23/tests.txt
