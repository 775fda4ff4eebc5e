mkdir -p gpurun_out/r3c
timeout 1500 python tools/ab_bench.py --rounds ${ROUNDS:-2} --cases ${CASES:-poly} "$@" > gpurun_out/r3c/ab.txt 2>&1; cat gpurun_out/r3c/ab.txt
