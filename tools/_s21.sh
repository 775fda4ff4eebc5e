mkdir -p gpurun_out/s21
timeout 600 python3 tools/_dbg16.py 16384 3600 > gpurun_out/s21/dbg16.txt 2>&1
cat gpurun_out/s21/dbg16.txt | grep -v amdgpu.ids
