V=phaserotate.lv2_amd/lib/variants
PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly 2>&1 | grep -v "amdgpu.ids" > gpurun_out/stamps_now.txt; cat gpurun_out/stamps_now.txt
