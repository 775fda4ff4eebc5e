mkdir -p gpurun_out/s4
V=phaserotate.lv2_amd/lib/variants
for w in 2 3; do
PROT_POLY_WPS=$w PROT_WG_PER_CU=$w PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s4/stamps_poly_$w.txt 2>&1; cat gpurun_out/s4/stamps_poly_$w.txt
done
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py cli > gpurun_out/s4/stamps_cli_3.txt 2>&1; cat gpurun_out/s4/stamps_cli_3.txt
for r in 1 2; do
PROT_POLY_WPS=3 PROT_WG_PER_CU=3 timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k tun >> gpurun_out/s4/ab_wps3.txt 2>&1
PROT_POLY_WPS=4 PROT_WG_PER_CU=3 timeout 300 python tools/ab_bench.py --rounds 1 --cases poly,cli,cli1024,poly_96k tun >> gpurun_out/s4/ab_wps4.txt 2>&1
done
cat gpurun_out/s4/ab_wps3.txt gpurun_out/s4/ab_wps4.txt
