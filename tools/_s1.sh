set -x
mkdir -p gpurun_out/s1
V=phaserotate.lv2_amd/lib/variants
PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py poly > gpurun_out/s1/stamps_poly.txt 2>&1
PROT_LIB_PATH=$V/libprot_hip_stamps.so timeout 300 python tools/phase_stamps.py cli > gpurun_out/s1/stamps_cli.txt 2>&1
timeout 300 python tools/power_probe.py poly > gpurun_out/s1/power.txt 2>&1
timeout 900 python tools/ab_bench.py --rounds 3 --cases poly main stream comp valu lds novalu nolds > gpurun_out/s1/abl.txt 2>&1
for d in 0 60000; do PROT_DBG_DYNLDS=$d timeout 300 python tools/ab_bench.py --rounds 2 --cases poly,cli,cli_direct tun > gpurun_out/s1/occ_$d.txt 2>&1; done
PROT_DBG_DYNLDS=20000 timeout 300 python tools/ab_bench.py --rounds 2 --cases cli_direct tun > gpurun_out/s1/occ_20000.txt 2>&1
cat gpurun_out/s1/*.txt
