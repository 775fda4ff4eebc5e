V=phaserotate.lv2_amd/lib/variants
for w in 1.45,0.95,0.6 1.5,0.95,0.55 1.55,0.95,0.5 1.5,1.0,0.5 1.4,1.0,0.6 1.5,0.9,0.6; do
echo "== W=$w"
PROT_RANK_W=$w PROT_LIB_PATH=$V/libprot_hip_stamps2.so timeout 300 python tools/xcd_timeline.py 4 2>&1 | grep -v "amdgpu.ids" | tail -3 | cut -c1-200
done > gpurun_out/xcd_timeline_w.txt; cat gpurun_out/xcd_timeline_w.txt
for w in 1.45,0.95,0.6 1.5,0.95,0.55 1.55,0.95,0.5 1.5,1.0,0.5 1.4,1.0,0.6 1.5,0.9,0.6; do
  echo "W=$w"; PROT_RANK_W=$w timeout 300 python tools/ab_bench.py --rounds 4 --cases poly,cli main 2>&1 | tail -1
done > gpurun_out/w2.txt 2>&1; cat gpurun_out/w2.txt
