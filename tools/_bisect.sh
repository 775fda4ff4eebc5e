V=phaserotate.lv2_amd/lib/variants
for tag in main tearly notm noelse nowbs base47; do
  if [ $tag = main ]; then unset PROT_LIB_PATH; else export PROT_LIB_PATH=$V/libprot_hip_$tag.so; fi
  echo "== $tag"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
done > gpurun_out/bisect.txt 2>&1; cat gpurun_out/bisect.txt
