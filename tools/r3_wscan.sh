mkdir -p gpurun_out/r3c
for w in "1.4,1.1,0.9,0.6" "1.6,1.1,0.8,0.5" "1.7,1.05,0.75,0.5" "1.8,1.1,0.7,0.4" "1.9,1.0,0.7,0.4" "2.0,1.0,0.6,0.4" "1.7,1.2,0.7,0.4" "1.6,1.2,0.8,0.4"; do
  echo "== W $w"; python tools/sv_timeline.py clk PROT_RANK_W=$w | grep -E "span|ids"
done > gpurun_out/r3c/wscan.txt 2>&1
cat gpurun_out/r3c/wscan.txt
