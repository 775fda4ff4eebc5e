mkdir -p gpurun_out/s14
for w in "1.45,0.95,0.6" "1.6,0.9,0.5" "1.5,1.0,0.5" "1.7,0.85,0.45" "1.55,1.0,0.45" "1.4,1.1,0.5" "1.8,0.8,0.4"; do
echo "W=$w"; PROT_RANK_W=$w timeout 300 python tools/ab_bench.py --rounds 2 --cases poly,cli,cli1024 main 2>&1 | tail -1
done > gpurun_out/s14/ab2.txt; cat gpurun_out/s14/ab2.txt
