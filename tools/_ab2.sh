mkdir -p gpurun_out/ab2
timeout 900 python tools/ab_bench.py --rounds 3 --cases poly,cli,cli1024,poly_96k,cli8192,poly_192k,cli32768 main wbs2 > gpurun_out/ab2/ab.txt 2>&1; cat gpurun_out/ab2/ab.txt
for w in 1.45,0.95,0.6 1.3,1.0,0.7 1.6,0.9,0.5 1.2,1.0,0.8 1.7,0.8,0.5 1.5,1.0,0.5 1,1,1; do
  echo "W=$w"; PROT_RANK_W=$w timeout 300 python tools/ab_bench.py --rounds 2 --cases poly,cli,cli1024 main 2>&1 | tail -1
done > gpurun_out/ab2/w.txt 2>&1; cat gpurun_out/ab2/w.txt
