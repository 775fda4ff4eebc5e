mkdir -p gpurun_out/s22
timeout 600 python3 tools/_dbg16.py 16384 3600 2>&1 | grep -v amdgpu.ids | head -12
timeout 900 python3 bench.py --no-extra --no-cpu-baseline > gpurun_out/s22/bench.json 2> gpurun_out/s22/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/s22/bench.json").read().strip().splitlines()[-1])
sw = d["sweep"]
for k in ("harmonic", "noise_worst_case", "blksiz_16384", "blksiz_8192"):
    line = sw if k == "harmonic" else sw[k]
    p8 = line.get("predicted_8gpu", {})
    print(k, "ms", round(line.get("ms"), 3), "cold", round(line.get("cold_ms"), 3), "pred", round(p8.get("predicted_speedup"), 2), "ident", p8.get("merged_table_identical_single_pass"), [round(v, 3) for v in p8.get("per_shard_ms", [])])
PY
