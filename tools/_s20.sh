mkdir -p gpurun_out/s20
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s20/tests.txt 2>&1
tail -4 gpurun_out/s20/tests.txt
timeout 900 python3 bench.py --no-extra --no-cpu-baseline > gpurun_out/s20/bench.json 2> gpurun_out/s20/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s20/bench.json").read().strip().splitlines()[-1])
    print("headline", d["ms_per_step"], d["roofline"]["frac"])
    sw = d["sweep"]
    for k in ("harmonic", "noise_worst_case", "blksiz_16384", "blksiz_8192"):
        line = sw if k == "harmonic" else sw[k]
        p8 = line.get("predicted_8gpu", {})
        print(k, "ms", line.get("ms"), "cold", line.get("cold_ms"), "unpruned", line.get("unpruned_ms"), "pred", p8.get("predicted_speedup"), "ident", p8.get("merged_table_identical_single_pass"), line.get("table_identical_unpruned"), [round(v, 3) for v in p8.get("per_shard_ms", [])])
    print(sw["blksiz_16384"].get("oracle_check_10s"))
except Exception as ex:
    print("parse failed", ex)
    print(open("gpurun_out/s20/bench.err").read()[-3000:])
PY
