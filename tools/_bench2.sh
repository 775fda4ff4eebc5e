mkdir -p gpurun_out/bench
timeout 1500 python bench.py --no-extra --no-cpu-baseline > gpurun_out/bench/bench2.json 2> gpurun_out/bench/bench2.err; echo rc=$?; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench/bench2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d['sweep']['host_inclusive']))
PY
