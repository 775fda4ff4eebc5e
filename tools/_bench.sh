mkdir -p gpurun_out/bench
timeout 1500 python bench.py > gpurun_out/bench/bench.json 2> gpurun_out/bench/bench.err; echo rc=$?; tail -3 gpurun_out/bench/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline')},indent=0))
for k in ('direct_form','cli_8192','also','fp16','lv2_stereo_batched','lv2_run_host','cli_file_host','sweep','cpu_baseline'):
    print(k, json.dumps(d.get(k))[:1500])
PY
