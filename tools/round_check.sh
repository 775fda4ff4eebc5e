# the round's evidence on one GPU box: all -m gpu tests, smoke(), the default bench line, the rocprofv3 passes (tools/profile_round.sh r06);
# summaries land in gpurun_out/profiles_r06/ (copy into profiles/), the bench line in gpurun_out/bench/bench.json (-> profiles/rNN_bench_line.json; the full record bench/bench_detail.json -> profiles/rNN_bench_detail.json)
mkdir -p gpurun_out/chk
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/chk/pytest.txt 2>&1; tail -3 gpurun_out/chk/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/chk/smoke.txt 2>&1; tail -2 gpurun_out/chk/smoke.txt
mkdir -p gpurun_out/bench
timeout 1500 python bench.py > gpurun_out/bench/bench.json 2> gpurun_out/bench/bench.err; echo rc=$?; tail -3 gpurun_out/bench/bench.err
cp gpurun_out/bench_detail.json gpurun_out/bench/bench_detail.json   # the rocprofv3 bench runs below write their own
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 bash tools/profile_round.sh r06 > gpurun_out/chk/profile.txt 2>&1; tail -40 gpurun_out/chk/profile.txt
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_* profiles/traffic_latest.json profiles/pmc_latest.json profiles/sweep_traffic_latest.json gpurun_out/profiles_r06/ 2>/dev/null
rm -rf gpurun_out/prof_r06/stats gpurun_out/prof_r06/stats_full gpurun_out/prof_r06/fetch gpurun_out/prof_r06/write gpurun_out/prof_r06/sq1 gpurun_out/prof_r06/sq2 gpurun_out/prof_r06/sw1 gpurun_out/prof_r06/sw2 gpurun_out/prof_r06/swt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline')},indent=0))
for k in ('direct_form','cli_8192','also','fp16','lv2_stereo_batched','lv2_run_host','cli_file_host','sweep','cpu_baseline'):
    print(k, json.dumps(d.get(k))[:1500])
PY
