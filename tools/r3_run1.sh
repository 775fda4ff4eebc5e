# round 3, GPU session 1: geometry probe, all GPU tests, default bench line
mkdir -p gpurun_out/r3a
timeout 300 tools/ubench/copy_geom > gpurun_out/r3a/copy_geom.txt 2>&1; tail -3 gpurun_out/r3a/copy_geom.txt
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3a/pytest.txt 2>&1; tail -3 gpurun_out/r3a/pytest.txt
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo rc=$?; tail -2 gpurun_out/r3a/bench.err
