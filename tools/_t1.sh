mkdir -p gpurun_out/t1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/_abq.sh poly,cli,cli1024,poly_96k main nopipe
