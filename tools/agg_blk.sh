cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_groupby_paths.py tests/test_gpu_groupby_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r2_t5.log 2>&1; tail -4 gpurun_out/r2_t5.log
timeout -k 10 200 python3 tools/agg_shapes.py 1e9 3
echo "== generic dense"; AQG_DENSE_GENERIC=1 timeout -k 10 200 python3 tools/agg_shapes.py 1e9 3 | grep Q2
