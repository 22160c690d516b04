cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AMD_LOG_LEVEL=1 timeout -k 10 600 python -m pytest tests/test_gpu_scan_join.py tests/test_gpu_configs.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r2_t7.log 2>&1; tail -4 gpurun_out/r2_t7.log; grep -v "^  File\|^$" gpurun_out/r2_t7.log | head -30
