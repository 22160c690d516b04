cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_groupby_paths.py tests/test_gpu_groupby_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r2_t2.log 2>&1; tail -3 gpurun_out/r2_t2.log
AQG_P1_MAX=0 timeout -k 10 400 python -m pytest tests/test_gpu_groupby_paths.py tests/test_gpu_groupby_fuzz.py -x -q -m gpu > gpurun_out/r2_t3.log 2>&1; tail -3 gpurun_out/r2_t3.log
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_swC -o p -- python3 tools/q5_probe.py 1e9 q5,q3,q7 > gpurun_out/r2_swC.log 2>&1 || true
grep "^q" gpurun_out/r2_swC.log; python3 tools/kstats.py gpurun_out/r2_swC/p_results.db 12 | cut -c1-60,100-
