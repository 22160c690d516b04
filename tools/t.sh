cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AQG_FUZZ_BASE=5000 AQG_FUZZ_SEEDS=700 timeout -k 10 1000 python -m pytest tests/test_gpu_groupby_fuzz.py tests/test_gpu_scan_fuzz.py tests/test_gpu_misc_fuzz.py -x -q -m gpu > gpurun_out/fuzz_more2.log 2>&1; tail -4 gpurun_out/fuzz_more2.log
