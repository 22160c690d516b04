#!/bin/bash
# run on the GPU box: longer seeded sweeps of the fuzz suites, then the group-by suites with the ordering tail forced onto every partition plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AQG_FUZZ_SEEDS=${1:-400} timeout -k 10 900 python -m pytest tests/test_gpu_groupby_fuzz.py tests/test_gpu_scan_fuzz.py tests/test_gpu_misc_fuzz.py -x -q -m gpu > gpurun_out/fuzz_more.log 2>&1; tail -5 gpurun_out/fuzz_more.log
AQG_SORTED_TAIL_MIN=1 timeout -k 10 600 python -m pytest tests/test_gpu_groupby_paths.py tests/test_gpu_groupby_fuzz.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/fuzz_sorted.log 2>&1; tail -4 gpurun_out/fuzz_sorted.log
