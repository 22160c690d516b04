#!/bin/bash
# run on the GPU box: the partition / ordering paths forced onto the group-by tests, then the kernel statistics of h2o Q10 at 1e9 rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AQG_SORTED_TAIL_MIN=1 timeout -k 10 400 python -m pytest tests/test_gpu_groupby_paths.py tests/test_gpu_groupby_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/q10_tests.log 2>&1; tail -4 gpurun_out/q10_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/q10_prof -o p -- python3 tools/perf_probe.py 1e9 q10 > gpurun_out/q10_prof.log 2>&1 || true
grep "Q10\|Error\|error" gpurun_out/q10_prof.log | head; python3 tools/kstats.py gpurun_out/q10_prof/p_results.db 14 | cut -c1-60,100-
