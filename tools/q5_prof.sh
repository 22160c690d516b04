#!/bin/bash
# run on the GPU box: per-kernel split of h2o Q5 / Q3 / Q7 at 1e9 rows (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-q5}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/q5_prof -o p -- python3 tools/q5_probe.py 1e9 $W > gpurun_out/q5_prof.log 2>&1 || true
grep "rep\|rror" gpurun_out/q5_prof.log | head; python3 tools/kstats.py gpurun_out/q5_prof/p_results.db 24 | cut -c1-70,100-
