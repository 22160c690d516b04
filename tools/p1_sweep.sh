cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_q10 -o p -- python3 tools/perf_probe.py 1e9 q10 > gpurun_out/r2_q10.log 2>&1 || true
grep "Q10\|Error\|error" gpurun_out/r2_q10.log | head; python3 tools/kstats.py gpurun_out/r2_q10/p_results.db 14 | cut -c1-60,100-
