cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_gb -o p -- python3 tools/perf_probe.py 1e9 gb > gpurun_out/r2_gb.log 2>&1 || true
cat gpurun_out/r2_gb.log | grep -v "^$" | head -20; python3 tools/kstats.py gpurun_out/r2_gb/p_results.db 24 | cut -c1-70,100-
