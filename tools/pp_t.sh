cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_build -o p -- python3 tools/perf_probe.py 1e9 build > gpurun_out/r2_build.log 2>&1 || true
grep -v "^W2026\|^E2026" gpurun_out/r2_build.log | cut -c1-150 | head; python3 tools/kstats.py gpurun_out/r2_build/p_results.db 22 | cut -c1-70,100-
