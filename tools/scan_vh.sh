cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_scan_fuzz.py tests/test_gpu_scan_join.py -x -q -m gpu > gpurun_out/r2_t6.log 2>&1; tail -5 gpurun_out/r2_t6.log
timeout -k 10 200 python3 tools/perf_probe.py 1e9 scan 2>&1 | grep -E "minw|maxw"
for c in 5 7 9 11; do echo "C=$c"; AQG_VANHERK_C=$c timeout -k 10 200 python3 tools/perf_probe.py 1e9 scan 2>&1 | grep -E "maxw"; done
