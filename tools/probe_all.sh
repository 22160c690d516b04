cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/perf_probe.py 1e9 all > gpurun_out/r2_probe_all.txt 2>&1; cat gpurun_out/r2_probe_all.txt | cut -c1-150
