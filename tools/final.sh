cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2_full.log 2>&1; tail -4 gpurun_out/r2_full.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-400 gpurun_out/bench_final.json
