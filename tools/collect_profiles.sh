#!/bin/bash
# run on the GPU box (through gpurun): bench line + rocprofv3 kernel statistics for the bench step and for every probe shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r2}
O=gpurun_out/prof_$R
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench done"; cut -c1-300 $O/bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-secondary > $O/bench_prof.log 2>&1
echo "bench profile done"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe -o probe -- python3 tools/perf_probe.py 1e9 all > $O/probe.log 2>&1
echo "probe profile done"
grep -v "simple_timer\|^W2026\|^E2026" $O/probe.log > $O/probe_clean.log
find $O -name "*kernel_stats.csv"
