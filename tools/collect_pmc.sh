#!/bin/bash
# run on the GPU box (through gpurun): HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) for the bench step
# and the high-cardinality group-by plans; tables under gpurun_out/pmc_<round>/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r2}
O=gpurun_out/pmc_$R
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/bench_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary > $O/bench_$c.log 2>&1
  echo "bench $c done"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/hicard_$c -- python3 tools/q5_probe.py 1e9 q5,q3,q7 > $O/hicard_$c.log 2>&1
  echo "hicard $c done"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/q10_$c -- python3 tools/perf_probe.py 1e9 q10 > $O/q10_$c.log 2>&1
  echo "q10 $c done"
done
python3 tools/pmc_table.py $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE > $O/bench_table.md
python3 tools/pmc_table.py $O/hicard_FETCH_SIZE $O/hicard_WRITE_SIZE > $O/hicard_table.md
python3 tools/pmc_table.py $O/q10_FETCH_SIZE $O/q10_WRITE_SIZE > $O/q10_table.md
rm -rf $O/*_SIZE
cat $O/bench_table.md $O/hicard_table.md
