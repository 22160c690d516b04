set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_r1b
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/bench_$c -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_$c.log 2>&1
  echo "bench $c done"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/generic_$c -- python tools/perf_probe.py 1e9 generic > $O/generic_$c.log 2>&1
  echo "generic $c done"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/scan_$c -- python tools/perf_probe.py 1e9 scan > $O/scan_$c.log 2>&1
  echo "scan $c done"
done
python tools/pmc_table.py $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE > $O/bench_table.md
python tools/pmc_table.py $O/generic_FETCH_SIZE $O/generic_WRITE_SIZE > $O/generic_table.md
python tools/pmc_table.py $O/scan_FETCH_SIZE $O/scan_WRITE_SIZE > $O/scan_table.md
rm -rf $O/*_SIZE
cat $O/bench_table.md
