cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_pmc
mkdir -p $O
for b in 256 2900; do
for c in FETCH_SIZE WRITE_SIZE; do
  AQG_P1_BINS=$b timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/q5_${b}_$c -- python3 tools/q5_probe.py 1e9 q5 > $O/q5_${b}_$c.log 2>&1 || true
done
echo "== bins $b"; python3 tools/pmc_table.py $O/q5_${b}_FETCH_SIZE $O/q5_${b}_WRITE_SIZE | tee $O/q5_${b}_table.md
done
rm -rf $O/*_SIZE
