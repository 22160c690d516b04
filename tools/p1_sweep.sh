#!/bin/bash
# run on the GPU box: the one-level partition scatter with the bin count forced (profiles/r2_partition_bins_pmc.md)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for bins in 64 256 1024 2900; do
  AQG_P1_BINS=$bins AQG_P1_MAX=4096 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/p1_bins_$bins -o p -- python3 tools/q5_probe.py 1e9 q5 > gpurun_out/p1_bins_$bins.log 2>&1 || true
  echo "== $bins bins"; python3 tools/kstats.py gpurun_out/p1_bins_$bins/p_results.db 6 | cut -c1-60,100-
done
