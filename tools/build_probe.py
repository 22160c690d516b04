"""scratch: aqg_groupby_build at high cardinality (for rocprofv3 --kernel-trace --stats)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100
d = A.Device(0)
key = d.gen_column(ck.GEN_ID3, 42, 0, n, n, K)
for rep in range(3):
    d.sync(); d.timer_start()
    gb = d.groupby_build([key])
    print("build: %.3f ms, groups %d, plan %d" % (d.timer_stop_ms(), gb.ngroups, gb.plan), flush=True)
    gb.destroy()
d.close()
