"""scratch: aqg_grouped_reduce at high cardinality (for rocprofv3 --kernel-trace --stats)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100
d = A.Device(0)
key = d.gen_column(ck.GEN_ID3, 42, 0, n, n, K)
v1 = d.gen_column(ck.GEN_V3 if len(sys.argv) > 3 and sys.argv[3] == "v3" else ck.GEN_V1, 42, 0, n, n, 100)    # v3: a float column (nothing packs)
gb = d.groupby_build([key])
print("groups", gb.ngroups, flush=True)
out = d.empty(gb.ngroups, ck.I128)
import ctypes as C
for rep in range(3):
    d.sync(); d.timer_start()
    d._chk(d.lib.aqg_grouped_reduce(d.ctx, gb.h, ck.RED_SUM, v1.tag, C.c_void_p(v1.ptr), C.c_void_p(out.ptr)), "aqg_grouped_reduce")
    print("grouped_reduce sum(v1): %.3f ms, plan %d" % (d.timer_stop_ms(), gb.plan), flush=True)
d.close()
