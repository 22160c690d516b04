"""scratch: the few-group shapes of VERDICT item 6 (var, sum int64, max+min, Q1, Q4, Q2 dense), `reps` dispatches each, in a fixed order"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = A.Device(0)
col = lambda c: d.gen_column(c, 42, 0, n, n, 100)
id1, id2, v1, v2, v3 = (col(c) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
v1l = d.ewise(ck.OP_MUL, v1, np.int64(3_000_000_007), ot=ck.INT64, keep=True)
shapes = [("Q1 sum(v1)", [id1], [ck.RED_SUM], [v1], 8), ("Q4 avg x3", [id1], [ck.RED_AVG] * 3, [v1, v2, v3], 16), ("var(v1)", [id1], [ck.RED_VAR], [v1], 8),
          ("max(v1),min(v2)", [id1], [ck.RED_MAX, ck.RED_MIN], [v1, v2], 12), ("sum(v1 int64)", [id1], [ck.RED_SUM], [v1l], 12),
          ("Q2 sum(v1) by id1,id2", [id1, id2], [ck.RED_SUM], [v1], 12)]
for name, keys, ops, vals, bpr in shapes:
    h = None
    best = 1e9
    for _ in range(reps):
        d.sync(); d.timer_start(); h = d.groupby_agg(keys, ops, vals, hint=16384 if len(keys) > 1 else 1024, handle=h); ms = d.timer_stop_ms()
        best = min(best, d.last_kernel_ms())
    print(f"{name:26s} kernel {best:7.3f} ms  {bpr * n / best / 1e6 / 80:5.1f} % of 8 TB/s", flush=True)
    h.destroy()
