"""scratch: shapes that take the generic small-LDS kernel (agg_kernel<LDS,...,256>), 1e9 rows"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
d = A.Device(0)
K = 100
c = {x: d.gen_column(x, 42, 0, n, n, K) for x in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3)}
k12 = d.ewise(ck.OP_MOD, c[ck.GEN_ID2], np.int32(7), keep=True)
cases = [("max(v1),min(v2) by id1", [c[ck.GEN_ID1]], [ck.RED_MAX, ck.RED_MIN], [c[ck.GEN_V1], c[ck.GEN_V2]], 12),
         ("var(v1) by id1", [c[ck.GEN_ID1]], [ck.RED_VAR], [c[ck.GEN_V1]], 8),
         ("min(v3) by id1", [c[ck.GEN_ID1]], [ck.RED_MIN], [c[ck.GEN_V3]], 8),
         ("sum(v1) by id1, id2%7 (700 groups)", [c[ck.GEN_ID1], k12], [ck.RED_SUM], [c[ck.GEN_V1]], 12),
         ("sum(v1),max(v2),min(v3),avg(v1) by id1", [c[ck.GEN_ID1]], [ck.RED_SUM, ck.RED_MAX, ck.RED_MIN, ck.RED_AVG], [c[ck.GEN_V1], c[ck.GEN_V2], c[ck.GEN_V3], c[ck.GEN_V1]], 16)]
for name, keys, ops, vals, bpr in cases:
    h = None; best = 1e9; kb = 1e9
    for rep in range(3):
        d.sync(); d.timer_start()
        h = d.groupby_agg(keys, ops, vals, hint=1024, handle=h)
        best = min(best, d.timer_stop_ms()); kb = min(kb, d.last_kernel_ms())
    print(f"{name:42s} groups={h.ngroups:5d} call {best:7.3f} ms kernel {kb:7.3f} ms = {bpr * n / kb / 1e6 / 80:5.1f} % of 8 TB/s", flush=True)
