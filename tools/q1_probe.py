"""scratch: time the fused Q1 kernel on the GPU box (not part of the test suite)"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
d = A.Device(0)
id1 = d.gen_column(ck.GEN_ID1, 42, 0, n, n, 100)
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
d.sync()
gb = None
for it in range(8):
    d.timer_start()
    gb = d.groupby_agg([id1], [ck.RED_SUM], [v1], hint=128, handle=gb)
    ms = d.timer_stop_ms()
    print(f"iter {it}: {ms:.3f} ms  {n/ms/1e6:.1f} Grows/s  {8*n/ms/1e9:.3f} TB/s  G={gb.ngroups}")
res = gb.result(0, ck.RED_SUM, ck.INT32)
print("sum of sums", sum(ck.i128_to_int(res)))
for it in range(3):
    d.timer_start()
    s = d.reduce(ck.RED_SUM, v1)
    ms = d.timer_stop_ms()
    print(f"reduce sum: {ms:.3f} ms {4*n/ms/1e9:.3f} TB/s  s={s}")
