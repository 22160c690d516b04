"""scratch: seam A at high cardinality -- build + grouped reduces through the C-ABI the header layer uses (HIP events, whole calls)"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
v3 = d.gen_column(ck.GEN_V3, 42, 0, n, n, 100)
for K in [int(float(a)) for a in (sys.argv[2:] or ["100", "1000", "10000"])]:
    key = d.gen_column(ck.GEN_ID3, 42, 0, n, n, K)
    for rep in range(2):
        d.sync(); d.timer_start()
        gb = d.groupby_build([key])
        tb = d.timer_stop_ms()
        if rep == 0: gb.destroy()
    out = d.empty(gb.ngroups, ck.I128)
    ts = []
    for x in (v1, v3):
        best = 1e9
        for rep in range(3):
            d.sync(); d.timer_start()
            d._chk(d.lib.aqg_grouped_reduce(d.ctx, gb.h, ck.RED_SUM, x.tag, C.c_void_p(x.ptr), C.c_void_p(out.ptr)), "aqg_grouped_reduce")
            best = min(best, d.timer_stop_ms())
        ts.append(best)
    print(f"{gb.ngroups:>9d} groups: build {tb:7.2f} ms (plan {gb.plan}), grouped sum(int32) {ts[0]:6.2f} ms, sum(float) {ts[1]:6.2f} ms", flush=True)
    gb.destroy(); key.free()
d.close()
