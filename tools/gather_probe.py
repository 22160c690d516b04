"""scratch: 1e9 random 4-byte gathers out of a table of T entries (is a key -> id lookup table in the Infinity Cache cheaper than routing ids back by row?)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
for K in (100, 10, 1000):
    T = n // K + 1
    idx = d.gen_column(ck.GEN_ID6, 42, 0, n, n, K)          # values 1 .. n / K
    table = d.gen_column(ck.GEN_V1, 7, 0, T + 1, T + 1, 100)
    out = d.empty(n, np.int32)
    best = 1e9
    for _ in range(3):
        d.sync(); d.timer_start()
        d._chk(d.lib.aqg_gather(d.ctx, table.tag, A.capi.C.c_void_p(table.ptr), A.capi.C.c_void_p(idx.ptr), n, A.capi.C.c_void_p(out.ptr)), "gather")
        best = min(best, d.timer_stop_ms())
    print(f"table of {T} int32 ({T * 4 / 1e6:.0f} MB): {best:.3f} ms for {n:.0e} gathers", flush=True)
    idx.free(); table.free(); out.free()
