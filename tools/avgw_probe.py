"""scratch: why avgw(5) is 2.33 ms in bench.py's secondary and 2.07 ms in perf_probe.py: buffer placement"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = 1_000_000_000
d = A.Device(0)
def t(label, price, out):
    best = 1e9
    for _ in range(5):
        d.sync(); d.timer_start(); d.scan(ck.SCAN_AVGW, price, 5, keep=True, out=out); best = min(best, d.timer_stop_ms())
    print(f"{label:60s} {best:.3f} ms  kernel {d.last_kernel_ms():.3f}  price@{price.ptr:#x} out@{out.ptr:#x}", flush=True)
price = d.gen_column(ck.GEN_PRICE, 42, 0, n, n, 100)
out = d.empty(n, np.float64)
t("fresh context: exact-size double buffer", price, out)
big = d.empty(n, ck.I128)
t("double result at the front of a 16 GB buffer", price, A.DevBuf(d, big.ptr, np.float64, n, owned=False))
t("... at its back half", price, A.DevBuf(d, big.ptr + 8 * n, np.float64, n, owned=False))
out.free(); big.free(); price.free()
# what bench.py does before: Q5 / Q2 / Q10 with their workspaces (100+ GB arena), columns freed and re-made
ids = [d.gen_column(c, 42, 0, n, n, 100) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6)]
v3 = d.gen_column(ck.GEN_V3, 42, 0, n, n, 100)
g = d.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=n)
g.destroy()
for c in ids: c.free()
v3.free()
price = d.gen_column(ck.GEN_PRICE, 42, 0, n, n, 100)
big = d.empty(n, ck.I128)
t("after Q10 (130 GB arena alive): front of a 16 GB buffer", price, A.DevBuf(d, big.ptr, np.float64, n, owned=False))
out = d.empty(n, np.float64)
t("after Q10: exact-size double buffer", price, out)
d.close()
