"""scratch: timings of the per-group scan family at size (HIP events around the whole C-ABI call), for rocprofv3 --kernel-trace --stats too"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "all"
d = A.Device(0)

def timeit(name, bpr, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        d.sync(); d.timer_start(); r = fn(); best = min(best, d.timer_stop_ms())
    gbs = bpr * n / best / 1e6
    print(f"{name:46s} {best:9.3f} ms  {n/best/1e6:8.2f} Grows/s  {gbs:7.1f} GB/s ({gbs/80:.1f}%)", flush=True)
    return r

price = d.gen_column(ck.GEN_PRICE, 42, 0, n, n, 100)
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
v2 = d.gen_column(ck.GEN_V2, 42, 0, n, n, 100)
out16 = d.empty(n, ck.I128)
def obuf(op, x):
    ot = d.lib.aqg_scan_out_dtype(op, x.tag)
    return A.DevBuf(d, out16.ptr, A.capi.TAG2NP[ot], n, owned=False)
for label, K in (("100 groups", None), ("1e5 groups", max(n // 100_000, 1)), ("n/100 groups", 100)):
    if which not in ("all", label.split()[0]):
        continue
    key = d.gen_column(ck.GEN_ID1, 42, 0, n, n, 100) if K is None else d.gen_column(ck.GEN_ID3, 42, 0, n, n, K)
    gb = timeit(f"[{label}] build", 12, lambda: d.groupby_build([key]), reps=2)
    print("   groups", gb.ngroups)
    xf = timeit(f"[{label}] flatten int32", 12, lambda: d.grouped_flatten(gb, price, keep=True))
    for name, w, bpr in (("avgw", 5, 12), ("sumw", 5, 20), ("minw", 10, 8), ("maxw", 1000, 8), ("sums", 0, 20), ("avgs", 0, 12), ("mins", 0, 8), ("deltas", 0, 8), ("ratiow", 1, 8)):
        op = ck.SCAN_NAMES[name]
        timeit(f"[{label}] scan_flat {name}({w})", bpr, lambda: d.grouped_scan(gb, op, xf, w, flat=True, keep=True, out=obuf(op, xf)))
    op = ck.SCAN_AVGW
    timeit(f"[{label}] q7 avgw(5) flatten+scan (16 B/row)", 16, lambda: d.grouped_scan(gb, op, price, 5, keep=True, out=obuf(op, price)))
    for name in ("sum", "max", "avg", "var", "first"):
        timeit(f"[{label}] reduce_flat {name}", 4, lambda: d.grouped_reduce_flat(gb, ck.RED_NAMES[name], xf))
    timeit(f"[{label}] grouped_corr(v1, v2) (12 B/row)", 12, lambda: d.grouped_corr(gb, v1, v2))
    timeit(f"[{label}] grouped_reduce sum(v1) (8 B/row)", 8, lambda: d.grouped_reduce(gb, ck.RED_SUM, v1))
    xf.free(); gb.destroy(); key.free()
d.close()
