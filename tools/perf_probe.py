"""scratch: kernel timings for every config shape on the GPU box (HIP events around the whole call + dominant kernel)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "all"
d = A.Device(0)
K = 100
cols = {}
def col(c):
    if c not in cols:
        cols[c] = d.gen_column(c, 42, 0, n, n, K)
    return cols[c]

def timeit(name, bytes_per_row, fn, reps=4, kernel=True):
    best, kbest = 1e9, 1e9
    for _ in range(reps):
        d.sync(); d.timer_start(); r = fn(); ms = d.timer_stop_ms()
        best = min(best, ms)
        if kernel:
            try: kbest = min(kbest, d.last_kernel_ms())
            except Exception: kbest = float('nan')
    gbs = bytes_per_row * n / best / 1e6
    kg = bytes_per_row * n / kbest / 1e6 if kernel and kbest == kbest else float('nan')
    # the library times ONE kernel per call (its dominant one); for calls made of several kernels that kernel moves only part of the call's
    # bytes, so a rate is printed only where it can be one (a fraction above 100 % of the peak is an accounting artefact, not a measurement)
    if kernel and kbest == kbest and kbest < 1e8 and kg <= 8000.0:
        ktxt = f"kernel {kbest:8.3f} ms {kg:7.1f} GB/s ({kg/80:.1f}%)"
    elif kernel and kbest == kbest and kbest < 1e8:
        ktxt = f"kernel {kbest:8.3f} ms (one of several kernels of this call: no rate)"
    else:
        ktxt = "kernel        - (no single dominant launch timed)"
    print(f"{name:34s} {best:9.3f} ms  {n/best/1e6:8.1f} Grows/s  call {gbs:7.1f} GB/s ({gbs/80:.1f}%)  {ktxt}", flush=True)
    return r

if which in ("all", "gb"):
    id1, id2, id3, id6, v1, v2, v3 = (col(c) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID6, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
    h = {}
    def agg(key, keys, ops, vals, hint):
        def f():
            h[key] = d.groupby_agg(keys, ops, vals, hint=hint, handle=h.get(key))
            return h[key]
        return f
    timeit("Q1 sum(v1) by id1", 8, agg("q1", [id1], [ck.RED_SUM], [v1], 128))
    timeit("Q2 sum(v1) by id1,id2", 12, agg("q2", [id1, id2], [ck.RED_SUM], [v1], 16384))
    timeit("Q4 avg(v1,v2,v3) by id1", 16, agg("q4", [id1], [ck.RED_AVG] * 3, [v1, v2, v3], 128))
    g = timeit("Q3 sum(v1),avg(v3) by id3", 12, agg("q3", [id3], [ck.RED_SUM, ck.RED_AVG], [v1, v3], n // K + 1024), reps=2)
    print("   Q3 groups", g.ngroups)
    g = timeit("Q5 sum(v1,v2,v3) by id6", 16, agg("q5", [id6], [ck.RED_SUM] * 3, [v1, v2, v3], n // K + 1024), reps=2)
    print("   Q5 groups", g.ngroups)
    timeit("Q7 max(v1),min(v2) by id3", 12, agg("q7", [id3], [ck.RED_MAX, ck.RED_MIN], [v1, v2], n // K + 1024), reps=2)
    for k in list(h): h[k].destroy()
if which in ("all", "generic"):
    # shapes beyond Q1 / Q4 that still take the fast LDS kernel: MIN / MAX / VAR kinds, two key columns, mixed aggregates
    id1, id2, v1, v2, v3 = (col(c) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
    k7 = d.ewise(ck.OP_MOD, id2, np.int32(7), keep=True)
    hg = {}
    def aggg(key, keys, ops, vals):
        def f():
            hg[key] = d.groupby_agg(keys, ops, vals, hint=1024, handle=hg.get(key))
        return f
    timeit("max(v1),min(v2) by id1", 12, aggg("a", [id1], [ck.RED_MAX, ck.RED_MIN], [v1, v2]))
    timeit("var(v1) by id1", 8, aggg("b", [id1], [ck.RED_VAR], [v1]))
    timeit("min(v3) by id1", 8, aggg("c", [id1], [ck.RED_MIN], [v3]))
    timeit("sum(v1) by id1,id2%7 (700 groups)", 12, aggg("d", [id1, k7], [ck.RED_SUM], [v1]))
    timeit("sum(v1),max(v2),min(v3),avg(v1) by id1", 16, aggg("e", [id1], [ck.RED_SUM, ck.RED_MAX, ck.RED_MIN, ck.RED_AVG], [v1, v2, v3, v1]))
    # 8-byte value columns (doubles, int64): the same kernel with 8-byte value loads
    v3d = d.ewise(ck.OP_MUL, v3, np.float64(1.0), ot=ck.DOUBLE, keep=True)
    v1l = d.ewise(ck.OP_MUL, v1, np.int64(3_000_000_007), ot=ck.INT64, keep=True)
    assert v3d.dtype == np.float64 and v1l.dtype == np.int64
    timeit("sum(v3 double) by id1", 12, aggg("f", [id1], [ck.RED_SUM], [v3d]))
    timeit("avg(v3 double) by id1", 12, aggg("g", [id1], [ck.RED_AVG], [v3d]))
    timeit("min(v3 double) by id1", 12, aggg("h", [id1], [ck.RED_MIN], [v3d]))
    timeit("sum(v1 int64) by id1", 12, aggg("i", [id1], [ck.RED_SUM], [v1l]))
    timeit("sum(v3 double) by id1,id2%7", 16, aggg("j", [id1, k7], [ck.RED_SUM], [v3d]))
    timeit("sum(v1),avg(v3 double) by id1", 16, aggg("n", [id1], [ck.RED_SUM, ck.RED_AVG], [v1, v3d]))
    m8 = d.ewise(ck.OP_GT, v1, np.int32(2), keep=True)                       # a bool mask (1 byte per row) as the value column
    timeit("sum(v1 > 2) by id1 (1-byte mask)", 5, aggg("o", [id1], [ck.RED_SUM], [m8]))
    m8.free()
    id1l = d.ewise(ck.OP_MUL, id1, np.int64(1_000_000_007), ot=ck.INT64, keep=True)
    timeit("sum(v1) by id1 (one int64 key)", 12, aggg("l", [id1l], [ck.RED_SUM], [v1]))
    timeit("sum(v3 double) by id1 (int64 key)", 16, aggg("m", [id1l], [ck.RED_SUM], [v3d]))
    id1l.free()
    v2d = d.ewise(ck.OP_MUL, v2, np.float64(1.0), ot=ck.DOUBLE, keep=True)
    timeit("avg(v3d),avg(v2d),sum(v1 int64) by id1", 28, aggg("k", [id1], [ck.RED_AVG, ck.RED_AVG, ck.RED_SUM], [v3d, v2d, v1l]))
    v2d.free()
    for k in list(hg): hg[k].destroy()
    k7.free(); v3d.free(); v1l.free()
if which == "q10":
    ids = [col(c) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6)]
    v3 = col(ck.GEN_V3)
    hq = {}
    def q10():
        hq["h"] = d.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=n, handle=hq.get("h"))
        return hq["h"]
    g = timeit("Q10 sum(v3),count by id1..id6", 28, q10, reps=2)
    print("   Q10 groups", g.ngroups, flush=True)
if which in ("all", "build"):
    id1 = col(ck.GEN_ID1)
    hb = {}
    def build():
        if "b" in hb: hb["b"].destroy()
        hb["b"] = d.groupby_build([id1], hint=128); return hb["b"]
    g = timeit("build reversemap by id1", 12, build, reps=3)
    off = d.empty(g.ngroups + 1, np.uint32); rows = d.empty(n, np.uint32)
    import ctypes as C
    def pp():
        d._chk(d.lib.aqg_groupby_postproc(g.h, C.c_void_p(off.ptr), C.c_void_p(rows.ptr)), "pp")
    timeit("ht_postproc (1 radix pass)", 12, pp, reps=3)
    v1 = col(ck.GEN_V1)
    timeit("grouped_reduce sum(v1[vecs])", 8, lambda: d.grouped_reduce(g, ck.RED_SUM, v1), reps=3)
    g.destroy()
    if which == "build":
        id3 = col(ck.GEN_ID3)
        hb3 = {}
        def build3():
            if "b" in hb3: hb3["b"].destroy()
            hb3["b"] = d.groupby_build([id3], hint=n // K + 1024); return hb3["b"]
        g3 = timeit("build reversemap by id3 (1e7 groups)", 12, build3, reps=2)
        timeit("grouped_reduce sum(v1[vecs]) 1e7 groups", 8, lambda: d.grouped_reduce(g3, ck.RED_SUM, v1), reps=2)
        off3 = d.empty(g3.ngroups + 1, np.uint32); rows3 = d.empty(n, np.uint32)
        timeit("ht_postproc 1e7 groups (3 radix passes)", 36, lambda: d._chk(d.lib.aqg_groupby_postproc(g3.h, C.c_void_p(off3.ptr), C.c_void_p(rows3.ptr)), "pp"), reps=2)
        g3.destroy()
if which in ("all", "scan"):
    price, v1 = col(ck.GEN_PRICE), col(ck.GEN_V1)
    outs = {}
    big = d.empty(n, ck.I128)          # one 16-byte-per-row output buffer reused by every scan (allocation is not what is timed)
    def sc(op, x, w, key):
        ot = d.lib.aqg_scan_out_dtype(op, x.tag)
        o = A.DevBuf(d, big.ptr, A.capi.TAG2NP[ot], n, owned=False)
        def f():
            d.scan(op, x, w, keep=True, out=o)
        return f
    for name, bpr, w in (("mins", 8, 0), ("maxs", 8, 0), ("sums", 20, 0), ("avgs", 12, 0), ("deltas", 8, 0), ("prev", 8, 0),
                         ("ratiow", 8, 1), ("avgw", 12, 5), ("avgw", 12, 100), ("sumw", 20, 5), ("minw", 8, 3), ("minw", 8, 10), ("minw", 8, 100), ("maxw", 8, 1000)):
        timeit(f"{name}({w}) price", bpr, sc(ck.SCAN_NAMES[name], price, w, "o"), reps=3)
    for k in list(outs): outs[k].free()
if which in ("all", "ew"):
    price, v1, v3 = col(ck.GEN_PRICE), col(ck.GEN_V1), col(ck.GEN_V3)
    outs = {}
    big = d.empty(n, ck.I128)
    def ew(op, l, r, key="o", ot=None):
        lt = l.tag; rt = r.tag if isinstance(r, A.DevBuf) else ck.tag_of(np.atleast_1d(r))
        o_t = ot if ot is not None else d.lib.aqg_ewise_out_dtype(op, lt, rt)
        o = A.DevBuf(d, big.ptr, A.capi.TAG2NP[o_t], n, owned=False)
        def f():
            d.ewise(op, l, r, ot=ot, keep=True, out=o)
            d.sync()
        return f
    timeit("price + v1 (int32)", 12, ew(ck.OP_ADD, price, v1), kernel=False)
    timeit("price * v1 (->int128)", 24, ew(ck.OP_MUL, price, v1), kernel=False)
    timeit("price > 275 (->bool)", 5, ew(ck.OP_GT, price, np.int32(275)), kernel=False)
    timeit("v3 * 2.0f (->double)", 12, ew(ck.OP_MUL, v3, np.float32(2.0)), kernel=False)
    timeit("sum(price)", 4, lambda: d.reduce(ck.RED_SUM, price), kernel=False)
    timeit("max(price)", 4, lambda: d.reduce(ck.RED_MAX, price), kernel=False)
    timeit("sum(v3) float", 4, lambda: d.reduce(ck.RED_SUM, v3), kernel=False)
    for k in list(outs): outs[k].free()
if which in ("all", "join"):
    # config 4: fact JOIN small(id4, w) ON id4, then sum(v1 * w) by id1 -- composed from the C-ABI pieces
    import ctypes as C
    id1, id4, v1 = col(ck.GEN_ID1), col(ck.GEN_ID4), col(ck.GEN_V1)
    rng = np.random.default_rng(4)
    dim_key = d.to_device(rng.permutation(np.arange(1, K + 1, dtype=np.int32)))
    dim_w = d.to_device(rng.integers(1, 50, K).astype(np.int32))
    look = d.empty(n, np.uint32); wrow = d.empty(n, np.int32); prod = d.empty(n, np.int64)
    def lookup():
        d._chk(d.lib.aqg_join_lookup(d.ctx, ck.INT32, C.c_void_p(dim_key.ptr), C.c_uint32(K), C.c_void_p(id4.ptr), C.c_uint32(n), C.c_void_p(look.ptr)), "lookup")
    def gather():
        d._chk(d.lib.aqg_gather(d.ctx, ck.INT32, C.c_void_p(dim_w.ptr), C.c_void_p(look.ptr), C.c_uint32(n), C.c_void_p(wrow.ptr)), "gather")
    timeit("join_lookup(dim.id4, fact.id4)", 8, lookup, kernel=False)
    timeit("gather w[lookup]", 12, gather, kernel=False)
    timeit("v1 * w -> int64", 16, lambda: d.ewise(ck.OP_MUL, v1, wrow, ot=ck.INT64, keep=True, out=prod), kernel=False)
    hj = {}
    def agg():
        hj["h"] = d.groupby_agg([id1], [ck.RED_SUM], [prod], hint=128, handle=hj.get("h"))
    timeit("sum(prod int64) by id1", 12, agg)
    def whole():
        lookup(); gather(); d.ewise(ck.OP_MUL, v1, wrow, ot=ck.INT64, keep=True, out=prod); agg()
    timeit("config 4 composed (12 B/row alg.)", 12, whole, kernel=False)
    hf = {}
    def fused():
        hf["h"] = d.join_groupby_sum(dim_key, dim_w, id4, id1, v1, hint=128, handle=hf.get("h"))
    timeit("config 4 fused aqg_join_groupby_sum", 12, fused)
    assert ck.i128_to_int(hf["h"].result(0, ck.RED_SUM, ck.INT64)) == ck.i128_to_int(hj["h"].result(0, ck.RED_SUM, ck.INT64))
    big = d.gen_column(ck.GEN_ID3, 42, 0, n, n, K)       # random row ids in [1, n/K]
    timeit("gather v1[random idx]", 12, lambda: d._chk(d.lib.aqg_gather(d.ctx, ck.INT32, C.c_void_p(v1.ptr), C.c_void_p(big.ptr), C.c_uint32(n), C.c_void_p(wrow.ptr)), "g"), kernel=False)
    mask = d.empty(n, np.uint8)
    d.ewise(ck.OP_GT, v1, np.int32(2), keep=True, out=A.DevBuf(d, mask.ptr, np.bool_, n, owned=False))
    mh = C.c_uint32()
    timeit("compact v1[v1 > 2] (60 % kept)", 5 + 2.4, lambda: d._chk(d.lib.aqg_compact(d.ctx, ck.INT32, C.c_void_p(v1.ptr), C.c_void_p(mask.ptr), C.c_uint32(n), C.c_void_p(wrow.ptr), C.byref(mh)), "c"), kernel=False)
