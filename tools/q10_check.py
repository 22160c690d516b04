"""h2o Q10 at full size, checked against its own input: with seed 42 every one of the 1e9 (id1 .. id6) tuples is distinct, so the group
table must BE the input -- group g = row g: keys = the key columns, first rows = 0 .. n-1, counts = 1, sum(v3) = v3.  (The oracle
cannot hold 1e9 groups in a time a test may take; tests/test_gpu_configs.py checks the same query against it at 3e6 rows.)"""
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
ids = [d.gen_column(c, 42, 0, n, n, 100) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6)]
v3 = d.gen_column(ck.GEN_V3, 42, 0, n, n, 100)
gb = d.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=n)
G = gb.ngroups
print("groups", G, "of", n, "rows", flush=True)
assert G == n, "the check needs all-distinct tuples"
tmp = d.empty(n, np.int32)
for k in range(6):
    d._chk(d.lib.aqg_groupby_keys(gb.h, k, C.c_void_p(tmp.ptr)), "aqg_groupby_keys")
    ne = d.ewise(ck.OP_NE, tmp, ids[k], keep=True)
    bad = int(d.reduce(ck.RED_SUM, ne)); ne.free()
    print(f"key column {k}: {bad} groups differ from their row", flush=True)
    assert bad == 0
fr = gb.first_rows()
step = 1 << 26
for lo in range(0, n, step):
    hi = min(n, lo + step)
    assert np.array_equal(fr[lo:hi], np.arange(lo, hi, dtype=np.uint32)), lo
print("first rows are 0 .. n-1", flush=True)
del fr
cnt = gb.result(1, ck.RED_COUNT, ck.FLOAT)
assert int(cnt.min()) == 1 and int(cnt.max()) == 1
print("counts are all 1", flush=True)
del cnt
s = gb.result(0, ck.RED_SUM, ck.FLOAT)
h3 = v3.to_host()
for lo in range(0, n, step):
    hi = min(n, lo + step)
    assert np.array_equal(s[lo:hi], h3[lo:hi].astype(np.float64)), lo
print("sum(v3) equals v3 row by row", flush=True)
print("Q10 at", n, "rows: the group table equals the input")
