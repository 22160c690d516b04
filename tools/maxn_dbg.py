import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
d = A.Device(0)
o = ck.load_oracle()
K = 100
for n in (2**32 - 2**20, 2**32 - 2**20 - 1, 2**32 - 2**20 - 7, 2**31 + 5):
    id1 = d.gen_column(ck.GEN_ID1, 42, 0, n, n, K)
    v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, K)
    total = int(d.reduce(ck.RED_SUM, v1))
    gb = d.groupby_agg([id1], [ck.RED_SUM, ck.RED_COUNT], [v1, v1], hint=128)
    gsum = sum(ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32)))
    gcnt = int(gb.counts().astype(np.uint64).sum())
    sums = d.scan(ck.SCAN_SUMS, v1, keep=True)
    last = ck.i128_to_int(A.DevBuf(d, sums.ptr + (n - 1) * 16, ck.I128, 1, owned=False).to_host())[0]
    tail = A.DevBuf(d, v1.ptr + (n - 16) * 4, np.int32, 16, owned=False).to_host()
    want_tail = o.gen_column(ck.GEN_V1, 42, n - 16, 16, n, K)
    print(n, "reduce", total, "groupby", gsum, "diff", gsum - total, "count diff", gcnt - n, "scan last diff", last - total, "tail ok", np.array_equal(tail, want_tail), tail[-8:], flush=True)
    gb.destroy(); sums.free(); id1.free(); v1.free()
