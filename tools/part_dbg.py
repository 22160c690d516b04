import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A, checker as ck
n = int(float(sys.argv[1]))
d = A.Device(0)
K = 100
id3, id6, v1, v2, v3 = (d.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID3, ck.GEN_ID6, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
for name, keys, ops, vals in (("Q5", [id6], [ck.RED_SUM] * 3, [v1, v2, v3]), ("Q3", [id3], [ck.RED_SUM, ck.RED_AVG], [v1, v3]), ("Q7", [id3], [ck.RED_MAX, ck.RED_MIN], [v1, v2])):
    h = None
    best = 1e9; kb = 1e9
    for rep in range(3):
        d.sync(); d.timer_start()
        h = d.groupby_agg(keys, ops, vals, hint=n // K + 1024, handle=h)
        best = min(best, d.timer_stop_ms()); kb = min(kb, d.last_kernel_ms())
    print(name, "LF1000", os.environ.get("AQG_PART_LF1000"), "KB", os.environ.get("AQG_PART_LDSKB"), "call ms %.2f  agg kernel ms %.2f" % (best, kb), "groups", h.ngroups, flush=True)
    h.destroy()
