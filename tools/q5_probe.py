"""scratch: h2o Q5 / Q3 / Q7 (1e7 groups) at n rows, whole-call time (run under rocprofv3 for the per-kernel split)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["q5"]
d = A.Device(0)
K = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100                  # rows per group
col = lambda c: d.gen_column(c, 42, 0, n, n, K)
id3, id6, v1, v2, v3 = (col(c) for c in (ck.GEN_ID3, ck.GEN_ID6, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
print('ptrs', ' '.join(f'{c.ptr:#x}' for c in (id3, id6, v1, v2, v3)), flush=True)
shapes = {"q3": ([id3], [ck.RED_SUM, ck.RED_AVG], [v1, v3], 12), "q5": ([id6], [ck.RED_SUM] * 3, [v1, v2, v3], 16), "q7": ([id3], [ck.RED_MAX, ck.RED_MIN], [v1, v2], 12)}
for w in which:
    keys, ops, vals, bpr = shapes[w]
    h = None
    for rep in range(3):
        d.sync(); d.timer_start()
        try:
            h = d.groupby_agg(keys, ops, vals, hint=n // K + 1024, handle=h)
            ms = d.timer_stop_ms()
            print(f"{w} n={n:.0e} rep{rep}: {ms:8.3f} ms  plan {h.plan}  groups {h.ngroups}  {bpr * n / ms / 1e6:7.1f} GB/s ({bpr * n / ms / 8e7:.1f}%)", flush=True)
        except Exception as e:
            print(w, "failed:", e, flush=True); break
