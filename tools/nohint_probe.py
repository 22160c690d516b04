"""scratch: group-by calls WITHOUT a group-count hint (what the header layer passes) at 1e9 rows"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
d = A.Device(0)
K = 100
c = {x: d.gen_column(x, 42, 0, n, n, K) for x in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_V1)}
for name, keys in (("by id1", [c[ck.GEN_ID1]]), ("by id1,id2", [c[ck.GEN_ID1], c[ck.GEN_ID2]]), ("by id3", [c[ck.GEN_ID3]])):
    for rep in range(2):
        d.sync(); d.timer_start()
        h = d.groupby_agg(keys, [ck.RED_SUM], [c[ck.GEN_V1]], hint=0)      # a fresh handle every time: no remembered hint
        ms = d.timer_stop_ms()
        print(f"{name:12s} hint=0 groups={h.ngroups:9d} call {ms:9.3f} ms", flush=True)
        h.destroy()
    d.sync(); d.timer_start()
    g = d.groupby_build(keys)
    print(f"{name:12s} build hint=0 groups={g.ngroups:9d} call {d.timer_stop_ms():9.3f} ms", flush=True)
    g.destroy()
for rep in range(2):
    d.sync(); d.timer_start()
    h = d.groupby_agg([c[ck.GEN_ID3]], [ck.RED_SUM], [c[ck.GEN_V1]], hint=n // K + 1024)
    print(f"by id3 explicit hint, fresh handle: call {d.timer_stop_ms():9.3f} ms", flush=True)
    h.destroy()
h = None
for rep in range(3):
    d.sync(); d.timer_start()
    h = d.groupby_agg([c[ck.GEN_ID3]], [ck.RED_SUM], [c[ck.GEN_V1]], hint=0, handle=h)
    print(f"by id3 hint=0, reused handle: call {d.timer_stop_ms():9.3f} ms", flush=True)
