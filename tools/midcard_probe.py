"""scratch: fast LDS plan at a few thousand groups (sparse and dense key values), 1e9 rows"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
d = A.Device(0)
id3 = d.gen_column(ck.GEN_ID3, 42, 0, n, n, 100)       # U{1..n/100}
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
for G, mult, hint in [(G, m, G) for G in (100, 1000) for m in (1, 2, 3, 5, 7, 10, 16, 100, 1000, 1024, 4096, 65536, 1000003)]:
    k = d.ewise(ck.OP_MOD, id3, np.int32(G), keep=True)
    if mult != 1:
        k2 = d.ewise(ck.OP_MUL, k, np.int32(mult), ot=ck.INT32, keep=True); k.free(); k = k2
    h = None; best = 1e9; kb = 1e9
    for rep in range(3):
        d.sync(); d.timer_start()
        h = d.groupby_agg([k], [ck.RED_SUM], [v1], hint=hint, handle=h)
        best = min(best, d.timer_stop_ms()); kb = min(kb, d.last_kernel_ms())
    print(f"G={G:5d} mult={mult:8d} hint={hint:5d} groups={h.ngroups:5d} call {best:7.3f} ms kernel {kb:7.3f} ms", flush=True)
    h.destroy(); k.free()
