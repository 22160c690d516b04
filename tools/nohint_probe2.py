import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
n = 10**9; K = 100
d = A.Device(0)
id3 = d.gen_column(ck.GEN_ID3, 42, 0, n, n, K); v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, K)
s = 1 << 20
sample = A.DevBuf(d, id3.ptr, np.int32, s, owned=False)
for hint in (4096, 65536, s, 0):
    d.sync(); t0 = time.perf_counter()
    try:
        h = d.groupby_agg([sample], [], [], hint=hint)
        msg = f"groups {h.ngroups}"
        h.destroy()
    except Exception as e:
        msg = "error " + str(e)[:60]
    d.sync(); print(f"sample 2^20 rows, hint={hint}: {1e3 * (time.perf_counter() - t0):9.3f} ms  {msg}", flush=True)
d.sync(); t0 = time.perf_counter()
h = d.groupby_agg([id3], [ck.RED_SUM], [v1], hint=0)
d.sync(); print(f"full, hint=0: {1e3 * (time.perf_counter() - t0):9.3f} ms groups {h.ngroups}", flush=True)
h.destroy()
d.sync(); t0 = time.perf_counter()
h = d.groupby_agg([id3], [ck.RED_SUM], [v1], hint=0)
d.sync(); print(f"full again, hint=0 (fresh handle): {1e3 * (time.perf_counter() - t0):9.3f} ms", flush=True)
