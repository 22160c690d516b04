"""scratch: hint-0 group-by over a column SORTED by its key (every key in one run of K rows): what the cardinality estimate from a sample makes of it"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
K = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100
d = A.Device(0)
key = d.to_device((np.arange(n, dtype=np.int64) // K).astype(np.int32))
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
for hint in (0, n // K + 1024):
    for rep in range(2):
        d.sync(); t0 = time.perf_counter()
        gb = d.groupby_agg([key], [ck.RED_SUM], [v1], hint=hint)
        d.sync(); dt = (time.perf_counter() - t0) * 1e3
        print(f"agg   hint {hint:>9}: {dt:8.2f} ms  groups {gb.ngroups} plan {gb.plan}", flush=True)
        gb.destroy()
for rep in range(2):
    d.sync(); t0 = time.perf_counter()
    gb = d.groupby_build([key])
    d.sync(); dt = (time.perf_counter() - t0) * 1e3
    print(f"build hint 0: {dt:8.2f} ms  groups {gb.ngroups} plan {gb.plan}", flush=True)
    gb.destroy()
