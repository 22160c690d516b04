"""scratch: group-by over a 4-byte key that is unique per row (1e9 groups of one row) -- packed keys beyond the partition plans' 2^25 groups"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
key = d.to_device(((np.arange(n, dtype=np.int64) * 2654435761) % (1 << 31)).astype(np.int32)) if len(sys.argv) > 2 else d.to_device(np.arange(n, dtype=np.int32))
v3 = d.gen_column(ck.GEN_V3, 42, 0, n, n, 100)
for hint in (n, 0):
    gb = None
    for rep in range(3):
        d.sync(); t0 = time.perf_counter(); d.timer_start()
        gb = d.groupby_agg([key], [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=hint, handle=gb)
        ev = d.timer_stop_ms(); d.sync(); dt = (time.perf_counter() - t0) * 1e3
        print(f"agg hint {hint:>10}: {ev:9.2f} ms on the stream, {dt:9.2f} ms wall  groups {gb.ngroups} plan {gb.plan}", flush=True)
    gb.destroy()
