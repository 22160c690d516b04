"""scratch: does the fast / slow mode of h2o Q5 follow the PROCESS or the ALLOCATION?  Several contexts one after the other in one process, each with its own
columns and workspace; optionally ballast allocations in between to shift where the next buffers land."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = 1_000_000_000
K = 100
keep = []
for trial in range(6):
    d = A.Device(0)
    cols = [d.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID6, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3)]
    h = None
    ms = []
    for rep in range(3):
        d.sync(); d.timer_start()
        h = d.groupby_agg([cols[0]], [ck.RED_SUM] * 3, cols[1:], hint=n // K + 1024, handle=h)
        ms.append(d.timer_stop_ms())
    print(f"trial {trial}: {ms[1]:.3f} {ms[2]:.3f} ms   id6@{cols[0].ptr:#x} v3@{cols[3].ptr:#x}", flush=True)
    h.destroy()
    for c in cols: c.free()
    if trial % 2 == 1:
        keep.append(d.empty(300_000_000 + 77_777 * trial, np.int32))    # ballast that stays: the next context's buffers land elsewhere
        keep_dev = d
        continue
    d.close()
