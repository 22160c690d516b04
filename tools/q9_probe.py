"""scratch: h2o Q9 grouped corr at 1e9 rows (for rocprofv3)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
id2, id4, v1, v2 = (d.gen_column(c, 42, 0, n, n, 100) for c in (ck.GEN_ID2, ck.GEN_ID4, ck.GEN_V1, ck.GEN_V2))
gb = d.groupby_build([id2, id4])
for rep in range(3):
    d.sync(); d.timer_start(); d.grouped_corr(gb, v1, v2); print("corr: %.3f ms groups %d" % (d.timer_stop_ms(), gb.ngroups), flush=True)
d.close()
