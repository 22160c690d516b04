"""scratch: h2o Q1 through a FRESH handle with hint 0 (what the header layer's first call does): whole-call time; run under rocprofv3 for the kernels"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
d = A.Device(0)
id1, v1 = d.gen_column(ck.GEN_ID1, 42, 0, n, n, 100), d.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
for rep in range(6):
    d.sync(); d.timer_start()
    h = d.groupby_agg([id1], [ck.RED_SUM], [v1], hint=0)
    ms = d.timer_stop_ms()
    print(f"rep{rep}: {ms:.3f} ms plan {h.plan} groups {h.ngroups}", flush=True)
    h.destroy()
