cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_pppmc
mkdir -p $O
cat > $O/probe.py <<'PY'
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
d = A.Device(0)
n = 1_000_000_000
id1 = d.gen_column(ck.GEN_ID1, 42, 0, n, n, 100)
g = d.groupby_build([id1], hint=128)
off = d.empty(g.ngroups + 1, np.uint32); rows = d.empty(n, np.uint32)
for _ in range(2):
    d._chk(d.lib.aqg_groupby_postproc(g.h, C.c_void_p(off.ptr), C.c_void_p(rows.ptr)), "pp")
d.sync()
PY
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$i -o p -- python3 $O/probe.py > $O/p$i.log 2>&1 || { echo "pmc $pmc failed"; tail -3 $O/p$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r2_pppmc/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "radix_" not in k: continue
        rows.setdefault(k[k.find("radix"):][:40], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in rows.items(): print(k, {n: f"{sum(v)/len(v):.3g}" for n, v in c.items()})
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
