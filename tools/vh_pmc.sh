cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_vhpmc
mkdir -p $O
cat > $O/probe.py <<'PY'
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
d = A.Device(0)
n = 1_000_000_000
price = d.gen_column(ck.GEN_PRICE, 42, 0, n, n, 100)
out = d.empty(n, np.int32)
for _ in range(2):
    d.scan(ck.SCAN_MAXW, price, 1000, keep=True, out=out)
d.sync()
print("kernel ms", d.last_kernel_ms())
PY
i=0
for pmc in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$i -o p -- python3 $O/probe.py > $O/p$i.log 2>&1 || { echo "pmc $pmc failed"; tail -5 $O/p$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r2_vhpmc/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "window_minmax" not in k: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for did, c in rows.items(): print(did, {n: f"{v:.3g}" for n, v in c.items()})
PY
tail -2 $O/p1.log
rm -rf $O/p1 $O/p2 $O/p3
