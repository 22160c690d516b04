cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_aggpmc
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 tools/q5_probe.py 1e9 q5 > $O/a.log 2>&1 || true
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 tools/q5_probe.py 1e9 q5 > $O/b.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r2_aggpmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "p1_agg" in k or "p2_scatter" in k: acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: f"{sum(x)/len(x):.3e}" for c, x in v.items()})
PY
tail -3 $O/a.log $O/b.log | grep -v simple_timer | head
rm -rf $O/a $O/b
