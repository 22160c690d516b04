#!/bin/bash
# run on the GPU box: SQ counters of the kernels whose name contains $1 while `python3 tools/q5_probe.py 1e9 $2` runs
# (counters in their own passes, no kernel trace beside them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PAT=${1:-p1_agg}; W=${2:-q5}; O=gpurun_out/kpmc; CMD=${3:-"tools/q5_probe.py 1e9 $W"}
rm -rf $O; mkdir -p $O
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/p$i -o p -- python3 $CMD > $O/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $O/p$i.log; }
done
python3 - "$PAT" <<'PY'
import csv, glob, collections, sys
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/kpmc/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sys.argv[1] not in k: continue
        rows.setdefault(k[k.find(sys.argv[1]):][:60], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in rows.items():
    print(k)
    for n, v in c.items(): print(f"   {n:28s} {sum(v)/len(v):14.4g}  (x{len(v)})")
PY
rm -rf $O/p?
