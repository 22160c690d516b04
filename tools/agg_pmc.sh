cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_aggpmc
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -i -E "icache|ifetch|SQC_|SQ_INST_LEVEL|SQ_WAVES|SQ_INSTS_VMEM|SQ_INSTS_SMEM|SQ_WAIT_INST|SQ_ACTIVE_INST_SCA|SQ_ACTIVE_INST_MISC|SQ_INSTS_BRANCH|TCP_" | sort -u | tr -s '\t ' ' ' | cut -d: -f2 | tr '\n' ' ' > $O/names.txt
i=0
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$i -o p -- python3 tools/agg_shapes.py 1e9 1 > $O/p$i.log 2>&1 || echo "pmc $pmc failed"
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r2_aggpmc/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "agg32_kernel" not in k and "dense_agg" not in k: continue
        did = int(r["Dispatch_Id"])
        rows.setdefault((k[k.find("agg"):][:34]), collections.OrderedDict()).setdefault(did, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, ds in rows.items():
    print(k)
    for did, c in ds.items(): print("   ", did, {n: f"{v:.3g}" for n, v in c.items()})
PY
cat $O/names.txt | fold -w 200
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
