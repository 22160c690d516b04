cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "lds|SQ_INST_CYCLES_VMEM|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_LDS" | cut -c1-200 | sort -u | head -60 > gpurun_out/lds_counters.txt
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $pmc | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc -d gpurun_out/ldspmc_$tag -o p -- python3 tools/perf_probe.py 1e9 generic > gpurun_out/ldspmc_$tag.log 2>&1 || echo "pmc $pmc failed"
done
python3 - <<'PY'
import sqlite3, glob, collections
res = collections.defaultdict(dict)
for db in glob.glob("gpurun_out/ldspmc_*/p_results.db"):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    q = f"select s.kernel_name, i.name, avg(e.value), count(*) from {pc} e join {ip} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by 1,2"
    for k, n, v, cnt in c.execute(q):
        if "agg32" in k: res[k[:60]][n] = v
for k, d in sorted(res.items()):
    print(k, {n: f"{v:.3g}" for n, v in sorted(d.items())})
PY
