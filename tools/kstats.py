import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6,max(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in list(db.execute(q))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r[0][:100]:100s} n={r[1]:4d} tot={r[2]:9.2f} avg={r[3]:8.3f} min={r[4]:8.3f} max={r[5]:8.3f}")
