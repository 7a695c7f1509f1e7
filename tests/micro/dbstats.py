"""per-kernel stats of a rocprofv3 results database: python tests/micro/dbstats.py <db> [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d "
     f"join {ks} s on d.kernel_id=s.id group by 1 order by 5 desc limit {n}")
for r in db.execute(q):
    print("%-78s %5d avg %9.1f min %9.1f us" % (r[0][:78], r[1], r[2], r[3]))
