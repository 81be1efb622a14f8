"""Per-kernel totals of a rocprofv3 --kernel-trace database (rocpd SQLite): python tools/kt_summary.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = db.execute(f"select s.display_name, count(*), avg(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 "
                  f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.display_name order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-24s %5d avg %8.2f max %8.1f tot %8.2f ms %5.1f%%" % (r[0][:24], r[1], r[2], r[3], r[4], 100 * r[4] / tot))
print("total %.2f ms" % tot)
