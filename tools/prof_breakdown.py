"""Per-forward kernel breakdown of a rocprofv3 kernel-trace database (second half of the run; forwards are counted by
k_index_sort launches): python tools/prof_breakdown.py <results.db> [rows]."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[len(rows) // 2:]
nrep = max(1, len([r for r in rows if "k_index_sort" in r[0]]))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[n[:90]]
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values()) / nrep
own = sum(v[1] for k, v in agg.items() if "lgcn" in k) / nrep
print("forwards %d  kernel us/fwd %.1f  (lgcn kernels %.1f, others %.1f)  launches/fwd %.1f  span/fwd %.1f"
      % (nrep, tot, own, tot - own, len(rows) / nrep, (rows[-1][2] - rows[0][1]) / 1e3 / nrep))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-92s %6.2f /fwd %7.1f us/fwd %6.2f us/call" % (k, v[0] / nrep, v[1] / nrep, v[1] / v[0]))
