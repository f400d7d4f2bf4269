"""Per-kernel summary of a rocprofv3 --kernel-trace database (rocpd sqlite): calls, average / total duration, and
the average number of kernels in flight.  Usage: python tools/trace_summary.py results.db [last_fraction [stats.csv]]"""
import collections
import sqlite3
import sys

import numpy as np


def short(n):
    n = n.replace("_ZN4lgcn", "")
    for k in ("k_lc_tile", "k_lc_combine", "k_lc_plan", "k_att_pairs", "k_att_fused", "k_agg_mlp_bf2", "k_agg_mlp_bf", "k_agg_mlp",
              "k_pairs_rows", "k_pairs_scan", "k_csr_edges", "k_csr_sort", "k_scan", "k_graph_gather", "k_mapnet_input", "k_zero2",
              "k_widen"):
        if k in n:
            return k
    return n[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    rows = rows[int(len(rows) * (1 - frac)):]          # the steady-state tail of the run
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    wall = (t1 - t0) / 1e3
    agg = collections.defaultdict(list)
    for n, s, e, q in rows:
        agg[short(n)].append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("window %.1f us, %d dispatches, queues %s, kernel time / wall = %.2f in flight" % (
        wall, len(rows), sorted(set(r[3] for r in rows)), tot / wall))
    print("%-22s %7s %9s %10s %6s" % ("kernel", "calls", "avg us", "total us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-22s %7d %9.2f %10.1f %6.1f" % (k, len(v), np.mean(v), sum(v), 100 * sum(v) / tot))
    if len(sys.argv) > 3:      # per-kernel stats over the WHOLE run (full names), the layout of rocprofv3 --stats
        full = collections.defaultdict(list)
        for n, s, e, q in c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id" % (kd, ks)):
            full[n].append(e - s)
        allt = sum(sum(v) for v in full.values())
        with open(sys.argv[3], "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
            for n, v in sorted(full.items(), key=lambda kv: -sum(kv[1])):
                v = np.asarray(v, np.float64)
                f.write('"%s",%d,%d,%.6f,%.4f,%d,%d,%.6f\n' % (n, len(v), v.sum(), v.mean(), 100 * v.sum() / allt, v.min(), v.max(), v.std()))


if __name__ == "__main__":
    main()
