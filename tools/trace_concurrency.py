"""Concurrency inside the busiest multi-queue window of a rocprofv3 kernel trace: kernels in flight on average and the
per-kernel durations there.  Usage: python tools/trace_concurrency.py results.db [window_dispatches]"""
import collections
import sqlite3
import sys

import numpy as np


def main():
    db = sqlite3.connect(sys.argv[1])
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    st = np.array([r[1] for r in rows]); en = np.array([r[2] for r in rows]); q = np.array([r[3] for r in rows])
    best = None
    for i in range(0, max(len(rows) - W, 1), 100):
        nq = len(set(q[i:i + W]))
        wall = en[i:i + W].max() - st[i]
        busy = (en[i:i + W] - st[i:i + W]).sum()
        if nq >= 3 and (best is None or busy / wall > best[0]):
            best = (busy / wall, i)
    if best is None:
        print("no multi-queue window")
        return
    i = best[1]
    sel = rows[i:i + W]
    wall = (max(r[2] for r in sel) - sel[0][1]) / 1e3
    agg = collections.defaultdict(list)
    for n, s, e, _ in sel:
        agg[n].append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    # time with k kernels running
    ev = sorted([(r[1], 1) for r in sel] + [(r[2], -1) for r in sel])
    lvl, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[lvl] += t - last
        lvl += d
        last = t
    span = sum(hist.values())
    print("window of %d dispatches: wall %.0f us, kernel time %.0f us, %.2f kernels in flight on average" % (len(sel), wall, tot, tot / wall))
    print("share of the time with k kernels running:", ", ".join("%d: %.0f%%" % (k, 100 * v / span) for k, v in sorted(hist.items())))
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("%5d %8.2f %9.1f  %s" % (len(v), np.mean(v), sum(v), n[:100]))


if __name__ == "__main__":
    main()
