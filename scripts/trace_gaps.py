#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel trace (results.db of `rocprofv3 --kernel-trace`): per iteration of the
training loop — an iteration starts at the kernel named by --marker (default: the first proposal density kernel of the level with
the most samples) — the busy time (union of kernel intervals), the idle time, and the largest gaps with the kernels on either side.
    python scripts/trace_gaps.py <results.db> [--marker substring] [--skip N] [--show K]"""
import argparse
import sqlite3
import statistics

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--marker", default="piecewise_bins_kernel")
ap.add_argument("--marker2", default="select_bins_kernel")
ap.add_argument("--skip", type=int, default=30, help="iterations to skip at the start (warm-up, capture)")
ap.add_argument("--show", type=int, default=6)
a = ap.parse_args()
db = sqlite3.connect(a.db)
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
tab = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
rows = db.execute(f"select name, start, end from {tab} order by start").fetchall()
starts = [i for i, r in enumerate(rows) if a.marker in r[0] or a.marker2 in r[0]]
its = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)][a.skip:]
busy, idle, span, gaps = [], [], [], []
for lo, hi in its:
    ks = rows[lo:hi]
    t0, t1 = ks[0][1], rows[hi][1]
    cur_end, b = t0, 0
    for n, s, e in ks:
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, prev, n))
        b += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end, prev = e, n
    if t1 > cur_end:
        gaps.append(((t1 - cur_end) / 1e3, prev, "<next iteration>"))
    busy.append(b / 1e3)
    span.append((t1 - t0) / 1e3)
    idle.append((t1 - t0 - b) / 1e3)
print(f"{len(its)} iterations: span median {statistics.median(span):.1f} us, busy {statistics.median(busy):.1f} us, idle {statistics.median(idle):.1f} us")
agg = {}
for g, p, n in gaps:
    k = (p[:48], n[:48])
    agg.setdefault(k, []).append(g)
top = sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: a.show * 3]
print("largest idle gaps by (kernel before -> kernel after): total us per iteration, count per iteration, mean us")
for (p, n), v in top:
    print(f"  {sum(v) / len(its):7.1f} {len(v) / len(its):5.2f} {sum(v) / len(v):7.1f}   {p}  ->  {n}")
