#!/usr/bin/env python3
"""Per-stream timeline of ONE bench step from a rocprofv3 --kernel-trace CSV (tools only).

    python tools/timeline.py <kernel_trace.csv> [first-kernel-substring]
Steps are delimited by the first kernel of a step (default: prior_argmax); the last complete step is analysed: span,
per-queue busy time, time during which 1 / 2 / 3+ kernels run, the longest kernels, gaps on the busiest queue."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prior_argmax"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 2:
    sys.exit("need two steps in the trace")
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
t0 = step[0][0]
t1 = max(r[1] for r in step)
print("step: %d kernels, span %.3f ms, sum of kernel durations %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in step) / 1e6))
byq = defaultdict(list)
for r in step:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
    print("  queue %s: %3d kernels, busy %.3f ms, first +%.3f ms, last end +%.3f ms" % (
        q, len(rs), sum(r[1] - r[0] for r in rs) / 1e6, (rs[0][0] - t0) / 1e6, (max(r[1] for r in rs) - t0) / 1e6))
ev = []
for s, e, _, _ in step:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
print("  concurrency: idle %.3f ms, 1 kernel %.3f, 2 kernels %.3f, 3+ %.3f" % tuple(hist[k] / 1e6 for k in range(4)))
mainq = max(byq.items(), key=lambda kv: sum(r[1] - r[0] for r in kv[1]))[0]
rs = byq[mainq]
gaps = sorted(((rs[i + 1][0] - max(r[1] for r in rs[:i + 1]), rs[i][2][:50], rs[i + 1][2][:50]) for i in range(len(rs) - 1)), reverse=True)
print("  largest gaps on queue %s:" % mainq)
for g, a, b in gaps[:8]:
    print("    %7.1f us between %s -> %s" % (g / 1e3, a, b))
agg = defaultdict(lambda: [0, 0])
for s, e, n, _ in step:
    k = n.split("(")[0][:60]
    agg[k][0] += e - s
    agg[k][1] += 1
print("  kernels by time:")
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
    print("    %8.3f ms x%-3d %s" % (t / 1e6, c, k))
if len(sys.argv) > 3:                      # dump one queue: start, duration (us), name
    for s, e, n, q in step:
        if q == sys.argv[3]:
            print("    +%8.1f us %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:70]))
