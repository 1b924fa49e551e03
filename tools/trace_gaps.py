#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the GPU idle time between consecutive kernels (launch-bound
steps show up as gaps).  usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]                       # steady state: drop warm-up
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, gaps, cur_end = 0, [], None
per = defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-70:]
    per[name][0] += e - s
    per[name][1] += 1
    if cur_end is None:
        cur_end = e
        busy += e - s
        continue
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
wall = t1 - t0
print("kernels %d  wall %.3f ms  busy(union) %.3f ms (%.1f%%)  idle %.3f ms in %d gaps (median gap %.2f us)" % (
    len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
tot = sum(v[0] for v in per.values())
for name, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%7.3f ms %5.1f%% n=%5d avg %8.2f us  %s" % (ns / 1e6, 100.0 * ns / tot, n, ns / n / 1e3, name))
