#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue, grid, kernel -- plus per-queue busy
time and the idle time of the union.  The step is the span between two consecutive adam_kernel launches near the end of the trace.
usage: python tools/step_timeline.py <kernel_trace.csv> [steps-from-end | 0 = the median step of the last 20]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
if back > 0:
    a, b = idx[-back], idx[-back + 1]
else:                                   # 0: the step of median length among the last 20 (a one-off stall does not end up as "the" timeline)
    spans = sorted((int(rows[y]["End_Timestamp"]) - int(rows[x]["End_Timestamp"]), x, y) for x, y in zip(idx[-21:-1], idx[-20:]))
    _, a, b = spans[len(spans) // 2]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
queues = {}
cur_end, idle = t0, 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    queues.setdefault(q, [0, 0])
    queues[q][0] += e - s
    queues[q][1] += 1
    gap = s - cur_end
    if gap > 0:
        idle += gap
    cur_end = max(cur_end, e)
    print("%9.2f +%8.2f us  q%-3s gap %7.2f  grid %8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap / 1e3, r.get("Grid_Size_X", "?"),
                                                           r["Kernel_Name"].split("(")[0][-70:]))
print("step %.3f ms, %d kernels, idle (no kernel on any queue) %.3f ms" % ((cur_end - t0) / 1e6, len(step), idle / 1e6))
for q, (ns, n) in queues.items():
    print("queue %s: %d kernels, %.3f ms" % (q, n, ns / 1e6))
