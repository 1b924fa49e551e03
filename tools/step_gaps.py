#!/usr/bin/env python3
"""Every training step of a rocprofv3 --kernel-trace CSV (spans between consecutive adam_tick launches): length, idle time of the union of the
queues, and the largest holes (no kernel on any queue) with the kernel that ends them -- to tell a hole every step has from a one-off.
usage: python tools/step_gaps.py <kernel_trace.csv> [holes-per-step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nholes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_tick" in r["Kernel_Name"]]
for a, b in zip(idx[:-1], idx[1:]):
    step = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    cur_end, idle, holes = t0, 0, []
    for k, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_end:
            idle += s - cur_end
            holes.append((s - cur_end, k, (cur_end - t0) / 1e3, r["Kernel_Name"].split("(")[0][-48:]))
        cur_end = max(cur_end, e)
    holes.sort(reverse=True)
    print("step %.3f ms  %3d kernels  idle %.3f ms   " % ((cur_end - t0) / 1e6, len(step), idle / 1e6) +
          "  ".join("%.0f us @%.0f #%d %s" % (h[0] / 1e3, h[2], h[1], h[3]) for h in holes[:nholes]))
