#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, like --stats CSV.
usage: python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db [steps] > profiles/rNN_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(dn::[a-zA-Z0-9_]+(<[^(]*>)?)", name)
    if m:
        return m.group(1)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += dur
    total = sum(v[1] for v in agg.values())
    print("# rocprofv3 --kernel-trace summary: %d dispatches, %.3f ms of kernel time%s" % (
        len(rows), total / 1e6, (" over %d steps" % steps) if steps else ""))
    print("%-70s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-70s %8d %12.3f %12.2f %6.2f%%" % (k, n, t / 1e6, t / n / 1e3, 100.0 * t / total))


if __name__ == "__main__":
    main()
