#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh.

Units / corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units derived
from the L2's fabric request counters; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes, i.e. reports HALF of a
wide coalesced read -- doubled here.  Both factors are checked against a kernel of known traffic in the same run:
dn::adam_kernel streams 4 arrays in (p, g, m, v) and 3 out over the whole arena; the summary prints measured / expected for
it so the calibration is visible next to the numbers it calibrates.
usage: python tools/pmc_traffic_summary.py gpurun_out/pmc_<tag>"""
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(dn::[a-zA-Z0-9_]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:80]


def collect(folder, counter):
    agg = {}
    for f in glob.glob(folder + "/" + counter + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            a = agg.setdefault(short(row["Kernel_Name"]), [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row.get("Dispatch_Id"))
    return {k: (v[0], len(v[1])) for k, v in agg.items()}


def main():
    folder = sys.argv[1]
    arena_bytes = float(sys.argv[2]) if len(sys.argv) > 2 else 19873156 * 4.0
    fetch, write = collect(folder, "FETCH_SIZE"), collect(folder, "WRITE_SIZE")
    out = {"unit": "bytes per launch", "corrections": {"FETCH_SIZE": "x1024 (KiB) x2 (gfx950 128-B requests tallied as 64 B)",
                                                        "WRITE_SIZE": "x1024 (KiB)"}, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out["kernels"][k] = {"launches": max(nf, nw), "read_bytes": f * 1024.0 * 2.0 / max(nf, 1), "write_bytes": w * 1024.0 / max(nw, 1)}
    for name in ("dn::adam_kernel", "dn::adam_dev_kernel"):
        ad = out["kernels"].get(name)
        if ad and ad["write_bytes"] > 0:
            # with the per-bucket update (FusedAdam.overlap_backward, the default at 32 images) one launch covers one gradient bucket:
            # the average launch is 1/k of the arena, k = the number of buckets
            k = max(1, int(round(3 * arena_bytes / ad["write_bytes"])))
            out["calibration"] = {"kernel": name, "launches_per_arena_pass": k, "expected_read_bytes": 4 * arena_bytes, "expected_write_bytes": 3 * arena_bytes,
                                  "read_measured_over_expected": k * ad["read_bytes"] / (4 * arena_bytes),
                                  "write_measured_over_expected": k * ad["write_bytes"] / (3 * arena_bytes)}
            break
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
