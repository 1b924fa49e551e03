#!/usr/bin/env python3
"""Registers / LDS / spills of every kernel in the built objects (supervised_dispnet_amd/csrc/_obj/*.o), from the code objects' metadata.
usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt      (after csrc/build.py; needs /opt/rocm/lib/llvm/bin)"""
import glob
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(ROOT, "supervised_dispnet_amd", "csrc", "_obj", "*.o"))):
            name = os.path.basename(f)[:-2]
            fb, co = os.path.join(tmp, name + ".fatbin"), os.path.join(tmp, name + ".co")
            subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", f, fb], check=False)
            if not os.path.exists(fb) or os.path.getsize(fb) == 0:
                continue
            lst = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=" + fb], capture_output=True, text=True).stdout.split()
            tgt = [t for t in lst if "gfx950" in t]
            if not tgt:
                continue
            subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt[0], "--input=" + fb, "--output=" + co], check=False)
            out = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in out.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
                if not m:
                    continue
                cur[m.group(1)] = m.group(2).strip()
                if m.group(1) == "wavefront_size":
                    rows.append((name, cur.get("name"), cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"),
                                 cur.get("group_segment_fixed_size"), cur.get("vgpr_spill_count"), cur.get("private_segment_fixed_size")))
                    cur = {}
    names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("# kernel resources of the gfx950 code objects (static LDS only: kernels with `extern __shared__` add their launch-time bytes)")
    print("%-18s %5s %5s %5s %8s %6s %7s  %s" % ("file", "vgpr", "agpr", "sgpr", "lds(B)", "spill", "scratch", "kernel"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)
        print("%-18s %5s %5s %5s %8s %6s %7s  %s" % (r[0], r[2], r[3], r[4], r[5], r[6], r[7], n))


if __name__ == "__main__":
    main()
