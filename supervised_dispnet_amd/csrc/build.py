#!/usr/bin/env python3
"""Build libdispnet_hip.so (gfx950 only) in-tree with hipcc.  No torch, no cmake: `python build.py`.

The shared object lands next to the Python package (supervised_dispnet_amd/libdispnet_hip.so) so it travels with the
source snapshot to the GPU box; objects are cached per source mtime under csrc/_obj/.
"""
import os
import pathlib
import shutil
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
PKG = HERE.parent
ROOT = PKG.parent
SOURCES = ["dn_plan.hip", "dn_conv.hip", "dn_pointwise.hip", "dn_loss.hip", "dn_warp.hip", "dn_ordinal.hip", "dn_ordhead.hip", "dn_direct.hip", "dn_winograd.hip", "dn_winograd8.hip", "dn_winograd_wgrad.hip", "dn_thin.hip", "dn_lds3.hip", "dn_lds3k.hip", "dn_lds3_wgrad.hip", "dn_stemk.hip", "dn_wgrad_x3.hip", "dn_input.hip", "dn_zoo.hip", "dn_ubench.hip", "dn_tape.hip"]
OUT = PKG / "libdispnet_hip.so"
ARCH = "gfx950"


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and pathlib.Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build(force=False, verbose=False):
    cc = hipcc()
    obj_dir = HERE / "_obj"
    obj_dir.mkdir(exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", str(ROOT / "include"), "-I", str(HERE),
             "-Wall", "-Wno-unused-function"]
    deps = sorted(HERE.glob("*.h")) + sorted((ROOT / "include").glob("*.h")) + [pathlib.Path(__file__)]     # every header: an edit rebuilds all objects
    newest_dep = max(p.stat().st_mtime for p in deps)
    objs, procs = [], []
    for src in SOURCES:
        s = HERE / src
        o = obj_dir / (s.stem + ".o")
        objs.append(o)
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, newest_dep):
            cmd = [cc, *flags, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or not OUT.exists():
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(OUT), *map(str, objs)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
