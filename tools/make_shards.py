#!/usr/bin/env python3
"""Pre-decode a scene-folder dataset (the reference's KITTI layout: scene/{*.jpg, *.npy, cam.txt} + train.txt / val.txt) into the
uint8 shards supervised_dispnet_amd.shards.ShardLoader reads.   usage: tools/make_shards.py DATA_ROOT OUT_DIR [--val] [--with-gt]"""
import argparse
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from supervised_dispnet_amd.shards import write_shards  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("data")
ap.add_argument("out")
ap.add_argument("--val", action="store_true")
ap.add_argument("--with-gt", action="store_true", help="validation set with ground truth (every frame a sample)")
ap.add_argument("--sequence-length", type=int, default=3)
a = ap.parse_args()
m = write_shards(a.data, a.out, train=not a.val, sequence_length=a.sequence_length, with_gt=a.with_gt)
print("%d frames, %d samples, %dx%d -> %s" % (m["frames"], len(m["samples"]), m["H"], m["W"], a.out))
