#!/usr/bin/env python3
"""Time single conv layers through the C ABI (HIP events), for kernel tuning.
   python tools/conv_microbench.py [--reps 20] [--layers name,...]"""
import argparse, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import torch.nn as nn
from supervised_dispnet_amd import engine
from supervised_dispnet_amd._lib import ACT_NONE

LAYERS = {  # name: (cin, cout, H, W, transposed)
    "c64_64_128x416": (64, 64, 128, 416, False),
    "c128_128_64x208": (128, 128, 64, 208, False),
    "c256_256_32x104": (256, 256, 32, 104, False),
    "c512_512_16x52": (512, 512, 16, 52, False),
    "c512_512_8x26": (512, 512, 8, 26, False),
    "c768_256_8x26": (768, 256, 8, 26, False),
    "t512_256_4x13": (512, 256, 4, 13, True),
    "c16_16_128x416": (16, 16, 128, 416, False),
    "c32_16_128x416": (32, 16, 128, 416, False),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", default=",".join(LAYERS))
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--affine", action="store_true", help="give the input a pending BatchNorm-apply + ReLU (forward / wgrad loaders)")
    ap.add_argument("--stats", action="store_true", help="forward: also write the BatchNorm partial statistics (the encoder's convolutions)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.layers.split(","):
        if name in LAYERS:
            cin, cout, H, W, tr = LAYERS[name]
        else:                                   # free form: c<cin>_<cout>_<H>x<W> / t<cin>_<cout>_<H>x<W> / p<cin>_<cout>_<H>x<W> (1x1)
            a, b, hw = name[1:].split("_")
            cin, cout, (H, W), tr = int(a), int(b), map(int, hw.split("x")), name[0] == "t"
        k1 = name[0] == "p"                     # p<cin>_<cout>_<H>x<W>: a 1x1 convolution (ResNet bottleneck layers)
        mod = (nn.ConvTranspose2d(cin, cout, 4, 2, 1) if tr else (nn.Conv2d(cin, cout, 1, 1, 0) if (name not in LAYERS and k1) else nn.Conv2d(cin, cout, 3, 1, 1))).to(dev)
        layer = engine.ConvLayer(mod, transposed=tr)
        x = engine.Act(torch.randn(args.batch, H, W, cin, device=dev), args.batch, H, W, cin)
        if args.affine:
            x.scale = torch.rand(cin, device=dev) + 0.5
            x.shift = torch.rand(cin, device=dev) - 0.5
        pieces = [engine.Piece(x)]
        y, _, _ = engine.conv_forward(layer, pieces)
        OH, OW = y.shape[1], y.shape[2]
        dy = torch.randn_like(y)
        flops = 2 * layer.macs(args.batch, H, W, OH, OW)
        def run(kind):
            if kind == "fwd":
                engine.conv_forward(layer, pieces, bn_stats=args.stats)
            elif kind == "dgrad":
                x.grad = None
                engine.conv_dgrad(layer, dy, args.batch, OH, OW, pieces, (H, W))
            else:
                engine.conv_wgrad(layer, pieces, dy, (OH, OW))
        for kind in args.what.split(","):
            for _ in range(3):
                run(kind)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                run(kind)
            engine.join_side_stream()          # (weight gradients run on the side streams: the timing stream has to wait for them)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print("%-18s %-6s %8.3f ms  %7.1f TFLOP/s  %s" % (name, kind, ms, flops / ms / 1e9, engine._lib.load().dn_last_kernel().decode()))

if __name__ == "__main__":
    main()
