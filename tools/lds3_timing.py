#!/usr/bin/env python3
"""Phase ticks inside the LDS-resident weight-gradient kernel of iconv1 (DN_LDS3_DBG=1): per wave and tile, the clock64 ticks spent waiting
for the next tile's loads, splitting + writing LDS, at the two barriers, issuing loads and in the matrix loop."""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch, torch.nn as nn
dev = torch.device("cuda:0")
buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
os.environ["DN_LDS3_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "1"      # 2: no loads, 3: loads from a 64 KB window (ablations)
os.environ["DN_WINO_DBGPTR"] = hex(buf.data_ptr())
from supervised_dispnet_amd import engine, _lib
which = sys.argv[2] if len(sys.argv) > 2 else "iconv1"
if which == "iconv1":
    N, H, W, cout, nblk, nw = 32, 64, 208, 32, 256, 8
    mod = nn.Conv2d(97, 32, 3, 1, 1).to(dev)
    a = engine.Act(torch.randn(N, H, W, 32, device=dev), N, H, W, 32)
    b = engine.Act(torch.randn(N, H, W, 64, device=dev), N, H, W, 64)
    d = engine.Act(torch.randn(N, H // 2, W // 2, 1, device=dev), N, H // 2, W // 2, 1)
    pieces = [engine.Piece(a), engine.Piece(b), engine.Piece(d, up=1)]
else:                       # iconv0
    N, H, W, cout, nblk, nw = 32, 128, 416, 16, 512, 4
    mod = nn.Conv2d(17, 16, 3, 1, 1).to(dev)
    a = engine.Act(torch.randn(N, H, W, 16, device=dev), N, H, W, 16)
    d = engine.Act(torch.randn(N, H // 2, W // 2, 1, device=dev), N, H // 2, W // 2, 1)
    pieces = [engine.Piece(a), engine.Piece(d, up=1)]
layer = engine.ConvLayer(mod)
dy = torch.randn(N, H, W, cout, device=dev)
names = ["load wait", "split + LDS writes", "barrier A", "load issue", "matrix loop", "barrier B"]
fwd = len(sys.argv) > 3 and sys.argv[3] == "fwd"
if fwd and which == "iconv1":
    names = ["load wait", "split + LDS writes", "barrier A", "load issue", "matrix", "exchange + stores", "barrier B"]
elif fwd:
    names = ["load wait", "split + LDS writes", "barrier A", "load issue", "matrix + stores", "barrier B"]
for it in range(3):
    if fwd:
        engine.conv_forward(layer, pieces)
    else:
        engine.conv_wgrad(layer, pieces, dy, (H, W))
torch.cuda.synchronize()
print("kernel:", _lib.load().dn_last_kernel().decode())
t = buf[: nblk * 8 * 8].view(nblk, 8, 8).cpu().double()[:, :nw]
nk = len(names)
tiles = t[:, :, 7 if nk == 7 else 6].clamp(min=1)
per = t[:, :, :nk] / tiles[:, :, None]
print("tiles per block: %.1f" % tiles.mean().item())
print("%-22s" % "wave" + "".join("%10d" % w for w in range(nw)) + "      mean")
for k, n in enumerate(names):
    print("%-22s" % n + "".join("%10.0f" % per[:, w, k].mean().item() for w in range(nw)) + "%10.0f" % per[:, :, k].mean().item())
print("%-22s" % "sum" + "".join("%10.0f" % per[:, w, :].sum(-1).mean().item() for w in range(nw)))
