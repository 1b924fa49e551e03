#!/usr/bin/env python3
"""Per-wave phase times of dn::wino_conv8_kernel's main loop (DN_WINO_DBG=2052: in-kernel timestamps at the chunk's head, after every
12 of its 48 slots and around its barrier).  usage: python tools/wino8_phases.py [affine]"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch, torch.nn as nn
from supervised_dispnet_amd import engine, _lib
dev = torch.device("cuda:0")
buf = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
os.environ["DN_WINO_DBG"] = "2052"
os.environ["DN_WINO_DBGPTR"] = hex(buf.data_ptr())
AFF = len(sys.argv) > 1 and sys.argv[1] == "affine"
for cin, cout, H, W in [(128, 128, 64, 208), (256, 256, 32, 104), (512, 512, 16, 52)]:
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    layer = engine.ConvLayer(mod)
    x = engine.Act(torch.randn(32, H, W, cin, device=dev), 32, H, W, cin)
    if AFF:
        x.scale = torch.rand(cin, device=dev) + 0.5
        x.shift = torch.rand(cin, device=dev) - 0.5
    for _ in range(3):
        engine.conv_forward(layer, [engine.Piece(x)], bn_stats=AFF)
    torch.cuda.synchronize()
    name = _lib.load().dn_last_kernel().decode()
    nblk = ((32 * H * W // 4 + 63) // 64) * (cout // 64)
    grid = (nblk + 7) // 8 * 8
    t = buf[: grid * 8].view(-1, 8).cpu()
    ph = buf[grid * 8: grid * 8 + grid * 64].view(grid, 8, 8).cpu().float()
    live = t[:, 3] > 0
    ph = ph[live]
    t = t[live]
    n = ph[:, :, 6].clamp(min=1)
    print("%s cin%d cout%d %dx%d%s: %d blocks, %d chunks" % (name, cin, cout, H, W, " affine+stats" if AFF else "", len(t), int(n[0, 0])))
    print("   block: prologue %.0f loop %.0f (%.0f per chunk) epilogue %.0f" % ((t[:, 1] - t[:, 0]).float().mean(), (t[:, 2] - t[:, 1]).float().mean(),
                                                                             (t[:, 2] - t[:, 1]).float().mean() / int(n[0, 0]), (t[:, 3] - t[:, 2]).float().mean()))
    print("   wave  simd   head    q0    q1    q2    q3  barrier  | sum")
    for w in range(8):
        v = [(ph[:, w, k] / n[:, w]).mean().item() for k in range(6)]
        simd = int(ph[0, w, 7].item()) >> 4 & 3
        print("   %4d  %4d  %5.0f %5.0f %5.0f %5.0f %5.0f  %6.0f  | %5.0f" % (w, simd, *v, sum(v)))
    buf.zero_()
