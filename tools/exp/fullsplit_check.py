#!/usr/bin/env python3
"""Round-5 experiment: the 8-wave Winograd kernel with EVERY tile split along K (DN_WINO8_FULLSPLIT=1) against the unsplit 4-wave
kernel on small grids: values (fp32 summation order apart), statistics partials, input gradients, determinism."""
import os, pathlib, sys
os.environ.setdefault("DN_WINO8_FULLSPLIT", "1")
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch, torch.nn as nn
from supervised_dispnet_amd import engine, _lib
DEV = torch.device("cuda:0")
torch.manual_seed(5)
CASES = [(4, 16, 52, [512], 512, False), (4, 16, 52, [256], 512, True), (4, 16, 52, [512], 256, False), (4, 8, 26, [512, 512], 512, False),
         (4, 16, 52, [256, 512], 256, True), (4, 32, 104, [128, 256], 128, False), (4, 16, 24, [128, 256, 1], 64, False), (8, 8, 26, [512], 512, True)]
bad = 0
for N, H, W, cins, cout, bn in CASES:
    mod = nn.Conv2d(sum(cins), cout, 3, 1, 1).to(DEV)
    layer = engine.ConvLayer(mod)
    acts = []
    for i, c in enumerate(cins):
        if c == 1:
            a = engine.Act(torch.rand(N, H // 2, W // 2, 1, device=DEV) * 2, N, H // 2, W // 2, 1)
        else:
            a = engine.Act(torch.randn(N, H, W, c, device=DEV), N, H, W, c)
            if bn and i == 0:
                a.scale = torch.rand(c, device=DEV) + 0.5
                a.shift = torch.rand(c, device=DEV) - 0.5
        acts.append(a)
    pieces = [engine.Piece(a, up=(a.C == 1)) for a in acts]
    dy = torch.randn(N, H, W, cout, device=DEV)
    res = {}
    for split in (True, False):
        engine.SPLITK = split
        for a in acts:
            a.grad = None
        y, partial, _ = engine.conv_forward(layer, pieces, bn_stats=True)
        kf = _lib.load().dn_last_kernel().decode()
        engine.conv_dgrad(layer, dy, N, H, W, pieces, (H, W))
        kd = _lib.load().dn_last_kernel().decode()
        torch.cuda.synchronize()
        res[split] = (y.clone(), partial.clone(), [a.grad.clone() for a in acts], kf, kd)
    engine.SPLITK = True
    y2, _, _ = engine.conv_forward(layer, pieces, bn_stats=True)
    torch.cuda.synchronize()
    ys, yn = res[True][0], res[False][0]
    e_y = float((ys - yn).abs().max()) / float(yn.abs().max())
    e_g = max(float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(res[True][2], res[False][2]))
    # the partial rows cover other tile groups (64 vs 32 tiles per block): compare the column sums
    ps, pn = res[True][1], res[False][1]
    e_p = float((ps[..., 0].sum(0) - pn[..., 0].sum(0)).abs().max()) / float(pn[..., 0].sum(0).abs().max())
    ok = e_y <= 2e-5 and e_g <= 2e-5 and e_p <= 1e-4 and torch.equal(y2, ys)
    bad += 0 if ok else 1
    print("%s N%d %dx%d %s->%d bn=%d  y %.2e  dx %.2e  stats %.2e  det %s  fwd %s | %s   dgrad %s | %s" % (
        "OK " if ok else "BAD", N, H, W, cins, cout, bn, e_y, e_g, e_p, torch.equal(y2, ys), res[True][3], res[False][3], res[True][4], res[False][4]))
sys.exit(1 if bad else 0)
