#!/usr/bin/env python3
"""In-kernel timestamps of dn::wino_conv_kernel (debug build mode DN_WINO_DBG=4): prologue / main loop / epilogue cycles."""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch, torch.nn as nn
from supervised_dispnet_amd import engine
dev = torch.device("cuda:0")
buf = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
BN_STATS = len(sys.argv) > 2 and sys.argv[2] == "bn"
os.environ["DN_WINO_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "4"
os.environ["DN_WINO_DBGPTR"] = hex(buf.data_ptr())
for cin, cout, H, W in [(64, 64, 128, 416), (128, 128, 64, 208), (256, 256, 32, 104), (512, 512, 16, 52)]:
    mod = nn.Conv2d(cin, cout, 3, 1, 1).to(dev)
    layer = engine.ConvLayer(mod)
    x = engine.Act(torch.randn(32, H, W, cin, device=dev), 32, H, W, cin)
    for _ in range(3):
        engine.conv_forward(layer, [engine.Piece(x)], bn_stats=BN_STATS)
    torch.cuda.synchronize()
    from supervised_dispnet_amd import _lib
    k8 = 'conv8' in _lib.load().dn_last_kernel().decode()
    bt = 64 if (os.environ.get('DN_WINO_MTW') == '2' or k8) else 32
    nblk = ((32 * H * W // 4 + bt - 1) // bt) * (cout // 64)
    t = buf[: ((nblk + 7) // 8 * 8) * 8].view(-1, 8).cpu()
    t = t[t[:, 3] > 0]
    if len(t) == 0:
        print('cin%d cout%d: this ablation is not compiled for the kernel that takes the layer' % (cin, cout)); buf.zero_(); continue
    pro = (t[:, 1] - t[:, 0]).float().mean().item(); loop = (t[:, 2] - t[:, 1]).float().mean().item(); epi = (t[:, 3] - t[:, 2]).float().mean().item()
    nch = cin // 16
    mfma2 = 3072 if engine.compute_mode() == "f32x3" else 8192      # 2 blocks x (48 x 32 | 64 x 64) cycles per 16-channel chunk
    print(("[8-wave] " if k8 else "") + "cin%d cout%d %dx%d: blocks %d prologue %.0f loop %.0f (%.0f per chunk; matrix-instruction time of the two co-resident blocks %d) epilogue %.0f  [clock64 ticks, compute %s]" % (cin, cout, H, W, len(t), pro, loop, loop / nch, mfma2, epi, engine.compute_mode()))
    span = (t[:, 3].max() - t[:, 0].min()).item()
    print("   kernel span %d ticks" % span)
    ex = (t[:, 4] - t[:, 2]).float().mean().item(); st = (t[:, 5] - t[:, 4]).float().mean().item(); so = (t[:, 3] - t[:, 5]).float().mean().item()
    buf.zero_()
    print("   epilogue: output transform + cross-wave exchange %.0f | statistics %.0f | bias/act/stores %.0f" % (ex, st, so))
