import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
import bench
from supervised_dispnet_amd import engine, models
import supervised_dispnet_amd.loss_functions as LF
from supervised_dispnet_amd.functional import reciprocal
fuse, borrow = int(sys.argv[1]), int(sys.argv[2])
LF.FUSE_MASKED_FWD = bool(fuse)
engine.BORROW_SEED_GRADS = bool(borrow)
dev = torch.device("cuda:0")
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
img, gt = bench.synthetic_batch(4, 128, 416, dev, 0)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    depth = [reciprocal(d) for d in net(img)]
    loss = LF.Multiscale_L1_loss(gt, depth)
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print("fuse %d borrow %d step %d: %.3f s loss %.5f" % (fuse, borrow, i, time.time() - t0, float(loss)), flush=True)
