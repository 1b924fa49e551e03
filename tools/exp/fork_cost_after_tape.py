"""How long a fork() of the training process takes before / after a launch tape has been recorded (DataLoader workers are forked at every epoch)."""
import os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
import bench
from supervised_dispnet_amd import engine, models
import supervised_dispnet_amd.loss_functions as LF
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.graph import TapedStep, backward
from supervised_dispnet_amd.optim import FusedAdam


def fork_ms(n=3):
    ts = []
    for _ in range(n):
        t0 = time.time()
        pid = os.fork()
        if pid == 0:
            os._exit(0)
        os.waitpid(pid, 0)
        ts.append((time.time() - t0) * 1e3)
    return min(ts)


def vmas():
    return sum(1 for _ in open("/proc/self/maps"))


print("import only: fork %.1f ms, %d mappings" % (fork_ms(), vmas()))
dev = torch.device("cuda:0")
net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False).to(dev).train()
opt = FusedAdam(net.parameters(), lr=1e-4)
img, gt = bench.synthetic_batch(4, 64, 96, dev, 0)


def step():
    depth = [reciprocal(d) for d in net(img)]
    loss = LF.l1_loss(gt, depth, "kitti")
    opt.zero_grad()
    backward(loss)
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
print("after 3 eager steps: fork %.1f ms, %d mappings" % (fork_ms(), vmas()))
ts = TapedStep(step, optimizer=opt, warmup=1).capture()
for _ in range(3):
    ts()
torch.cuda.synchronize()
print("after taping + 3 replays: fork %.1f ms, %d mappings" % (fork_ms(), vmas()))
t0 = time.time()
dl = torch.utils.data.DataLoader(list(range(64)), batch_size=4, num_workers=4, pin_memory=True)
for _ in dl:
    break
print("first batch of a 4-worker DataLoader: %.2f s" % (time.time() - t0))
