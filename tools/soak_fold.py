#!/usr/bin/env python3
"""Soak of the last-arrival epilogues (csrc/dn_fold.h): N training steps of Disp_vgg_BN with the BatchNorm statistics / BatchNorm-backward
sums finished inside the Winograd kernels (agent-scope relaxed atomics across XCDs, no fence) against the same N steps with the
stand-alone kernels, from the same initial state on changing batches.  The two are bit-identical by construction, so ONE stale read in
~N x 15 folded reductions shows up as diverging parameters.  Also through the launch tape (the form the benchmark times).
usage: python tools/soak_fold.py [batch] [steps]"""
import copy, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import bench
import supervised_dispnet_amd.loss_functions as LF
import supervised_dispnet_amd.models as models
from supervised_dispnet_amd import engine
from supervised_dispnet_amd.functional import reciprocal
from supervised_dispnet_amd.graph import TapedStep, backward
from supervised_dispnet_amd.optim import FusedAdam

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
batches = [bench.synthetic_batch(batch, 128, 416, dev, s) for s in range(8)]
torch.manual_seed(0)
ref = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
bench._quiet_init(ref)
sd0 = copy.deepcopy({k: v.detach().cpu().clone() for k, v in ref.state_dict().items()})


def run(fold, taped):
    engine.FOLD_FINALIZE = fold
    net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
    net.load_state_dict(sd0)
    net.to(dev).train()
    opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
    opt.capturable(True)
    img, gt = batches[0][0].clone(), batches[0][1].clone()

    def step():
        depth = [reciprocal(d) for d in net(img)]
        loss = LF.l1_loss(gt, depth, "kitti")
        opt.zero_grad()
        backward(loss)
        opt.step()
        return loss

    ts = TapedStep(step, optimizer=opt, warmup=0, static_inputs=(img, gt)) if taped else None
    last = None
    for s in range(steps):
        b = batches[s % len(batches)]
        img.copy_(b[0])
        gt.copy_(b[1])
        last = ts() if ts is not None else step()
    torch.cuda.synchronize()
    out = (float(last.item()), opt.arena.flat_p.clone(), {k: v.clone() for k, v in net.state_dict().items() if "running" in k})
    if ts is not None:
        ts.close()
    return out


ok = True
for taped in (False, True):
    a = run(True, taped)
    b = run(False, taped)
    same = a[0] == b[0] and torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    ok = ok and same
    print("batch %d, %d steps, %s: folded vs stand-alone reductions %s (final loss %.6f / %.6f, max |dp| %.3g)" % (
        batch, steps, "launch tape" if taped else "eager", "BIT-IDENTICAL" if same else "DIFFER", a[0], b[0], float((a[1] - b[1]).abs().max())))
sys.exit(0 if ok else 1)
