"""The library-owned RCCL path on the one GPU this pool gives a test: a single-rank communicator (ncclCommInitRank with nranks = 1)
exercises the ctypes binding, the library-owned stream and the event fences against BOTH compute streams; the result of three training
steps through GradReducer(comm="rccl") must be bit-identical to the same steps without any reducer (the sum over one rank is the
identity, 1/world = 1), with the weight gradients on their side stream in both runs.  Multi-rank behaviour is covered on CPU by
tests/test_distributed_cpu.py (gloo) and by construction (same bucket walk, same fences)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import bench  # noqa: E402
import supervised_dispnet_amd.loss_functions as LF  # noqa: E402
import supervised_dispnet_amd.models as models  # noqa: E402
from supervised_dispnet_amd import engine, rccl  # noqa: E402
from supervised_dispnet_amd.distributed import GradReducer  # noqa: E402
from supervised_dispnet_amd.functional import reciprocal  # noqa: E402
from supervised_dispnet_amd.optim import FusedAdam  # noqa: E402

DEV = torch.device("cuda:0")


def test_single_rank_communicator_allreduce_is_identity_and_fenced():
    comm = rccl.Communicator(device=DEV)
    assert comm.world == 1 and comm.rank == 0 and comm.stream != torch.cuda.current_stream()
    side = torch.cuda.Stream()
    x = torch.zeros(1 << 22, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):
            x += 1.0                                     # queued work the collective must wait for
    comm.all_reduce_sum_(x, [torch.cuda.current_stream(), side])
    comm.join()
    assert float(x.sum().item()) == 20.0 * x.numel()
    with pytest.raises(ValueError):
        comm.all_reduce_sum_(x.double(), [])
    comm.destroy()


def test_rccl_reducer_training_steps_bitwise_equal_no_reducer():
    assert engine.wgrad_stream_enabled()
    img, gt = bench.synthetic_batch(4, 128, 416, DEV, 0)
    out = {}
    for mode in ("none", "rccl"):
        torch.manual_seed(0)
        net = models.Disp_vgg_BN(datasets="kitti", with_classifier=False)
        bench._quiet_init(net)
        net.to(DEV).train()
        opt = FusedAdam(net._hot_parameters(), lr=1e-4, production_order=net._grad_production_order())
        red = GradReducer(opt.arena, comm="rccl") if mode == "rccl" else None
        engine.GradSink.reducer = red
        try:
            if red is not None:
                assert len(red.buckets) >= 3 and red.world == 1
            for _ in range(3):
                depth = [reciprocal(d) for d in net(img)]
                loss = LF.l1_loss(gt, depth, "kitti")
                opt.zero_grad()
                loss.backward()
                opt.step(grad_scale=red.finish() if red is not None else 1.0)
            torch.cuda.synchronize()
        finally:
            engine.GradSink.reducer = None
        out[mode] = (opt.arena.flat_p.clone(), opt.arena.flat_g.clone(), loss.item())
        if red is not None:
            red.comm.destroy()
    assert out["none"][2] == out["rccl"][2]
    assert torch.equal(out["none"][1], out["rccl"][1]) and torch.equal(out["none"][0], out["rccl"][0])
