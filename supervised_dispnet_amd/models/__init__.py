"""Mirror of the reference's `models` package for the hot path (reference models/__init__.py:1-15).

In-scope nets (incl. FCRN's up-projection blocks and the dilated ASPP nets) are real; the
one remaining export, the full `DORN` backbone (models/DORN.py + Dorn_backbone.py: average-pool + Linear full-image encoder, a
five-way concat, multi-channel bilinear upsampling; train.py:257-258 constructs it with a `datasets` keyword its __init__ does not
take, i.e. the reference's own `--network DORN` fails), raises a clear error rather than silently running something else.
"""
from .DispNetS import DispNetS
from .Disp_vgg_BN import Disp_vgg_BN
from .Disp_vgg_BN_DORN import Disp_vgg_BN_DORN
from .Disp_res_50 import Disp_res_50
from .Disp_res_18 import Disp_res_18
from .Disp_res import Disp_res
from .Disp_res_101 import Disp_res_101
from .Disp_vgg import Disp_vgg
from .Disp_vgg_feature import Disp_vgg_feature
from .PoseExpNet import PoseExpNet
from .monodepth2 import monodepth2
from .FCRN import FCRN
from .ASPP import deeplab_depth, res50_aspp


def _out_of_scope(name):
    class _Missing(object):
        def __init__(self, *a, **k):
            raise NotImplementedError(
                "models.%s is outside the MI355X hot-path scope of this build (SURVEY.md section 8); "
                "use the reference implementation for it" % name)
    _Missing.__name__ = name
    return _Missing


# the one model of the reference's `models/__init__.py` this build does not carry: the full DORN backbone
DORN = _out_of_scope("DORN")


def _no_replication(self):
    """nn.DataParallel(net) on ONE device calls the wrapped module directly (torch/nn/parallel/data_parallel.py: `if len(self.device_ids)
    == 1: return self.module(...)`), so the reference's `disp_net = torch.nn.DataParallel(disp_net)` / `disp_net.module.state_dict()`
    (train.py:316-317,378) keep working on a one-GPU box.  Beyond one device DataParallel replicates the module per forward: the
    engine's per-module runtime state (packed-weight caches, the launch schedule, arena-backed gradients) is per process, not per
    replica -- data parallelism here is one process per GPU (supervised_dispnet_amd.distributed.GradReducer)."""
    raise RuntimeError(
        "supervised_dispnet_amd models cannot be replicated by nn.DataParallel across several devices: run one process per GPU "
        "(python -m torch.distributed.run --nproc-per-node N train.py ...; gradients are summed by distributed.GradReducer over RCCL). "
        "nn.DataParallel(net, device_ids=[one device]) works.")


for _n, _c in list(globals().items()):
    if isinstance(_c, type) and hasattr(_c, "_hip_forward"):
        _c._replicate_for_data_parallel = _no_replication
