"""Mirror of the reference's `networks` package for the hot path (reference networks/__init__.py:1-5): the monodepth2-style
encoders / depth decoder on the HIP engine.  PoseDecoder / PoseCNN are dead code in the reference (imported, never
instantiated) and are out of scope."""
from .depth_decoder import DepthDecoder
from .resnet_encoder import ResnetEncoder
from .vgg_encoder import vggEncoder


def _out_of_scope(name):
    class _Missing(object):
        def __init__(self, *a, **k):
            raise NotImplementedError("networks.%s is never instantiated by the reference (dead code) and is outside this build's scope" % name)
    _Missing.__name__ = name
    return _Missing


PoseDecoder = _out_of_scope("PoseDecoder")
PoseCNN = _out_of_scope("PoseCNN")


def _no_replication(self):
    """See supervised_dispnet_amd.models._no_replication: nn.DataParallel over ONE device works, replicas over several do not."""
    from ..models import _no_replication as f
    return f(self)


for _c in (DepthDecoder, ResnetEncoder, vggEncoder):
    _c._replicate_for_data_parallel = _no_replication
