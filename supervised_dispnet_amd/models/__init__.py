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


for _n in ("Disp_res", "Disp_vgg", "Disp_vgg_feature", "FCRN", "deeplab_depth", "Disp_res_101", "DORN",
           "res50_aspp", "Disp_res_18", "PoseExpNet", "Disp_vgg_BN_DORN", "Disp_res_50", "monodepth2"):
    if _n not in globals():
        globals()[_n] = _out_of_scope(_n)
