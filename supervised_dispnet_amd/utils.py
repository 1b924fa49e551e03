"""Mirror of the hot-path pieces of the reference's utils.py: the SID discretisation used by the DORN head / loss."""
import torch

from . import _lib
from .engine import _stream, require_cuda


def _beta(dataset):
    if dataset == 'kitti':
        return 80.999
    if dataset in ('nyu', 'NYU'):
        return 10.999
    raise ValueError("undefined dataset %r" % (dataset,))


@torch.no_grad()
def get_labels_sid(depth, ordinal_c=71.0, dataset='kitti'):
    """reference utils.py:147-175: int32 labels = int(K * log(depth + 0.999) / log(beta)) (truncation toward zero)."""
    require_cuda(depth, "depth")
    if dataset == 'NYU':
        raise UnboundLocalError("get_labels_sid only knows 'kitti' and 'nyu' (reference utils.py:153-158)")
    d = depth.contiguous().float()
    out = torch.empty(d.shape, dtype=torch.int32, device=d.device)
    _lib.call("dn_sid_labels", d.data_ptr(), d.numel(), float(ordinal_c), _beta(dataset), out.data_ptr(), _stream())
    return out


@torch.no_grad()
def get_depth_sid(labels, ordinal_c=71.0, dataset='kitti'):
    """reference utils.py:106-133: 0.5 * (beta^(l/K) + beta^((l+1)/K)) - 0.999."""
    require_cuda(labels, "labels")
    l = labels.contiguous().to(torch.int64)
    out = torch.empty(l.shape, dtype=torch.float32, device=l.device)
    _lib.call("dn_sid_depth", l.data_ptr(), l.numel(), float(ordinal_c), _beta(dataset), out.data_ptr(), _stream())
    return out
