"""Mirror of the hot-path pieces of the reference's utils.py: the SID discretisation used by the DORN head / loss."""
import os

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


def load_model(pretrained_model, weights_folder):
    """Monodepth2 weight loader of the reference (train.py:726-751, test_disp.py:85-86): for n in ("encoder", "depth") read
    <weights_folder>/<n>.pth, keep the keys the module has, load.  Fails loudly when the folder or a file is missing."""
    folder = os.path.expanduser(weights_folder) if weights_folder else weights_folder
    if not folder or not os.path.isdir(folder):
        raise FileNotFoundError("Cannot find folder {}".format(folder))
    print("loading model from folder {}".format(folder))
    for n in ("encoder", "depth"):
        print("Loading {} weights...".format(n))
        path = os.path.join(folder, "{}.pth".format(n))
        model_dict = pretrained_model[n].state_dict()
        pretrained_dict = torch.load(path, map_location="cpu")
        model_dict.update({k: v for k, v in pretrained_dict.items() if k in model_dict})
        pretrained_model[n].load_state_dict(model_dict)
