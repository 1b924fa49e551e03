"""Mirror of the reference's loss_functions.py on HIP kernels (same function names, argument order and meaning).

Every function below launches kernels of libdispnet_hip.so through the C ABI; there is no PyTorch fallback (CPU tensors
raise).  PyTorch only provides the tensors, the stream and the autograd hand-off.
"""
import torch

from . import _lib
from .engine import _stream, hbm_call, require_cuda

_MAX_DEPTH = {"kitti": 80.0, "nyu": 10.0}


def _max_depth(datasets):
    if datasets not in _MAX_DEPTH:
        raise ValueError("undefined datasets %r" % (datasets,))   # the reference does `raise "undefined datasets"` (a TypeError)
    return _MAX_DEPTH[datasets]


def _f32c(t, what):
    require_cuda(t, what)
    t = t.detach() if not t.requires_grad else t
    return t.contiguous().float()


# ------------------------------------------------------------------------------------------------ masked depth losses
FUSE_MASKED_FWD = True          # A/B + the bitwise test: False = the three launches (+ the fill of the statistics rows)
_FOLD_COUNTERS = {}             # device -> [zeroed int32 tensor, next slot]: last-arrival counters of the fused forward


def _fold_counter(dev):
    """Address of a zero int32 for one fused launch: 64 slots per device handed out round-robin, so that launches that overlap on the
    device (other streams) count in different slots; every launch leaves its slot zero."""
    from . import engine
    rec = engine.TAPE
    if rec is not None and not rec.get("paused"):
        # a launch tape bakes the address in for good: it gets counters of its own (kept alive with the tape), never shared with the eager
        # launches that keep cycling through the 64 slots below -- validation, a second model's loss -- and may overlap with a replay
        ent = rec.get("fold_counters")
        if ent is None:
            with engine.outside_tape_pool():
                ent = rec["fold_counters"] = [torch.zeros(256, dtype=torch.int32, device=dev), 0]
            rec["keep"].append(ent[0])
        if ent[1] >= 256:
            raise RuntimeError("more than 256 fused loss launches in one recorded step")
        ent[1] += 1
        return ent[0].data_ptr() + 4 * (ent[1] - 1)
    ent = _FOLD_COUNTERS.get(dev)
    if ent is None:
        from .engine import outside_tape_pool
        with outside_tape_pool():                    # persistent: not from a launch tape's private pool
            ent = _FOLD_COUNTERS[dev] = [torch.zeros(64, dtype=torch.int32, device=dev), 0]
    ent[1] = (ent[1] + 1) % 64
    return ent[0].data_ptr() + 4 * ent[1]


class _MaskedLoss(torch.autograd.Function):
    """Sum of terms of the masked-loss family (C entry points dn_masked_loss_fwd/_bwd) as ONE autograd node.
    Term i: `groups[i]` runs of equal length over (gts[i], preds[i]), each with its own mask mean; contributes
    weights[i] * mean over its groups.  The kernels accumulate into one device scalar -- no host arithmetic.

    `sync` (a world size > 1): the terms are WHOLE-BATCH means (Multiscale_*: loss_functions.py:232-237 normalises by the valid
    count of the gathered batch) computed one process per GPU: every rank reduces its own pixels to (sum, count, max) statistics,
    the statistics are all-reduced (distributed.exchange_loss_stats) and the division happens afterwards, so the value is the
    reference's whole-batch number on every rank.  The backward then yields world * d(global loss)/d(local prediction): the
    data-parallel optimizer averages the ranks' parameter gradients (1/world folded into Adam), which restores the sum."""

    @staticmethod
    def forward(ctx, max_depth, kind, groups, weights, sync, gts, *preds):
        lib = _lib.load()
        dev = preds[0].device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        saved, cfg, work = [], [], []
        all_stats = torch.empty((sum(groups[:len(preds)]), _lib.LOSS_STATS), dtype=torch.float32, device=dev)
        # one launch per term (dn_masked_loss_fwd_fused: the last-arriving block reduces and divides) where the term allows it; the rows it
        # writes need no zero fill (columns 0-5 are all the backward reads)
        fused = FUSE_MASKED_FWD and sync <= 1 and kind != _lib.LOSS_BERHU and max(groups[:len(preds)]) <= 256
        if not fused:
            _lib.call("dn_fill", all_stats.data_ptr(), 0.0, all_stats.numel(), _stream())
        row = 0
        for i, (gt, pred) in enumerate(zip(gts, preds)):
            require_cuda(pred, "predicted depth")
            require_cuda(gt, "ground-truth depth")
            gtc, pc = gt.contiguous().float(), pred.contiguous()
            if gtc.numel() != pc.numel():
                raise ValueError("gt %s and prediction %s sizes differ" % (tuple(gt.shape), tuple(pred.shape)))
            g = groups[i]
            pixels = pc.numel() // g
            nbytes = lib.dn_masked_loss_workspace_bytes(g, pixels)
            ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
            stats = all_stats[row:row + g]
            row += g
            if sync > 1:
                _lib.call("dn_masked_loss_stats", gtc.data_ptr(), pc.data_ptr(), g, pixels, max_depth, kind, 0, stats.data_ptr(),
                          ws.data_ptr(), nbytes, _stream())
                work.append((gtc, pc, g, pixels, stats, ws, nbytes))
            elif fused:
                hbm_call("dn::masked_stats_fold_kernel", 8 * g * pixels, "dn_masked_loss_fwd_fused", gtc.data_ptr(), pc.data_ptr(), g, pixels, max_depth, kind,
                         weights[i], 0 if i == 0 else 1, stats.data_ptr(), ws.data_ptr(), nbytes, loss.data_ptr(), _fold_counter(dev), _stream())
            else:
                hbm_call("dn::masked_stats_kernel", 8 * g * pixels, "dn_masked_loss_fwd", gtc.data_ptr(), pc.data_ptr(), g, pixels, max_depth, kind, weights[i], 0 if i == 0 else 1,
                          stats.data_ptr(), ws.data_ptr(), nbytes, loss.data_ptr(), _stream())
            saved += [gtc, pc, stats]
            cfg.append((g, pixels, weights[i], pred.shape))
        if sync > 1:
            from .distributed import exchange_loss_stats
            exchange_loss_stats(all_stats, sum_cols=(0, 1, 2), max_cols=(3,))
            if kind == _lib.LOSS_BERHU:
                for gtc, pc, g, pixels, stats, ws, nbytes in work:
                    _lib.call("dn_masked_loss_stats", gtc.data_ptr(), pc.data_ptr(), g, pixels, max_depth, kind, 1, stats.data_ptr(),
                              ws.data_ptr(), nbytes, _stream())
                exchange_loss_stats(all_stats, sum_cols=(0, 2, 4), max_cols=())
            for i, (gtc, pc, g, pixels, stats, ws, nbytes) in enumerate(work):
                _lib.call("dn_masked_loss_finalize", stats.data_ptr(), g, kind, weights[i], 0 if i == 0 else 1, loss.data_ptr(), _stream())
        ctx.save_for_backward(*saved)
        ctx.cfg = (max_depth, kind, cfg, float(max(sync, 1)))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        max_depth, kind, cfg, up = ctx.cfg
        dl = dloss.contiguous().float()
        grads = []
        for i, (g, pixels, weight, shape) in enumerate(cfg):
            gtc, pc, stats = ctx.saved_tensors[3 * i:3 * i + 3]
            dpred = torch.empty(shape, dtype=torch.float32, device=pc.device)
            hbm_call("dn::masked_loss_bwd_kernel", 12 * g * pixels, "dn_masked_loss_bwd", gtc.data_ptr(), pc.data_ptr(), stats.data_ptr(), dl.data_ptr(), g, pixels, max_depth, kind,
                      weight * up, dpred.data_ptr(), _stream())
            grads.append(dpred)
        return (None, None, None, None, None, None) + tuple(grads)


def _scale0(depth):
    """depth[0][:,0] of the reference, handed over as the whole [B,1,H,W] tensor (same memory, no select node)."""
    d0 = depth[0]
    if d0.dim() != 4 or d0.shape[1] != 1:
        raise ValueError("expected depth[0] of shape [B,1,H,W], got %s" % (tuple(d0.shape),))
    return d0


def _per_sample(gt_depth, depth, datasets, kind):
    d0 = _scale0(depth)
    # per-sample means divided by the batch size: exactly shardable over ranks, no exchange needed
    return _MaskedLoss.apply(_max_depth(datasets), kind, [d0.shape[0]], [1.0], 0, [gt_depth], d0)


def l1_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:104-129 (uses scale 0 only: depth[0][:,0])."""
    return _per_sample(gt_depth, depth, datasets, _lib.LOSS_L1)


def l2_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:77-102; its 'nyu' branch is an L1 (line 97) and is reproduced as such."""
    _max_depth(datasets)
    return _per_sample(gt_depth, depth, datasets, _lib.LOSS_L1 if datasets == "nyu" else _lib.LOSS_L2)


def berhu_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:131-161.  The reference's 'nyu' branch lost its for-loop (NameError at :149); here it
    loops like the 'kitti' branch with max depth 10."""
    return _per_sample(gt_depth, depth, datasets, _lib.LOSS_BERHU)


def Scale_invariant_loss(gt_depth, depth, datasets):
    """reference loss_functions.py:163-189."""
    return _per_sample(gt_depth, depth, datasets, _lib.LOSS_SCALE_INV)


# ---------------------------------------------------------------------------------------------------------- pyramids
_PYR_MODE = {"max": 0, "avg": 1, "bilinear": 2}


@torch.no_grad()
def _pyramid(image, mode):
    require_cuda(image, "ground-truth depth")
    cur = image.contiguous().float()
    pyr = [cur]
    for _ in range(3):
        n, h, w = cur.shape
        nxt = torch.empty((n, h // 2, w // 2), dtype=torch.float32, device=cur.device)
        _lib.call("dn_pyramid_down2", cur.data_ptr(), n, h, w, mode, nxt.data_ptr(), _stream())
        pyr.append(nxt)
        cur = nxt
    return pyr


def generate_max_pyramid(image):
    """reference loss_functions.py:191-196"""
    return _pyramid(image, 0)


def generate_avg_pyramid(image):
    """reference loss_functions.py:198-203"""
    return _pyramid(image, 1)


def generate_bilinear_pyramid(image):
    """reference loss_functions.py:205-215"""
    return _pyramid(image, 2)


def _multiscale(gt_list, preds, kind):
    """One mask over the WHOLE batch per scale, weight 1/2^i, max depth hard-coded 80 (reference :229-238)."""
    n = min(len(gt_list), len(preds))
    from .distributed import data_parallel_world
    return _MaskedLoss.apply(80.0, kind, [1] * n, [1.0 / (2 ** i) for i in range(n)], data_parallel_world(), list(gt_list[:n]), *preds[:n])


def Multiscale_L1_loss(gt_depth, depth, pool_type="bilinear"):
    """reference loss_functions.py:217-238"""
    if pool_type not in _PYR_MODE:
        raise ValueError("undefined pool type")
    return _multiscale(_pyramid(gt_depth, _PYR_MODE[pool_type]), depth, _lib.LOSS_L1)


class _UpsampleInt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, mode):
        require_cuda(x, "prediction")
        xc = x.contiguous()
        b, c, h, w = xc.shape
        out = torch.empty((b, c, h * scale, w * scale), dtype=torch.float32, device=xc.device)
        _lib.call("dn_upsample_int_fwd", xc.data_ptr(), b * c, h, w, scale, mode, out.data_ptr(), _stream())
        ctx.cfg = (b * c, h, w, scale, mode, xc.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, h, w, scale, mode, shape = ctx.cfg
        d = dout.contiguous()
        dx = torch.empty(shape, dtype=torch.float32, device=d.device)
        _lib.call("dn_upsample_int_bwd", d.data_ptr(), n, h, w, scale, mode, dx.data_ptr(), _stream())
        return dx, None, None


def Multiscale_FULL_L1_loss(gt_depth, depth, pool_type="bilinear"):
    """reference loss_functions.py:240-257: predictions upsampled to full resolution with F.upsample(mode=pool_type)."""
    modes = {"nearest": 0, "bilinear": 1}
    if pool_type not in modes:
        raise NotImplementedError("F.upsample mode %r" % (pool_type,))
    ups = [_UpsampleInt.apply(d, 2 ** i, modes[pool_type]) for i, d in enumerate(depth)]
    return _multiscale([gt_depth] * len(depth), ups, _lib.LOSS_L1)


def Multiscale_L2_loss(gt_depth, depth):
    """reference loss_functions.py:259-273"""
    return _multiscale(_pyramid(gt_depth, 2), depth, _lib.LOSS_L2)


def Multiscale_berhu_loss(gt_depth, depth):
    """reference loss_functions.py:275-296"""
    return _multiscale(_pyramid(gt_depth, 2), depth, _lib.LOSS_BERHU)


def Multiscale_scale_inv_loss(gt_depth, depth):
    """reference loss_functions.py:298-315"""
    return _multiscale(_pyramid(gt_depth, 2), depth, _lib.LOSS_SCALE_INV)


# -------------------------------------------------------------------------------------------------------- photometric
_ROT = {"euler": 0, "quat": 1}
_PAD = {"zeros": 0, "border": 1}


class _Photometric(torch.autograd.Function):
    """photometric_reconstruction_loss (reference loss_functions.py:317-354) as ONE autograd node over all scales and
    reference images: per (scale, ref) one fused warp+|diff| launch forward and one fused backward launch."""

    @staticmethod
    def forward(ctx, tgt_img, intrinsics, intrinsics_inv, pose, rot, pad, align, n_ref, n_scale, has_mask, *rest):
        refs = rest[:n_ref]
        depths = rest[n_ref:n_ref + n_scale]
        masks = rest[n_ref + n_scale:] if has_mask else [None] * n_scale
        dev = tgt_img.device
        for t in (tgt_img, intrinsics, intrinsics_inv, pose) + tuple(refs) + tuple(depths):
            require_cuda(t, "photometric loss input")
        tgt = tgt_img.contiguous().float()
        refs = [r.contiguous().float() for r in refs]
        K, Kinv = intrinsics.contiguous().float(), intrinsics_inv.contiguous().float()
        posec = pose.contiguous().float()
        B, _, H, W = tgt.shape
        if posec.shape[1] != n_ref:
            raise AssertionError("pose.size(1) != len(ref_imgs)")
        lib = _lib.load()
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        saved = []
        first = True
        for d, m in zip(depths, masks):
            dc = d.contiguous().float()
            b, _, h, w = dc.shape
            if m is not None and tuple(m.shape[2:]) != (h, w):
                raise AssertionError("explainability mask and depth sizes differ")
            f = H // h
            down = float(H) / h
            if f == 1:
                tgt_s, refs_s = tgt, refs
            else:
                tgt_s = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
                hbm_call("dn::area_down_kernel", 12 * B * (H * W + h * w), "dn_area_down", tgt.data_ptr(), B * 3, H, W, f, tgt_s.data_ptr(), _stream())
                refs_s = []
                for r in refs:
                    rs = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
                    hbm_call("dn::area_down_kernel", 12 * B * (H * W + h * w), "dn_area_down", r.data_ptr(), B * 3, H, W, f, rs.data_ptr(), _stream())
                    refs_s.append(rs)
            mc = m.contiguous().float() if m is not None else None
            nb = lib.dn_warp_blocks(h, w)
            per_ref = []
            for i, rs in enumerate(refs_s):
                proj = torch.empty((B, 12), dtype=torch.float32, device=dev)
                kinv_s = torch.empty((B, 9), dtype=torch.float32, device=dev)
                _lib.call("dn_pose_proj_fwd", posec.data_ptr() + 4 * 6 * i, 6 * n_ref, K.data_ptr(), Kinv.data_ptr(), B, rot, down,
                          proj.data_ptr(), kinv_s.data_ptr(), _stream())
                partial = torch.empty(B * nb, dtype=torch.float32, device=dev)
                mptr = (mc.data_ptr() + 4 * i * h * w) if mc is not None else None
                msb = mc.shape[1] * h * w if mc is not None else 0
                hbm_call("dn::warp_fwd_kernel", 4 * B * h * w * (7 + (1 if mc is not None else 0)), "dn_photometric_fwd", tgt_s.data_ptr(), rs.data_ptr(), dc.data_ptr(), proj.data_ptr(), kinv_s.data_ptr(),
                          mptr, msb, B, h, w, pad, align, 1.0, 0 if first else 1, partial.data_ptr(), loss.data_ptr(), _stream())
                first = False
                per_ref.append((rs, proj, kinv_s))
            saved.append((dc, mc, tgt_s, per_ref, down, h, w))
        ctx.saved = saved
        ctx.cfg = (B, rot, pad, align, n_ref, n_scale, has_mask, K, posec, [d.shape for d in depths],
                   [m.shape if m is not None else None for m in masks])
        return loss

    @staticmethod
    def backward(ctx, dloss):
        B, rot, pad, align, n_ref, n_scale, has_mask, K, posec, dshapes, mshapes = ctx.cfg
        dev = posec.device
        dl = dloss.contiguous().float()
        lib = _lib.load()
        dpose = torch.zeros_like(posec)
        ddepths, dmasks = [], []
        first_pose = [True] * n_ref
        for (dc, mc, tgt_s, per_ref, down, h, w), dshape, mshape in zip(ctx.saved, dshapes, mshapes):
            nb = lib.dn_warp_blocks(h, w)
            dd = torch.empty(dshape, dtype=torch.float32, device=dev)
            dm = torch.zeros(mshape, dtype=torch.float32, device=dev) if mshape is not None else None
            for i, (rs, proj, kinv_s) in enumerate(per_ref):
                dpp = torch.empty((B, nb, 12), dtype=torch.float32, device=dev)
                mptr = (mc.data_ptr() + 4 * i * h * w) if mc is not None else None
                msb = mc.shape[1] * h * w if mc is not None else 0
                dmptr = (dm.data_ptr() + 4 * i * h * w) if dm is not None else None
                hbm_call("dn::warp_bwd_kernel", 4 * B * h * w * (8 + (2 if mc is not None else 0)), "dn_photometric_bwd", tgt_s.data_ptr(), rs.data_ptr(), dc.data_ptr(), proj.data_ptr(), kinv_s.data_ptr(),
                          mptr, msb, B, h, w, pad, align, 1.0, dl.data_ptr(), dd.data_ptr(), 0 if i == 0 else 1, dpp.data_ptr(),
                          dmptr, msb, _stream())
                _lib.call("dn_pose_proj_bwd", posec.data_ptr() + 4 * 6 * i, 6 * n_ref, K.data_ptr(), B, rot, down, dpp.data_ptr(), nb,
                          dpose.data_ptr() + 4 * 6 * i, 6 * n_ref, 0 if first_pose[i] else 1, _stream())
                first_pose[i] = False
            ddepths.append(dd)
            dmasks.append(dm)
        ctx.saved = None
        grads = [None, None, None, dpose, None, None, None, None, None, None] + [None] * n_ref + ddepths
        if has_mask:
            grads += dmasks
        return tuple(grads)


def photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask, pose,
                                    rotation_mode='euler', padding_mode='zeros', align_corners=False):
    """reference loss_functions.py:317-354.  `align_corners` is F.grid_sample's flag: the reference passes none, which means
    False on torch >= 1.3 (what its code yields today) and meant True on its pinned torch 1.0.1."""
    if type(explainability_mask) not in [tuple, list]:
        explainability_mask = [explainability_mask]
    if type(depth) not in [list, tuple]:
        depth = [depth]
    n = min(len(depth), len(explainability_mask))          # zip() semantics of the reference
    depth, masks = list(depth[:n]), list(explainability_mask[:n])
    has_mask = all(m is not None for m in masks)
    if not has_mask and any(m is not None for m in masks):
        raise ValueError("explainability masks must be given for every scale or for none")
    args = list(ref_imgs) + depth + (masks if has_mask else [])
    return _Photometric.apply(tgt_img, intrinsics, intrinsics_inv, pose, _ROT[rotation_mode], _PAD[padding_mode],
                              1 if align_corners else 0, len(ref_imgs), n, has_mask, *args)


class _Explainability(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *masks):
        dev = masks[0].device
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        saved = []
        lib = _lib.load()
        for i, m in enumerate(masks):
            require_cuda(m, "explainability mask")
            mc = m.contiguous().float()
            partial = torch.empty(lib.dn_reduce1d_blocks(mc.numel()), dtype=torch.float32, device=dev)
            _lib.call("dn_explainability_fwd", mc.data_ptr(), mc.numel(), 1.0, 0 if i == 0 else 1, partial.data_ptr(), loss.data_ptr(), _stream())
            saved.append(mc)
        ctx.save_for_backward(*saved)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dl = dloss.contiguous().float()
        out = []
        for mc in ctx.saved_tensors:
            g = torch.empty_like(mc)
            _lib.call("dn_explainability_bwd", mc.data_ptr(), dl.data_ptr(), mc.numel(), g.data_ptr(), _stream())
            out.append(g)
        return tuple(out)


def explainability_loss(mask):
    """reference loss_functions.py:357-364: sum over scales of binary_cross_entropy(mask, ones)."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    return _Explainability.apply(*mask)


# --------------------------------------------------------------------------------------------------------- smoothness
class _Smooth(torch.autograd.Function):
    """loss_functions.py:367-386 for a list of [B,C,H,W] maps (each channel treated as its own map)."""

    @staticmethod
    def forward(ctx, weight_decay, *maps):
        dev = maps[0].device
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        saved, cfg, w = [], [], 1.0
        lib = _lib.load()
        for m in maps:
            require_cuda(m, "prediction map")
            mc = m.contiguous()
            b, c, h, wd = mc.shape
            nb = lib.dn_smooth_blocks(b * c, h, wd)
            partial = torch.empty((nb, 4), dtype=torch.float32, device=dev)
            hbm_call("dn::smooth2_fwd_kernel", 4 * b * c * h * wd, "dn_smooth2_fwd", mc.data_ptr(), b * c, h, wd, w, partial.data_ptr(), loss.data_ptr(), _stream())
            saved.append(mc)
            cfg.append((b * c, h, wd, w))
            w /= weight_decay
        ctx.save_for_backward(*saved)
        ctx.cfg = cfg
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dl = dloss.contiguous().float()
        grads = []
        for mc, (bc, h, wd, w) in zip(ctx.saved_tensors, ctx.cfg):
            g = torch.empty_like(mc)
            hbm_call("dn::smooth2_bwd_kernel", 8 * bc * h * wd, "dn_smooth2_bwd", mc.data_ptr(), dl.data_ptr(), bc, h, wd, w, g.data_ptr(), _stream())
            grads.append(g)
        return (None,) + tuple(grads)


def smooth_loss(pred_map):
    """reference loss_functions.py:367-386: second-order differences, per-scale weight 1, 1/2.3, 1/2.3^2, ..."""
    if type(pred_map) not in [tuple, list]:
        pred_map = [pred_map]
    return _Smooth.apply(2.3, *pred_map)


def smooth_DORN_loss(pred_map):
    """reference loss_functions.py:388-399 (single map, all K channels in one mean)."""
    return _Smooth.apply(2.3, pred_map)


# --------------------------------------------------------------------------------------------------------------- DORN
class _OrdinalLoss(torch.autograd.Function):
    """DORN_loss as one autograd node.  `sync` (a world size > 1): the loss is a WHOLE-BATCH mean (loss_functions.py:69-73 divide by
    the valid-pixel count of the batch the reference's DataParallel gathered on GPU0): every rank reduces its own pixels to
    (sum, count), the pair is summed over the ranks (distributed.exchange_loss_stats) and the division happens afterwards
    (dn_ordinal_loss_finalize); the backward yields world * d(global loss)/d(local probabilities), which the data-parallel
    optimizer's 1/world turns back into the sum -- the same scheme as _MaskedLoss."""

    @staticmethod
    def forward(ctx, gt_depth, ord_labels, target, max_depth, sync=0):
        require_cuda(ord_labels, "ordinal probabilities")
        require_cuda(gt_depth, "ground-truth depth")
        require_cuda(target, "SID target labels")
        oc = ord_labels.contiguous().float()
        n, k, h, w = oc.shape
        gtc = gt_depth.contiguous().float()
        tc = target.contiguous().to(torch.int32)
        if gtc.numel() != n * h * w or tc.numel() != n * h * w:
            raise ValueError("gt_depth / target must be [N,H,W] matching ord_labels [N,K,H,W]")
        dev = oc.device
        nb = _lib.load().dn_ordinal_loss_blocks(n, h * w)
        partial = torch.empty((nb, 2), dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.call("dn_ordinal_loss_fwd", oc.data_ptr(), gtc.data_ptr(), tc.data_ptr(), n, h * w, k, max_depth, partial.data_ptr(),
                  stats.data_ptr(), loss.data_ptr(), _stream())
        if sync > 1:
            from .distributed import exchange_loss_stats
            exchange_loss_stats(stats.view(1, 2), sum_cols=(0, 1))
            _lib.call("dn_ordinal_loss_finalize", stats.data_ptr(), loss.data_ptr(), _stream())
        ctx.save_for_backward(oc, gtc, tc, stats)
        ctx.cfg = (n, k, h, w, max_depth, float(max(sync, 1)))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        oc, gtc, tc, stats = ctx.saved_tensors
        n, k, h, w, max_depth, up = ctx.cfg
        dl = dloss.contiguous().float()
        dord = torch.empty_like(oc)
        _lib.call("dn_ordinal_loss_bwd", oc.data_ptr(), gtc.data_ptr(), tc.data_ptr(), stats.data_ptr(), dl.data_ptr(), n, h * w, k,
                  max_depth, up, dord.data_ptr(), _stream())
        return None, dord, None, None, None


def DORN_loss(gt_depth, ord_labels, target, datasets):
    """reference loss_functions.py:16-74.  Under one process per GPU the (sum, num_valid) pair is exchanged before the division, so the
    value and the gradients are the reference's whole-batch ones."""
    from .distributed import data_parallel_world
    return _OrdinalLoss.apply(gt_depth, ord_labels, target, _max_depth(datasets), data_parallel_world())


# ------------------------------------------------------------------------------------------------------------ metrics
@torch.no_grad()
def compute_errors(gt, pred, dataset='kitti', crop=True, unsupervised=False):
    """reference loss_functions.py:401-448 -> [abs_diff, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3] (python floats)."""
    require_cuda(pred, "predicted depth")
    require_cuda(gt, "ground-truth depth")
    b, h, w = gt.shape
    if dataset == 'kitti':
        # (the reference leaves max_depth undefined for crop=False, loss_functions.py:411-417; 80 is what it means)
        mx = 80.0
        y1, y2, x1, x2 = (int(0.40810811 * h), int(0.99189189 * h), int(0.03594771 * w), int(0.96405229 * w)) if crop else (0, h, 0, w)
    else:
        mx = 10.0
        y1, y2, x1, x2 = 0, h, 0, w
    gtc, pc = gt.contiguous().float(), pred.contiguous().float()
    scratch = torch.empty((b, 11), dtype=torch.float32, device=pc.device)
    out = torch.empty(8, dtype=torch.float32, device=pc.device)
    _lib.call("dn_compute_errors", gtc.data_ptr(), pc.data_ptr(), b, h, w, mx, y1, y2, x1, x2, 1 if unsupervised else 0,
              scratch.data_ptr(), out.data_ptr(), _stream())
    return out.tolist()
