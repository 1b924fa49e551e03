"""Shared pieces of the drop-in model mirrors: parameter containers with the reference's state_dict layout and the
autograd bridge that hands a whole-network HIP forward/backward to PyTorch as ONE node."""
import torch
import torch.nn as nn

from .. import engine

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
VGG_STAGES = ((0, 7), (7, 14), (14, 24), (24, 34), (34, 44))   # features[a:b] slices, models/Disp_vgg_BN.py:137-141


class VGG16BNContainer(nn.Module):
    """Parameter container laid out like torchvision.models.vgg16_bn() (cfg "D" + BatchNorm): `.features` Sequential
    (indices 0..43), `.avgpool`, `.classifier`.  torchvision only contributes the layout to the reference
    (models/Disp_vgg_BN.py:84); the modules here are never called -- they hold parameters/buffers under the same
    state_dict keys.  `with_classifier=False` drops the 123.6 M never-used classifier parameters."""

    def __init__(self, with_classifier=True):
        super().__init__()
        layers, c = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        if with_classifier:
            self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))


class VGG16Container(nn.Module):
    """torchvision.models.vgg16() layout (cfg "D", no BatchNorm): `.features` indices 0..30, `.avgpool`, `.classifier` -- what
    models/Disp_vgg_feature.py:85 holds as `self.features` (state_dict keys `features.features.N.*`, `features.classifier.N.*`)."""

    def __init__(self, with_classifier=True):
        super().__init__()
        layers, c = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        if with_classifier:
            self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))


def xavier_init_like_reference(module):
    """init_weights() of the reference nets: xavier_uniform_ on every Conv2d / ConvTranspose2d / Linear weight, zero
    bias; the BatchNorm branch is unreachable there (models/Disp_vgg_BN.py:116-120) so BN keeps gamma=1, beta=0."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class HipNetFunction(torch.autograd.Function):
    """One autograd node for a whole network (or sub-network).  forward: run the net's HIP schedule, recording the engine
    tape; backward: seed the output gradients, run the tape in reverse, hand parameter gradients back (or nothing for
    parameters whose gradient the engine wrote in place into an optimizer arena) and the gradients of those inputs that
    require one (feature maps handed to a decoder)."""

    @staticmethod
    def forward(ctx, net, n_in, grad_mode, *rest):
        inputs, params = rest[:n_in], rest[n_in:]
        # needs_input_grad ignores torch.no_grad() (and grad mode is always off INSIDE forward): run_net passes the caller's mode,
        # so validation / test_disp forwards do not record the tape nor keep the activations alive
        recording = grad_mode and any(ctx.needs_input_grad[3:])
        # outputs the loss never reads (l1_loss: scales 1-3) come back as None instead of zero maps autograd would have to fill and
        # the tape to copy: the reference's autograd does not visit them either
        ctx.set_materialize_grads(False)
        tape = engine.Tape(recording)
        sink = engine.GradSink()
        in_acts = [engine.Act.from_nchw(x, needs_grad=recording and ctx.needs_input_grad[3 + i]) for i, x in enumerate(inputs)]
        with engine.stream_scope():
            if inputs:
                if recording:
                    engine.choose_side_streams(inputs[0].shape[0] * inputs[0].shape[-2] * inputs[0].shape[-1])
                engine.prepack_all(inputs[0].device)
            outs = net._hip_forward(tape, sink, *in_acts)     # list[Act]
        last = tape.last_out
        if last is not None and last.C == 1 and not last.planar and any(a is last for a in outs):
            last.seed_borrow = True      # the finest head: its seeded gradient is read once, out of place (no owned copy: engine.seed_grad)
        ctx.tape, ctx.sink, ctx.outs, ctx.params, ctx.in_acts = tape, sink, outs, params, in_acts
        results = tuple(a.t if a.planar else (a.t.view(a.N, 1, a.H, a.W) if a.C == 1 else a.t.permute(0, 3, 1, 2)) for a in outs)
        # 1 / disp of the one-channel heads, written by the head kernels themselves: run_net attaches them to the
        # tensors it returns and functional.reciprocal() hands them out without a launch
        object.__setattr__(net, "_dn_recips", [a.recip_t.view(a.N, 1, a.H, a.W) if (a.recip_t is not None and a.C == 1 and not a.planar) else None
                                               for a in outs])
        for r in results:
            if not torch.is_floating_point(r):
                ctx.mark_non_differentiable(r)
        return results

    @staticmethod
    def backward(ctx, *grads):
        with engine.stream_scope():
            for a, g in zip(ctx.outs, grads):
                if g is not None:
                    engine.seed_grad(a, g)
            ctx.tape.run_backward()
            ctx.sink.settle(ctx.params)
            engine.join_side_stream()
        ig = tuple((a.grad.permute(0, 3, 1, 2) if (a.needs_grad and a.grad is not None) else None) for a in ctx.in_acts)
        pg = tuple(ctx.sink.get(p) for p in ctx.params)
        ctx.tape = ctx.outs = ctx.in_acts = None
        return (None, None, None) + ig + pg


def run_net(net, *inputs):
    for x in inputs:
        engine.require_cuda(x, "input tensor")
        if x.dtype != torch.float32:
            raise TypeError("expected float32 input, got %s" % x.dtype)
    params = getattr(net, "_dn_param_cache", None)
    if params is None or not params[0].is_cuda:
        # the Parameter objects are stable (.to() / load_state_dict() swap .data in place); walking named_parameters() every
        # forward costs ~0.25 ms of a 6 ms host-side step
        params = [p for p in net._hot_parameters()]
        for p in params:
            engine.require_cuda(p, "model parameters")
        object.__setattr__(net, "_dn_param_cache", params)
    outs = HipNetFunction.apply(net, len(inputs), torch.is_grad_enabled(), *inputs, *params)
    recips = getattr(net, "_dn_recips", None)
    if recips is not None:
        object.__setattr__(net, "_dn_recips", None)
        for o, r in zip(outs, recips):
            if r is not None:
                o._dn_recip = (r, o._version)      # valid while the disparity tensor is unmodified
    return outs
