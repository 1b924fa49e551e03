"""Host-side execution engine: NHWC activation records, a define-by-run backward tape at *block* granularity, and the
launch helpers that drive libdispnet_hip.so through its C ABI.

PyTorch is plumbing here (device memory, the caching allocator, streams, autograd hand-off at the model boundary).
Every numerically significant operation is a HIP kernel launched through supervised_dispnet_amd._lib.

Design notes
  * A model's forward is ordinary Python that calls the block helpers below.  Each helper launches its forward kernels
    and, when recording, appends one closure to the tape; the model's backward runs the tape in reverse.  There is no
    tracing compiler and no graph rewriting: fusion decisions are explicit in the helpers (BN-apply+ReLU of the producer
    is applied by the consumer's loader, concat and nearest-upsample are virtual, bias/activation/BN-statistics live in
    the conv epilogue).
  * `Act.grad` always holds the gradient w.r.t. the LOGICAL value of the activation (after its pending transform).
    The first writer overwrites, later writers accumulate (skip connections have two consumers).
"""
import ctypes as C
import math
import os
import threading
import weakref

import torch

from . import _lib
from ._lib import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID_AFFINE, CONV_DGRAD, CONV_FWD, CONVT_DGRAD, CONVT_FWD,
                   ConvDesc)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# bumped by optimizers that update parameters behind autograd's back (FusedAdam); part of the packed-weight cache key
PARAM_EPOCH = 0


# Optional per-launch instrumentation (bench.py): a list receiving (kernel_name, algorithmic_flops, start_event, end_event)
# for every implicit-GEMM launch, bracketed with HIP events on the launch stream.  None = off (zero overhead).
PROFILE = None

SPLITK = True                   # small Winograd / direct grids split their K axis over several blocks (tests switch it off to compare)
PACK_ON_SIDE_STREAM = True      # batched weight re-lay of the large layers under the first layers of the forward
PACK_LATE_MIN_ELEMS = 400000
BN_SUMS_FUSION = True           # input-gradient kernels take the BatchNorm backward's column sums of the layer below
FOLD_FINALIZE = True            # ... and finish them / the forward's BatchNorm statistics in their last-arriving block when the partial rows are few
                                # (dn_conv_desc.bnf_* / bnb_dgamma, bnb_dbeta; csrc/dn_fold.h).  Tests switch it off to compare: same bits
BN_MATERIALIZE_DZ = False       # True: the r01 BatchNorm backward (reduce pass writes dz, apply pass reads it back); the equivalence test sets it

# Arithmetic of the matrix-core kernels that offer a choice (dn_conv_desc.compute, include/dispnet_hip.h), today the Winograd forward /
# input-gradient kernels.  Tensors in HBM, statistics, transforms and all other kernels are fp32 in every mode.
#   "f32x3" (default) fp32 products on the bf16 matrix cores: operands split exactly into three bf16 pieces, six partial products,
#           fp32 accumulation -- an fp32 result: the same error against fp64 as "f32" or less
#           (tests/test_gpu_kernels.py::test_winograd_error_vs_fp64, ::test_winograd_compute_modes; the whole GPU suite -- oracle,
#           reference goldens, fp64 yardsticks -- passes in either mode), 1.3x the fp32 instruction's speed on these kernels
#   "f32"   fp32 FMA chain on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32)
#   "bf16"  operands ROUNDED to bf16, fp32 accumulation: the "mixed precision" mode of BASELINE configs[4]; opt-in only
# DN_COMPUTE=... or set_compute(...).
COMPUTE_MODES = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f32x3": _lib.COMPUTE_F32X3}
if os.environ.get("DN_COMPUTE", "f32x3").lower() not in COMPUTE_MODES:
    raise ValueError("DN_COMPUTE=%r: expected one of %s" % (os.environ["DN_COMPUTE"], sorted(COMPUTE_MODES)))
COMPUTE = COMPUTE_MODES[os.environ.get("DN_COMPUTE", "f32x3").lower()]


def set_compute(mode):
    global COMPUTE
    if mode not in COMPUTE_MODES:
        raise ValueError("compute mode is one of %s, got %r" % (sorted(COMPUTE_MODES), mode))
    COMPUTE = COMPUTE_MODES[mode]
    bump_param_epoch()


def compute_mode():
    return [k for k, v in COMPUTE_MODES.items() if v == COMPUTE][0]


# True while graph.GraphedStep records a step into a hipGraph (nothing in the engine may synchronise or time launches then)
CAPTURING = False


# Weight gradients on a side HIP stream (DN_WGRAD_STREAM=auto|1|0; auto = on): a layer's weight gradient
# and its input gradient only share read-only operands, so the two launches run side by side and fill each other's last,
# partially occupied round of blocks (a Winograd weight-gradient block needs a whole CU's LDS, so it takes the CUs the input
# gradient's tail leaves idle).  Measured +2.3 % images/sec (r01_h A/B on one box, identical losses).  The fences:
#   side waits for main before every weight gradient (dy, the operands, this layer's bias / BatchNorm gradients);
#   main waits for side at the end of the backward pass (optimizer, next forward);
#   a data-parallel gradient bucket is fenced against BOTH streams before it is exchanged (compute_streams() -> the RCCL stream's
#   hipStreamWaitEvent, or fence_streams() on the torch.distributed path); tensors the side stream reads are record_stream()ed
#   against the caching allocator.
_WGRAD_STREAM_MODE = os.environ.get("DN_WGRAD_STREAM", "auto")
WGRAD_STREAM = _WGRAD_STREAM_MODE != "0"
# Number of side streams the weight gradients alternate between.  At the metric's b32 the device is full and one is as good as two; at
# 4 images per GPU (b32 over 8 GPUs) no kernel fills the chip and the ONE side stream's queue (2.4 ms of weight-gradient launches that
# cannot start before the loss) had become the critical path of the step.
WGRAD_STREAMS = 2
WGRAD_STREAMS_MAX_PIXELS = 16 * 128 * 416   # measured: b4 -3.8 %, b8 -0.4 %, b16 -1.3 %, b32 +-0
SIDE_STREAMS_ACTIVE = 1


def choose_side_streams(input_pixels):
    """Called with N * H * W of a network's input at the start of its forward pass: how many side streams its weight gradients use."""
    global SIDE_STREAMS_ACTIVE
    SIDE_STREAMS_ACTIVE = WGRAD_STREAMS if input_pixels <= WGRAD_STREAMS_MAX_PIXELS else 1
_SIDE = {}


def wgrad_stream_enabled():
    if _WGRAD_STREAM_MODE == "0" or PROFILE is not None:      # per-launch event timing wants one kernel at a time
        return False
    # "auto" == on, also under data parallelism: every gradient bucket is fenced against BOTH compute streams before it is
    # exchanged (rccl.Communicator.all_reduce_sum_ / fence_streams), so the two-stream schedule needs no special case there
    return True


def side_stream():
    dev = torch.cuda.current_device()
    st = _SIDE.get(dev)
    if st is None:
        sides = [torch.cuda.Stream(device=dev) for _ in range(max(1, WGRAD_STREAMS))]
        st = _SIDE[dev] = {"side": sides[0], "sides": sides, "handles": {x.cuda_stream for x in sides}, "main": None, "rr": 0}
    return st


# Launch tape being recorded (graph.TapedStep): {"handle": dn_tape_begin()'s, "keep": tensors kept alive until the recording ends}.
# While it is set, every fence between two streams is also put on the tape (stream_wait), and tensors a second stream reads are kept
# alive instead of relying on the caching allocator's event-guarded reuse (whose timing the replay cannot reproduce).
TAPE = None
DEVICE_SCOPE_FENCES = True      # A/B: bench.py --system-fences
BORROW_SEED_GRADS = True        # seed_grad() hands a block that only READS its seeded gradient the framework's tensor instead of a copy


def stream_wait(waiter, waitee, device_scope=False):
    """waiter.wait_stream(waitee), also recorded on the launch tape when one is being recorded.  device_scope: both are compute streams
    of this device (main / weight-gradient side streams) -- the tape's event then carries no system-scope fence (dn_tape_fence_device:
    2-3 us less of the recording stream's queue per fence, ~60 fences in a step); never for work another GPU or the host reads."""
    waiter.wait_stream(waitee)
    if TAPE is not None and not TAPE.get("paused"):
        _lib.call("dn_tape_fence_device" if (device_scope and DEVICE_SCOPE_FENCES) else "dn_tape_fence", TAPE["handle"], waiter.cuda_stream,
                  waitee.cuda_stream)


def tape_host_call(fn):
    """Run fn() now and -- when a launch tape is being recorded -- at this point of every replay too (graph.TapedStep): host work
    that is not a launch of this library but belongs to the step (a gradient bucket handed to torch.distributed).  Launches and fences
    fn makes itself are NOT recorded: they happen live at replay."""
    if TAPE is None:
        return fn()
    _lib.load().dn_tape_mark(TAPE["handle"])
    # (with the stream it is issued on: a gradient bucket completed by a weight gradient is handed over from the SIDE stream's context,
    #  and its fences make THAT stream wait for the others -- replayed on the main stream they would stall the input-gradient chain)
    TAPE["host_calls"].append((fn, torch.cuda.current_stream()))
    _lib.call("dn_tape_pause", TAPE["handle"], 1)
    TAPE["paused"] = True
    try:
        return fn()
    finally:
        TAPE["paused"] = False
        _lib.call("dn_tape_pause", TAPE["handle"], 0)


class outside_tape_pool(object):
    """Allocate (and initialise with framework ops) a PERSISTENT tensor while a launch tape is being recorded: the recording routes
    every allocation to the tape's private pool, where a fresh block may be one that an earlier kernel of the same step wrote -- the
    replay would repeat that write but not the framework-side initialisation.  Inside this block allocations come from the regular
    pool again."""

    def __enter__(self):
        self.rec = TAPE
        if self.rec is not None and self.rec.get("pool") is not None:
            torch._C._cuda_endAllocateToPool(*self.rec["pool"])

    def __exit__(self, *exc):
        if self.rec is not None and self.rec.get("pool") is not None:
            torch._C._cuda_beginAllocateToPool(*self.rec["pool"])
            torch._C._cuda_releasePool(*self.rec["pool"])        # (begin takes a reference of its own each time; keep the count at the recorder's)
        return False


def cross_stream_use(t, stream):
    """`t` is read / written by work enqueued on `stream`, which is not the stream it was allocated on."""
    t.record_stream(stream)
    if TAPE is not None:
        TAPE["keep"].append(t)


def join_side_stream():
    """Make the current stream wait for everything enqueued on the side stream (end of a backward pass)."""
    if WGRAD_STREAM and torch.cuda.is_available():
        st = _SIDE.get(torch.cuda.current_device())
        if st is not None:
            cur = torch.cuda.current_stream()
            for x in st["sides"]:
                stream_wait(cur, x, device_scope=True)


def compute_streams():
    """The HIP streams gradients may have been produced on so far in this backward pass: the current one (the autograd thread's)
    plus the weight-gradient side stream once it exists -- what a gradient bucket must be fenced against before it is exchanged."""
    cur = torch.cuda.current_stream()
    out = [cur]
    st = _SIDE.get(torch.cuda.current_device())
    if WGRAD_STREAM and st is not None:
        for x in st["sides"]:
            if x != cur:
                out.append(x)
        if st["main"] is not None and st["main"] != cur and st["main"] not in st["sides"]:
            out.append(st["main"])
    return out


def fence_streams():
    """Make the CURRENT stream wait for both the main and the side stream (called right before a gradient bucket is handed to
    RCCL: the bucket holds gradients produced on either stream, whichever stream the last of them came from)."""
    if not (WGRAD_STREAM and torch.cuda.is_available()):
        return
    st = _SIDE.get(torch.cuda.current_device())
    if st is None:
        return
    cur = torch.cuda.current_stream()
    for x in st["sides"]:
        if cur != x:
            stream_wait(cur, x)
    if st["main"] is not None and cur != st["main"]:
        stream_wait(cur, st["main"])


_SPLITK = {}      # (device index, raw stream handle) -> (zeroed workspace, bytes, -, pointer): dn_conv_desc.splitk_ws of that stream's launches
_SPLITK_BYTES = 80 << 20     # >= 4096 + 256 blocks x 8 splits x 32 KB of partial tiles: the largest workspace any launch asks for (dn_winograd.hip)


_XCD_OK = {}      # device index -> bool: blocks with one blockIdx.x share an XCD (what the fence-free K split relies on)


def xcd_placement_ok(device):
    """One-time probe per device: the K-split kernels publish partial tiles to ONE XCD's L2 and rely on every block with the
    same blockIdx.x (gridDim.x a multiple of 8) running on the same XCD.  HIP promises no placement, so the property is CHECKED on the
    device in hand -- three grid shapes of the kind the split launches use, on the current stream and, concurrently, on a second one --
    and the split is simply not offered (results identical, small grids slower) when it does not hold."""
    ok = _XCD_OK.get(device.index)
    if ok is not None:
        return ok
    ok = True
    # (a launch tape being recorded must not see the probe: its launches would be replayed into buffers that are long freed -- found by
    #  test_two_ranks_through_the_launch_tape..., whose FIRST step is the recorded one; graph.TapedStep also probes before it records)
    pause = TAPE is not None and not TAPE.get("paused")
    if pause:
        _lib.call("dn_tape_pause", TAPE["handle"], 1)
        TAPE["paused"] = True
    try:
        with torch.cuda.device(device), outside_tape_pool():
            props = torch.cuda.get_device_properties(device)
            if "gfx950" not in getattr(props, "gcnArchName", "gfx950") or props.multi_processor_count % 8:
                ok = False
            shapes = ((8, 16, 1), (104, 4, 1), (32, 2, 4))
            # (the engine's own side stream, NOT a new one: HIP deals streams onto a few hardware queues in creation order, and one
            #  extra stream created here put the weight-gradient side streams on other queues -- measured 3.99 -> 4.52 ms per 4-image step)
            side = side_stream()["sides"][0]
            outs = []
            for st in (torch.cuda.current_stream(device), side):
                for gx, gy, gz in shapes:
                    o = torch.full((gx * gy * gz,), -1, dtype=torch.int32, device=device)
                    side.wait_stream(torch.cuda.current_stream(device))
                    _lib.call("dn_xcd_probe", o.data_ptr(), gx, gy, gz, st.cuda_stream)
                    outs.append((o, gx))
            torch.cuda.synchronize(device)
            for o, gx in outs:
                ids = o.cpu().view(-1, gx)
                if int(ids.min()) < 0 or int(ids.max()) > 7 or not bool((ids == ids[0:1]).all()):
                    ok = False
    except Exception:                                 # noqa: BLE001 -- a probe that cannot run proves nothing: no split
        ok = False
    finally:
        if pause:
            TAPE["paused"] = False
            _lib.call("dn_tape_pause", TAPE["handle"], 0)
    if not ok:
        import warnings
        warnings.warn("supervised_dispnet_amd: block -> XCD placement on this device is not blockIdx.x % 8; K split of small grids disabled")
    _XCD_OK[device.index] = ok
    return ok


def _splitk_workspace(d, device):
    """Attach the K-split workspace to a conv descriptor (small grids: a 4-image shard of the metric's batch); the
    library decides per launch whether it splits (dn_conv_splitk_workspace_bytes).  One zeroed buffer per (device, stream) serves every
    launch of that stream (the counters reset themselves; launches on different streams must not share one); launches on the
    weight-gradient side streams never split.  No torch calls on the hot path: the step makes ~80 of these and is launch-bound at 4
    images."""
    if not SPLITK or device.type != "cuda" or not xcd_placement_ok(device):
        return
    h = _stream()
    cur = _SPLITK.get((device.index, h))
    if cur is None:
        with torch.cuda.device(device):
            sides = side_stream()["handles"]
        if h in sides:
            _SPLITK[(device.index, h)] = cur = (None, 0, None, None)
        else:
            with outside_tape_pool():
                buf = torch.zeros(_SPLITK_BYTES // 4, dtype=torch.float32, device=device)
            cur = _SPLITK[(device.index, h)] = (buf, _SPLITK_BYTES, None, buf.data_ptr())
    if cur[0] is None:
        return
    d.splitk_ws = cur[3]
    d.splitk_ws_bytes = cur[1]


def _pick_bn(ntot):
    return 32 if ntot <= 32 else (64 if ntot <= 64 else 128)


class _Timed(object):
    """Context manager bracketing one C-ABI launch with HIP events when PROFILE is on."""

    def __init__(self, name, flops, tag="", abytes=0):
        self.name, self.flops, self.tag, self.abytes = name, flops, tag, abytes      # abytes: operands read once + results written once

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            launched = _lib.load().dn_last_kernel().decode(errors="replace")     # what the library actually ran (rocprofv3 name)
            PROFILE.append((launched or self.name, self.flops, self.e0, self.e1, self.tag, 0, int(self.abytes)))
        return False


def hbm_call(kernel, nbytes, entry, *args):
    """_lib.call(entry, *args) for an HBM-bound family; when PROFILE is on the launch is bracketed with HIP events and recorded
    with its ALGORITHMIC byte count (each operand read once, each result written once) under the name rocprofv3 prints for
    its dominant kernel -- bench.py's `roofline_hbm` entry."""
    if PROFILE is None:
        return _lib.call(entry, *args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(entry, *args)
    e1.record()
    if kernel is None:               # the entry point chooses between kernels: take the symbol it launched (the name rocprofv3 prints)
        kernel = _lib.load().dn_last_kernel().decode(errors="replace")
    PROFILE.append((kernel, 0, e0, e1, entry, int(nbytes), int(nbytes)))


def bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


_TLS = threading.local()        # per-thread cached raw hipStream_t (forward runs on the caller's thread, backward on autograd's)


def _stream():
    """Raw handle of the stream the next launch goes to.  torch.cuda.current_stream() costs ~10 us of Python per call and a
    training step makes ~300 launches, so the network-level entry points cache it for their duration (stream_scope)."""
    h = getattr(_TLS, "handle", None)
    return h if h is not None else torch.cuda.current_stream().cuda_stream


class stream_scope(object):
    """Cache the current stream's raw handle for the launches made inside the block (re-entrant; `stream=` switches to another
    torch stream for the block, as the side-stream weight gradients do)."""

    def __init__(self, stream=None):
        self.stream = stream

    def __enter__(self):
        self.prev = getattr(_TLS, "handle", None)
        if self.stream is not None:
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
            _TLS.handle = self.stream.cuda_stream
        else:
            self.ctx = None
            _TLS.handle = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _TLS.handle = self.prev
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _nbt_ptr(bn):
    """nn.BatchNorm2d.num_batches_tracked (int64 scalar on the device, bumped inside dn_bn_finalize) or None when untracked."""
    t = getattr(bn, "num_batches_tracked", None)
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.int64:
        t += 1
        return None
    return t.data_ptr()


def _ptr(t):
    return t.data_ptr() if t is not None else None


def require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the MI355X (got a %s tensor); this package is the HIP path only and has no "
                           "CPU fallback" % (what, t.device))


class Act:
    """An NHWC fp32 activation [N,H,W,C] plus what the engine needs to know about it."""
    __slots__ = ("t", "N", "H", "W", "C", "scale", "shift", "mean", "invstd", "grad", "grad_is_dz", "partial",
                 "partial_rows", "partial_stride", "partial_offset", "needs_grad", "strides", "no_relu", "planar", "pool_src", "sums_ready",
                 "recip_t", "bn_owner", "sums_final", "seed_borrow", "grad_borrowed")

    def __init__(self, t, N, H, W, C, strides=None, needs_grad=True):
        self.t = t
        self.N, self.H, self.W, self.C = N, H, W, C
        self.scale = self.shift = self.mean = self.invstd = None
        self.grad = None
        self.grad_is_dz = False
        self.partial = None
        self.partial_rows = 0
        self.partial_stride, self.partial_offset = 2, 0
        self.no_relu = False            # pending BatchNorm affine WITHOUT ReLU (bottleneck bn3 / downsample BN): only
                                        # block_bn_add_relu may consume it
        self.sums_ready = False         # .partial already holds the BatchNorm backward's column sums of .grad (taken in the epilogue of the
                                        # input-gradient kernel that wrote .grad: conv_dgrad / dn_conv_desc.bnb_*)
        self.pool_src = None            # (dpooled, idx): the gradient arrives through a 2x2 max-pool and is expanded by the BatchNorm
                                        # backward itself (dn_bn_bwd_apply_pool); .grad is then only the destination buffer
        self.planar = False             # .t is [N,C,H,W] (a network OUTPUT the caller's API wants planar: ord_c1, decode_c)
        self.recip_t = None             # a one-channel disparity head's 1 / disp, written by the head kernel (dn_conv_desc.recip_out)
        self.bn_owner = None            # (BatchNorm2d, GradSink) of a pre-BatchNorm activation: where the input-gradient kernel of the layer
                                        # above may put d(gamma), d(beta) when it finishes the sums itself (dn_conv_desc.bnb_dgamma / bnb_dbeta)
        self.sums_final = None          # (dgamma, dbeta) tensors that already hold the two BatchNorm-backward sums
        self.seed_borrow = False        # the producing block reads a seeded gradient WITHOUT writing it (block_conv_act: out-of-place
                                        # activation backward): seed_grad() may hand the framework's tensor over instead of a copy
        self.grad_borrowed = False      # .grad is the framework's tensor: read-only
        self.needs_grad = needs_grad
        self.strides = strides or (H * W * C, W * C, C, 1)   # (n, h, w, c) element strides

    @property
    def rows(self):
        return self.N * self.H * self.W

    @staticmethod
    def from_nchw(x, needs_grad=False):
        """Any NCHW-shaped tensor (any strides: contiguous NCHW or channels_last) consumed in place through operand strides."""
        a = Act.from_nchw_image(x)
        a.needs_grad = needs_grad
        return a

    @staticmethod
    def from_nchw_image(x):
        """User image [N,C,H,W] (any strides) consumed in place through operand strides -- no transpose pass."""
        n, c, h, w = x.shape
        sn, sc, sh, sw = x.stride()
        return Act(x, n, h, w, c, strides=(sn, sh, sw, sc), needs_grad=False)

    def new_like(self):
        return torch.empty((self.N, self.H, self.W, self.C), dtype=torch.float32, device=self.t.device)


class Tape:
    def __init__(self, recording):
        self.recording = recording
        self.steps = []
        self.last_out = None            # the Act the LAST pushed block produced, when that block can read a seeded gradient without writing it
                                        # (block_conv_act): nothing after it can have consumed it, so no block accumulates into its gradient

    def push(self, fn, out=None):
        self.last_out = out
        if self.recording:
            self.steps.append(fn)

    def run_backward(self):
        for fn in reversed(self.steps):
            fn()
        self.steps = []


class Piece:
    """One operand of a (virtually concatenated) conv input: an Act, optionally nearest-x2 upsampled on the fly."""
    __slots__ = ("act", "up")

    def __init__(self, act, up=False):
        self.act, self.up = act, up

    @property
    def C(self):
        return self.act.C


def _fill_operand(op, piece):
    a = piece.act
    if a.no_relu:
        raise RuntimeError("an activation with a pending ReLU-less BatchNorm can only feed block_bn_add_relu")
    op.data = a.t.data_ptr()
    op.C = a.C
    op.up_shift = 1 if piece.up else 0
    op.stride_n, op.stride_h, op.stride_w, op.stride_c = a.strides
    op.scale = _ptr(a.scale)
    op.shift = _ptr(a.shift)


def _fill_result(res, tensor, C_, H, W, accumulate, ld=None):
    ld = ld or C_
    res.data = tensor.data_ptr()
    res.C = C_
    res.accumulate = 1 if accumulate else 0
    res.stride_w = ld
    res.stride_h = W * ld
    res.stride_n = H * W * ld


class ConvLayer:
    """Runtime companion of one nn.Conv2d / nn.ConvTranspose2d: packed-weight caches + descriptor builders."""

    def __init__(self, module, transposed=False, in_channels_split=None, reflect_pad=0):
        self.m = module
        self.transposed = transposed
        self.R, self.S = module.kernel_size
        self.stride = module.stride[0]
        self.pad = module.padding[0]
        self.dil = module.dilation[0] if not transposed else 1
        self.reflect = reflect_pad > 0          # nn.ReflectionPad2d(reflect_pad) in front of a pad-0 conv (layers.py:124-136)
        if self.reflect:
            if transposed or self.pad != 0:
                raise ValueError("reflection padding goes with an un-padded nn.Conv2d")
            self.pad = reflect_pad
        self.out_pad = module.output_padding[0] if transposed else 0
        self.Cin = module.in_channels
        self.Cout = module.out_channels
        self._packed = {}

    def macs(self, N, IH, IW, OH, OW):
        """Multiply-accumulates of one forward (== of its dgrad and of its wgrad)."""
        px = N * IH * IW if self.transposed else N * OH * OW
        return px * self.Cin * self.Cout * self.R * self.S

    # -- geometry
    def out_size(self, H, W):
        if self.transposed:
            return ((H - 1) * self.stride - 2 * self.pad + self.R + self.out_pad,
                    (W - 1) * self.stride - 2 * self.pad + self.S + self.out_pad)
        return ((H + 2 * self.pad - self.dil * (self.R - 1) - 1) // self.stride + 1, (W + 2 * self.pad - self.dil * (self.S - 1) - 1) // self.stride + 1)

    def _desc(self, kind, N, IH, IW, OH, OW):
        d = ConvDesc()
        d.kind = kind
        d.N, d.IH, d.IW, d.OH, d.OW = N, IH, IW, OH, OW
        d.R, d.S, d.stride, d.pad = self.R, self.S, self.stride, self.pad
        d.pad_mode = 1 if (self.reflect and kind == CONV_FWD) else 0
        d.compute = COMPUTE
        d.dilation = self.dil
        return d

    def packed(self, kind, desc):
        """Packed weights for `kind`, re-laid only when the parameter changed.  Every (layer, kind, layout) re-lay is also entered
        in the device's PackTable, so that from the second optimizer step on ALL of them run as one batched launch at the start of
        the forward pass (prepack_all) and this method only hands out the buffers."""
        w = self.m.weight
        lib = _lib.load()
        layout = lib.dn_conv_weight_layout(C.byref(desc))     # direct or Winograd: depends on the geometry of this call
        split = (tuple(desc.in_[i].C for i in range(desc.n_in)), tuple(desc.out[i].C for i in range(desc.n_out)))
        key = (w.data_ptr(), w._version, PARAM_EPOCH) + split
        hit = self._packed.get((kind, layout))
        if hit is not None and hit[0] == key:
            return hit[1]
        table = pack_table(w.device) if w.is_cuda else None
        if table is not None and hit is not None and hit[0][:2] + hit[0][3:] == key[:2] + key[3:] and table.fresh(hit[1], PARAM_EPOCH):
            self._packed[(kind, layout)] = (key, hit[1])      # re-laid by the batched launch of this epoch
            if table.pending is not None:
                table.consumer_wait(hit[1])
            return hit[1]
        n = lib.dn_conv_packed_weight_elems(C.byref(desc))
        if n < 0:
            raise _lib.DispnetHipError("dn_conv_packed_weight_elems: " + _lib.last_error())
        n = max(int(n), 1)
        buf = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == w.device) else torch.empty(n, dtype=torch.float32, device=w.device)
        wc = w.detach()
        contiguous = wc.is_contiguous()
        if not contiguous:
            wc = wc.contiguous()
        _lib.call("dn_conv_pack_weights", C.byref(desc), wc.data_ptr(), buf.data_ptr(), _stream())
        self._packed[(kind, layout)] = (key, buf)
        if table is not None and contiguous and not getattr(self.m, "no_batch_pack", False):
            table.register((id(self), kind, layout) + split, desc, w, buf, self)
        elif table is not None and table.rows.pop((id(self), kind, layout) + split, None) is not None:
            table.dirty = True                   # a row of this layer must never outlive the buffer it points at
        return buf


class PackTable(object):
    """All weight re-lays of a training step as ONE batched launch (dn_pack_many) instead of ~53 launches of 6-8 us each.
    Rows are collected the first time a (layer, kind, layout) is packed the ordinary way; the table is uploaded when it changed and
    replayed at the start of the first forward after every optimizer step.  A row holds only WEAK references (to its ConvLayer, which
    owns the packed buffer, and to the weight tensor): when a model is deleted its rows die with it -- no leaked HBM, no re-lays of
    dead weights on later steps -- and a row is also dropped when its weight tensor moved."""
    MAX_ROWS = 512

    def __init__(self, device):
        self.device = device
        self.rows = {}                  # key -> [entry bytes, wino flag, weakref(weight), weakref(owning ConvLayer), weight ptr, buffer ptr]
        self.dirty = True
        self.dev_table = None
        self.counts = (0, 0, 0, 0)
        self.epoch = -1                 # PARAM_EPOCH whose weights the buffers of `covered` hold
        self.covered = set()
        self.esize = int(_lib.load().dn_pack_entry_bytes())
        # The re-lay of the LARGE weights (the 256 / 512-channel layers: 95 % of the bytes) runs on the weight-gradient side stream, under
        # the first layers of the forward pass, which do not need them; `pending` = (side stream, buffers it covers) until the first
        # consumer of one of them has made its stream wait (consumer_wait).  0.15 ms off the critical path of every step.
        self.late_table = None
        self.late_counts = (0, 0, 0, 0)
        self.late = set()
        self.pending = None
        self.split_mode = None

    def register(self, key, desc, w, buf, owner):
        if len(self.rows) >= self.MAX_ROWS:
            self._prune()
        if len(self.rows) >= self.MAX_ROWS:
            self.rows.clear()
        entry = (C.c_char * self.esize)()
        wino = _lib.load().dn_pack_entry_fill(C.byref(desc), w.data_ptr(), buf.data_ptr(), entry)
        if wino < 0:
            raise _lib.DispnetHipError("dn_pack_entry_fill: " + _lib.last_error())
        self.rows[key] = [bytes(entry), int(wino), weakref.ref(w), weakref.ref(owner), w.data_ptr(), buf.data_ptr()]
        self.dirty = True

    def _prune(self):
        """Drop rows whose layer or weight is gone (the packed buffer died with the layer: it must not be written again) or moved."""
        dead = []
        for k, r in self.rows.items():
            w = r[2]()
            if w is None or r[3]() is None or w.data_ptr() != r[4]:
                dead.append(k)
        for k in dead:
            del self.rows[k]
            self.dirty = True

    def fresh(self, buf, epoch):
        return self.epoch == epoch and buf.data_ptr() in self.covered

    def run(self, epoch):
        """Re-lay every registered row from the current weights (called once per optimizer step, before the forward)."""
        if self.epoch == epoch or not self.rows:
            return
        self._prune()
        if not self.rows:
            return
        split = PACK_ON_SIDE_STREAM and wgrad_stream_enabled()
        if self.dirty or split != self.split_mode:
            def upload(rows):
                kinds = [[r for r in rows if r[1] == k] for k in (0, 1, 2, 3)]   # direct, Winograd fp32 / bf16 / 3 x bf16
                ordered = kinds[0] + kinds[1] + kinds[2] + kinds[3]
                if not ordered:
                    return None, (0, 0, 0, 0)
                blob = b"".join(r[0] for r in ordered)
                with outside_tape_pool():
                    return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device), tuple(len(k) for k in kinds)
            rows = list(self.rows.values())
            late = [r for r in rows if split and (r[2]() is not None and r[2]().numel() > PACK_LATE_MIN_ELEMS)]
            early = [r for r in rows if not any(r is q for q in late)]
            self.dev_table, self.counts = upload(early)
            self.late_table, self.late_counts = upload(late)
            self.late = {r[5] for r in late}
            self.covered = {r[5] for r in rows}
            self.dirty = False
            self.split_mode = split
        if TAPE is not None and not TAPE.get("paused"):
            # the launches below go on a launch tape: everything they point at must live as long as the tape does -- the device tables
            # (replaced whenever a row changes), and the weights / packed buffers of EVERY row, also those of another model that
            # happens to be alive now and is deleted later (its rows are re-laid needlessly at replay, never into freed memory)
            TAPE["keep"] += [t for t in (self.dev_table, self.late_table) if t is not None]
            for r in self.rows.values():
                TAPE["keep"] += [x for x in (r[2](), r[3]()) if x is not None]
        if self.dev_table is not None:
            _lib.call("dn_pack_many", self.dev_table.data_ptr(), *self.counts, _stream())
        if self.late_table is not None:
            st = side_stream()
            main, side = torch.cuda.current_stream(), st["side"]
            stream_wait(side, main, device_scope=True)   # the optimizer step that produced these weights
            _lib.call("dn_pack_many", self.late_table.data_ptr(), *self.late_counts, side.cuda_stream)
            self.pending = (side, main)
        self.epoch = epoch

    def consumer_wait(self, buf):
        """First use of a buffer the side stream is re-laying: the consumer's stream waits for it (once per step)."""
        if self.pending is None or buf.data_ptr() not in self.late:
            return
        side, main = self.pending
        h = _stream()
        if h == side.cuda_stream:
            return                                   # (stream order; the main stream's first consumer still has to wait)
        stream_wait(main if h == main.cuda_stream else torch.cuda.current_stream(), side, device_scope=True)
        self.pending = None


_PACK_TABLES = {}


def pack_table(device):
    t = _PACK_TABLES.get(device)
    if t is None:
        t = _PACK_TABLES[device] = PackTable(device)
    return t


def prepack_all(device):
    """Start of a recorded forward pass: one batched re-lay of every packed weight the previous steps used (no-op when nothing
    changed since the last call)."""
    t = _PACK_TABLES.get(device)
    if t is not None:
        t.run(PARAM_EPOCH)


FUSE_RECIP = True       # a one-channel disparity head also emits depth = 1 / disp (train.py:445); tests switch it off to compare


def conv_forward(layer, pieces, act=ACT_NONE, p0=0.0, p1=0.0, bn_stats=False, out_hw=None, out_view=None, recip=None, bn_fold=None):
    """Forward of conv / conv-transpose over virtually concatenated `pieces`.  Returns (y tensor NHWC, partial, rows).
    `out_view` = (tensor, element offset, (stride_n, stride_h, stride_w), accumulate): write the result into a strided view of an
    existing tensor instead of a fresh one (FCRN's interleaved up-projection maps; the ASPP classifier's sum of four convolutions)."""
    a0 = pieces[0].act
    N = a0.N
    IH = a0.H * (2 if pieces[0].up else 1)
    IW = a0.W * (2 if pieces[0].up else 1)
    OH, OW = out_hw or layer.out_size(IH, IW)
    kind = CONVT_FWD if layer.transposed else CONV_FWD
    d = layer._desc(kind, N, IH, IW, OH, OW)
    d.n_in = len(pieces)
    for i, p in enumerate(pieces):
        _fill_operand(d.in_[i], p)
    d.n_out = 1
    if out_view is None:
        y = torch.empty((N, OH, OW, layer.Cout), dtype=torch.float32, device=a0.t.device)
        _fill_result(d.out[0], y, layer.Cout, OH, OW, False)
    else:
        y, off, (vsn, vsh, vsw), vacc = out_view
        r = d.out[0]
        r.data = y.data_ptr() + 4 * off
        r.C = layer.Cout
        r.accumulate = 1 if vacc else 0
        r.stride_n, r.stride_h, r.stride_w = vsn, vsh, vsw
    d.w_packed = layer.packed(kind, d).data_ptr()
    b = layer.m.bias
    d.bias = _ptr(b.detach()) if b is not None else None
    d.act, d.act_p0, d.act_p1 = act, p0, p1
    _splitk_workspace(d, a0.t.device)
    partial, rows = None, 0
    if recip is not None and out_view is None and layer.Cout == 1 and _lib.load().dn_conv_fwd_fuses_reciprocal(C.byref(d)) == 1:
        # `recip`: a one-element list the caller gets the [N, OH, OW, 1] reciprocal back in
        recip.append(torch.empty((N, OH, OW, 1), dtype=torch.float32, device=a0.t.device))
        d.recip_out = recip[0].data_ptr()
    if bn_stats:
        rows = _lib.load().dn_conv_bn_partial_rows(C.byref(d))
        partial = torch.empty((rows, layer.Cout, 2), dtype=torch.float32, device=y.device)
        d.bn_partial = partial.data_ptr()
        if bn_fold is not None and FOLD_FINALIZE and d.splitk_ws:
            # `bn_fold` = [BatchNorm2d, mean, invstd, scale, shift, num_batches_tracked pointer]: let the launch finish the statistics in its last-arriving blocks (what
            # dn_bn_finalize would do next); the answer goes back in bn_fold[0] (True: done, the caller skips dn_bn_finalize)
            bn = bn_fold[0]
            d.bnf_gamma, d.bnf_beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            d.bnf_running_mean, d.bnf_running_var = _ptr(bn.running_mean), _ptr(bn.running_var)
            d.bnf_num_batches_tracked = bn_fold[5]
            d.bnf_momentum = bn.momentum if bn.momentum is not None else BN_MOMENTUM
            d.bnf_eps = bn.eps
            d.bnf_mean, d.bnf_invstd, d.bnf_scale, d.bnf_shift = (t.data_ptr() for t in bn_fold[1:5])
            if _lib.load().dn_conv_fwd_folds_bn_finalize(C.byref(d)) == 1:
                bn_fold[0] = True
            else:
                d.bnf_scale = None
    if bn_fold is not None and bn_fold[0] is not True:
        bn_fold[0] = False
    with _Timed("igemm_conv_kernel<128,%d>" % _pick_bn(layer.Cout), 2 * layer.macs(N, IH, IW, OH, OW),
                "%s %dx%d k%d s%d cin%d cout%d in %dx%dx%d" % ("convT_fwd" if layer.transposed else "conv_fwd", layer.R, layer.S,
                                                               layer.R, layer.stride, layer.Cin, layer.Cout, N, IH, IW),
                4 * (sum(p.act.rows * p.act.C for p in pieces) + N * OH * OW * layer.Cout)):
        _lib.call("dn_convT2d_fwd" if layer.transposed else "dn_conv2d_fwd", C.byref(d), _stream())
    return y, partial, rows


def conv_wgrad(layer, pieces, dy, out_hw, out=None, sink=None, first=None):
    """Weight gradient in the framework layout (same shape as module.weight); written into `out` when given.  With `sink` the
    gradient is also handed to it here (inside the side-stream context when DN_WGRAD_STREAM=1, so that a data-parallel bucket
    launched by this gradient is fenced against the stream that computes it)."""
    a0 = pieces[0].act
    IH = a0.H * (2 if pieces[0].up else 1)
    IW = a0.W * (2 if pieces[0].up else 1)
    OH, OW = out_hw
    kind = CONVT_FWD if layer.transposed else CONV_FWD
    d = layer._desc(kind, a0.N, IH, IW, OH, OW)
    d.n_in = len(pieces)
    for i, p in enumerate(pieces):
        _fill_operand(d.in_[i], p)
    d.n_out = 1
    d.out[0].C = layer.Cout
    lib = _lib.load()
    nbytes = lib.dn_conv_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise _lib.DispnetHipError("dn_conv_wgrad_workspace_bytes: " + _lib.last_error())

    def launch():
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
        dw = out if out is not None else torch.empty_like(layer.m.weight, memory_format=torch.contiguous_format)
        with _Timed("igemm_wgrad_kernel<%d>" % _pick_bn(layer.Cin if layer.transposed else layer.Cout),
                    2 * layer.macs(a0.N, IH, IW, OH, OW),
                    "%s k%d s%d cin%d cout%d in %dx%dx%d" % ("convT_wgrad" if layer.transposed else "conv_wgrad", layer.R, layer.stride,
                                                             layer.Cin, layer.Cout, a0.N, IH, IW),
                    4 * (sum(p.act.rows * p.act.C for p in pieces) + dy.numel())):
            _lib.call("dn_conv2d_wgrad", C.byref(d), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes, _stream())
        if sink is not None:
            sink.put(layer.m.weight, dw)
        return dw

    # `first` = (callable, tensor it reads): work only the optimizer waits for (the second stage of the bias gradient's column sums),
    # run where the weight gradient runs
    if not (dy.is_cuda and wgrad_stream_enabled()):
        if first is not None:
            first[0]()
        return launch()
    st = side_stream()
    main = torch.cuda.current_stream()
    side = st["sides"][st["rr"] % min(len(st["sides"]), max(1, SIDE_STREAMS_ACTIVE))]   # weight gradients of successive layers alternate between the side streams
    st["rr"] += 1
    st["main"] = main
    stream_wait(side, main, device_scope=True)   # dy, the operands and the bias / BatchNorm gradients of this layer are ready
    with stream_scope(side):
        if first is not None:
            fresh = first[0]()
            cross_stream_use(first[1], side)
            if fresh is not None:
                cross_stream_use(fresh, main)
        dw = launch()
    cross_stream_use(dy, side)                   # the caching allocator must not hand these to the main stream while side reads them
    for p in pieces:
        cross_stream_use(p.act.t, side)
        if p.act.scale is not None:
            cross_stream_use(p.act.scale, side)
            cross_stream_use(p.act.shift, side)
    if out is None:
        cross_stream_use(dw, main)
    return dw


def conv_dgrad(layer, dy, N, OH, OW, pieces, in_hw):
    """Input gradient of conv / conv-transpose, routed into each piece's Act.grad (write first, accumulate after)."""
    targets = [p for p in pieces]
    if not any(p.act.needs_grad for p in targets):
        return
    IH, IW = in_hw                      # logical spatial size of the forward input
    kind = CONVT_DGRAD if layer.transposed else CONV_DGRAD
    rp = layer.pad if layer.reflect else 0
    GH, GW = IH + 2 * rp, IW + 2 * rp   # extent the kernel writes: the reflection-padded input when the conv reflects
    # the dgrad kernel reads dy (spatial OH x OW) and writes the forward-input-shaped gradient (GH x GW)
    d = layer._desc(kind, N, OH, OW, GH, GW)
    if layer.reflect:
        d.pad = 0                       # conv over the padded extent has no padding of its own
    d.n_in = 1
    op = d.in_[0]
    op.data = dy.data_ptr()
    op.C = layer.Cout
    op.up_shift = 0
    op.stride_c, op.stride_w, op.stride_h, op.stride_n = 1, layer.Cout, OW * layer.Cout, OH * OW * layer.Cout
    d.n_out = len(targets)
    post = []
    for i, p in enumerate(targets):
        a = p.act
        if p.up or layer.reflect:
            # gradient w.r.t. the (padded / upsampled) view: scratch, folded afterwards
            tmp = torch.empty((N, GH, GW, a.C), dtype=torch.float32, device=dy.device)
            _fill_result(d.out[i], tmp, a.C, GH, GW, False)
            if a.needs_grad:
                post.append((p, tmp))
        elif not a.needs_grad:
            # still needs a destination: scratch (rare: a detached skip inside a concat)
            tmp = torch.empty((N, IH, IW, a.C), dtype=torch.float32, device=dy.device)
            _fill_result(d.out[i], tmp, a.C, IH, IW, False)
        else:
            first = a.grad is None
            if first:
                a.grad = a.new_like()
            else:
                a.sums_ready = False    # a second consumer adds to this gradient: sums taken by the first writer no longer describe it
                a.sums_final = None
            _fill_result(d.out[i], a.grad, a.C, IH, IW, not first)
    d.w_packed = layer.packed(kind, d).data_ptr()
    d.bias = None
    d.act = ACT_NONE
    # BatchNorm backward of the layer below: when this call is the (first) writer of the gradient of a pre-BatchNorm activation with a
    # pending BN + ReLU, let its epilogue take the column sums (sum dz, sum dz * xhat) the BatchNorm backward needs -- one full read of
    # the gradient and one launch less per layer (dn_conv_desc.bnb_*); bn_backward() skips its sums pass when .sums_ready
    _splitk_workspace(d, dy.device)
    fuse = None
    if BN_SUMS_FUSION and len(targets) == 1 and not layer.transposed and not layer.reflect:
        a = targets[0].act
        if (not targets[0].up and a.needs_grad and a.scale is not None and a.mean is not None and not a.no_relu and not a.grad_is_dz
                and d.out[0].accumulate == 0 and a.strides == (a.H * a.W * a.C, a.W * a.C, a.C, 1)):
            d.bnb_y, d.bnb_scale, d.bnb_shift = a.t.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr()
            d.bnb_mean, d.bnb_invstd = a.mean.data_ptr(), a.invstd.data_ptr()
            rows = _lib.load().dn_conv_bn_partial_rows(C.byref(d))
            scratch = torch.empty((max(rows, 1), a.C, 2), dtype=torch.float32, device=dy.device)
            d.bnb_partial = scratch.data_ptr()
            if _lib.load().dn_conv_dgrad_fuses_bn_sums(C.byref(d)) == 1:
                fuse = (a, scratch, rows, None)
                if FOLD_FINALIZE and a.bn_owner is not None and d.splitk_ws:
                    # ... and FINISH the two sums in the launch's last-arriving blocks: d(gamma), d(beta) of the BatchNorm below land where
                    # bn_backward() would have its own sums launch put them
                    obn, osink = a.bn_owner
                    dg, db = osink.dest(obn.weight), osink.dest(obn.bias)
                    if dg is None:
                        dg = torch.empty(a.C, dtype=torch.float32, device=dy.device)
                    if db is None:
                        db = torch.empty(a.C, dtype=torch.float32, device=dy.device)
                    d.bnb_dgamma, d.bnb_dbeta = dg.data_ptr(), db.data_ptr()
                    if _lib.load().dn_conv_dgrad_folds_bn_sums(C.byref(d)) == 1:
                        fuse = (a, scratch, rows, (dg, db))
                    else:
                        d.bnb_dgamma = d.bnb_dbeta = None
            else:
                d.bnb_y = d.bnb_partial = None
    with _Timed("igemm_conv_kernel<128,%d>" % _pick_bn(layer.Cin), 2 * layer.macs(N, IH, IW, OH, OW),
                "%s k%d s%d cin%d cout%d in %dx%dx%d" % ("convT_dgrad" if layer.transposed else "conv_dgrad", layer.R, layer.stride,
                                                         layer.Cin, layer.Cout, N, IH, IW),
                4 * (dy.numel() + sum(p.act.rows * p.act.C for p in targets))):
        _lib.call("dn_convT2d_dgrad" if layer.transposed else "dn_conv2d_dgrad", C.byref(d), _stream())
    if fuse is not None:
        a, scratch, rows, final = fuse
        a.partial, a.partial_rows, a.partial_stride, a.partial_offset = scratch, rows, 2, 0
        a.sums_ready = True
        a.sums_final = final
    for p, tmp in post:
        a = p.act
        cur = tmp
        if layer.reflect:
            if p.up:
                cur = torch.empty((N, IH, IW, a.C), dtype=torch.float32, device=dy.device)
                _lib.call("dn_reflect_fold", tmp.data_ptr(), N, IH, IW, a.C, rp, cur.data_ptr(), 0, _stream())
            else:
                first = a.grad is None
                if first:
                    a.grad = a.new_like()
                _lib.call("dn_reflect_fold", tmp.data_ptr(), N, IH, IW, a.C, rp, a.grad.data_ptr(), 0 if first else 1, _stream())
                continue
        first = a.grad is None
        if first:
            a.grad = a.new_like()
        if a.C == 1:
            _lib.call("dn_upsample2x_nearest_bwd", cur.data_ptr(), N, a.H, a.W, a.grad.data_ptr(), 0 if first else 1, _stream())
        else:
            _lib.call("dn_upsample2x_nearest_bwd_nhwc", cur.data_ptr(), N, a.H, a.W, a.C, a.grad.data_ptr(), 0 if first else 1, _stream())


def colsum(partial, rows, Cn, stride=1, offset=0, out=None):
    if out is None:
        out = torch.empty(Cn, dtype=torch.float32, device=partial.device)
    _lib.call("dn_colsum_finalize", partial.data_ptr(), rows, Cn, stride, offset, out.data_ptr(), _stream())
    return out


def act_bwd(g, y_post, act, p0, p1, rows, Cn, out=None, defer=False, src=None):
    """g <- g * act'(y_post) in place (`src` given: g <- src * act'(y_post), src is not written); returns the bias gradient (column sums
    of the result), written into `out` if given.
    `defer`: returns (finish, partial) instead -- finish() runs the second stage of the column sums and returns the bias gradient; only
    the optimizer reads it, so conv_wgrad runs it on the weight-gradient side stream (a 5 us launch per layer off the critical path)."""
    nblk = _lib.load().dn_reduce_blocks(rows, Cn)
    partial = torch.empty((nblk, Cn), dtype=torch.float32, device=g.device)
    kname = "dn::colreduce_kernel<dn::ActBwdOp, %d>" % (4 if Cn % 4 == 0 else 1)
    if src is not None:
        hbm_call(kname, rows * Cn * 12, "dn_act_bwd_reduce_from", src.data_ptr(), g.data_ptr(), _ptr(y_post), act, p0, p1, rows, Cn,
                 partial.data_ptr(), _stream())
    else:
        hbm_call(kname, rows * Cn * 12, "dn_act_bwd_reduce", g.data_ptr(), _ptr(y_post), act, p0, p1, rows, Cn, partial.data_ptr(), _stream())
    if defer:
        return (lambda: colsum(partial, nblk, Cn, out=out)), partial
    return colsum(partial, nblk, Cn, out=out)


class GradSink:
    """Where parameter gradients go.  Default: a dict handed back to autograd.  An optimizer arena can register
    per-parameter destination views to have the engine write gradients in place (no autograd accumulation pass)."""

    reducer = None      # optional distributed.GradReducer notified as arena gradients land (set by the trainer)

    def __init__(self):
        self.grads = {}

    @staticmethod
    def dest(param):
        """In-place destination registered by an optimizer arena (contiguous, parameter-shaped), or None."""
        return getattr(param, "_dn_grad_view", None)

    def put(self, param, g):
        dst = getattr(param, "_dn_grad_view", None)
        if dst is not None:
            if g.data_ptr() != dst.data_ptr():
                dst.copy_(g.view_as(dst))
            param._dn_zero_grad_slot = False          # the slice now holds a real gradient: a later "no gradient" must zero it again
            self.grads[id(param)] = None
            if GradSink.reducer is not None:
                GradSink.reducer.grad_ready(param)
        else:
            self.grads[id(param)] = g

    def put_zero(self, param):
        """A gradient that is identically zero.  An arena slice that nothing ever writes stays at its initial zeros, so with
        an arena this is bookkeeping only (no fill launch every step)."""
        dst = getattr(param, "_dn_grad_view", None)
        if dst is not None:
            if not getattr(param, "_dn_zero_grad_slot", False):
                _lib.call("dn_fill", dst.data_ptr(), 0.0, dst.numel(), _stream())      # (this library's fill: visible to a launch tape)
                param._dn_zero_grad_slot = True
            self.grads[id(param)] = None
            if GradSink.reducer is not None:
                GradSink.reducer.grad_ready(param)
        else:
            self.grads[id(param)] = torch.zeros_like(param)

    def put_none(self, param):
        """No gradient reaches this parameter in this backward pass (its block's output was not used by the loss): autograd sees None;
        an arena slice keeps its zeros and the reducer is told the slot is settled, so its bucket is not held back to the end."""
        if getattr(param, "_dn_grad_view", None) is not None:
            self.put_zero(param)

    def settle(self, params):
        """End of a backward pass: every arena-backed parameter no block reported a gradient for (a block whose output the loss did not
        read returns early) is settled as "no gradient" -- its arena slice is zero, not last step's values."""
        for p in params:
            if id(p) not in self.grads and getattr(p, "_dn_grad_view", None) is not None and p.requires_grad:
                self.put_zero(p)

    def get(self, param):
        return self.grads.get(id(param))


def bn_backward(y, bn, sink, training):
    """BatchNorm backward of an activation stored pre-BN with the affine pending (block_conv_bn, block_upproject): y.grad holds
    dL/d(relu(bn(y))) -- or dz with its column sums in y.partial when y.grad_is_dz -- and becomes dL/dy in place; d(gamma), d(beta) go
    to the sink.  Returns the gradient tensor."""
    g, y_t, Cn, dev = y.grad, y.t, y.C, y.t.device
    if not training:
        raise NotImplementedError("backward through eval-mode BatchNorm (frozen statistics) is not implemented")
    relu_pending = not y.grad_is_dz           # g is dL/d(relu output): the mask is applied by the kernels below
    final = None
    if relu_pending and y.sums_ready and not BN_MATERIALIZE_DZ:
        y.sums_ready = False                  # the column sums came out of the epilogue of the kernel that wrote g (conv_dgrad)
        final, y.sums_final = y.sums_final, None      # ... already summed over the blocks too (dgamma, dbeta): no sums launch at all
    elif relu_pending:
        y.sums_ready = False
        if y.no_relu:
            raise RuntimeError("a ReLU-less BatchNorm output must be consumed by block_bn_add_relu / block_bn_plain")
        nblk = _lib.load().dn_reduce_blocks(y.rows, Cn)
        y.partial = torch.empty((nblk, Cn, 2), dtype=torch.float32, device=dev)
        y.partial_rows = nblk
        if BN_MATERIALIZE_DZ:
            hbm_call("dn::colreduce_kernel<dn::BnReluBwdOp, 4>", y.rows * Cn * 12, "dn_bn_relu_bwd_reduce", g.data_ptr(),
                     y_t.data_ptr(), y.scale.data_ptr(), y.shift.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), y.rows, Cn,
                     y.partial.data_ptr(), _stream())
        else:                                 # sums only: the apply pass re-derives the mask from y (one full-size write less)
            hbm_call("dn::colreduce_kernel<dn::BnReluBwdOp, 4>", y.rows * Cn * 8, "dn_bn_relu_bwd_sums", g.data_ptr(),
                     y_t.data_ptr(), y.scale.data_ptr(), y.shift.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), y.rows, Cn,
                     y.partial.data_ptr(), _stream())
    # written straight into the optimizer arena's gradient slices when there is one (no copy launches)
    if final is not None:
        dgamma, dbeta = final
    else:
        dgamma = sink.dest(bn.weight)
        dbeta = sink.dest(bn.bias)
        if dgamma is None:
            dgamma = torch.empty(Cn, dtype=torch.float32, device=dev)
        if dbeta is None:
            dbeta = torch.empty(Cn, dtype=torch.float32, device=dev)
    if y.pool_src is not None:                # gradient through the 2x2 max-pool: expanded here, never materialised as dz
        dpooled, pidx = y.pool_src
        hbm_call("dn::bn_bwd_apply_pool_kernel", y.rows * Cn * 8 + y.rows * Cn * 5 // 4, "dn_bn_bwd_apply_pool", dpooled.data_ptr(),
                 pidx.data_ptr(), y_t.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), bn.weight.data_ptr(), y.partial.data_ptr(),
                 y.partial_rows, y.partial_stride, y.partial_offset, y.N, y.H, y.W, Cn, g.data_ptr(), dgamma.data_ptr(),
                 dbeta.data_ptr(), _stream())
        y.pool_src = None
    elif relu_pending and not BN_MATERIALIZE_DZ:
        hbm_call(None, y.rows * Cn * 12, "dn_bn_bwd_apply_relu", g.data_ptr(), y_t.data_ptr(),
                 y.scale.data_ptr(), y.shift.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), bn.weight.data_ptr(),
                 None if final is not None else y.partial.data_ptr(), y.partial_rows, y.partial_stride, y.partial_offset, y.rows, Cn,
                 dgamma.data_ptr(), dbeta.data_ptr(), _stream())
    else:
        hbm_call(None, y.rows * Cn * 12, "dn_bn_bwd_apply", g.data_ptr(), y_t.data_ptr(), y.mean.data_ptr(),
                 y.invstd.data_ptr(), bn.weight.data_ptr(), y.partial.data_ptr(), y.partial_rows, y.partial_stride,
                 y.partial_offset, y.rows, Cn, dgamma.data_ptr(), dbeta.data_ptr(), _stream())
    sink.put(bn.weight, dgamma)
    sink.put(bn.bias, dbeta)
    return g


def _bn_pending(y, bn, partial, prow, training, conv_bias=None):
    """Fold batch statistics (training: from `partial`, also updating the running buffers) or running statistics (eval) of `bn` into the
    per-channel (scale, shift) pending on `y`."""
    Cn, dev = y.C, y.t.device
    y.scale = torch.empty(Cn, dtype=torch.float32, device=dev)
    y.shift = torch.empty(Cn, dtype=torch.float32, device=dev)
    if training:
        y.mean = torch.empty(Cn, dtype=torch.float32, device=dev)
        y.invstd = torch.empty(Cn, dtype=torch.float32, device=dev)
        _lib.call("dn_bn_finalize", partial.data_ptr(), prow, Cn, y.rows, _ptr(conv_bias.detach()) if conv_bias is not None else None,
                  bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                  bn.momentum if bn.momentum is not None else BN_MOMENTUM, bn.eps, y.mean.data_ptr(), y.invstd.data_ptr(),
                  y.scale.data_ptr(), y.shift.data_ptr(), _nbt_ptr(bn), _stream())
    else:
        _lib.call("dn_bn_eval_affine", Cn, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), bn.eps, y.scale.data_ptr(), y.shift.data_ptr(), _stream())


# ------------------------------------------------------------------------------------------------- block helpers
def block_conv_bn(tape, sink, x_piece, layer, bn, training, relu=True):
    """conv3x3 -> BatchNorm (batch statistics when training) -> ReLU, with the BN-apply+ReLU left pending on the result
    (the consumer's loader applies it).  Reference: torchvision vgg16_bn features triplets used at
    models/Disp_vgg_BN.py:137-141."""
    xa = x_piece.act
    dev = xa.t.device
    Cn = layer.Cout
    scale = torch.empty(Cn, dtype=torch.float32, device=dev)
    shift = torch.empty(Cn, dtype=torch.float32, device=dev)
    mean = invstd = fold = nbt = None
    if training:
        mean = torch.empty(Cn, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cn, dtype=torch.float32, device=dev)
        nbt = _nbt_ptr(bn)
        fold = [bn, mean, invstd, scale, shift, nbt]
    y_t, partial, prow = conv_forward(layer, [x_piece], ACT_NONE, bn_stats=training, bn_fold=fold)
    OH, OW = y_t.shape[1], y_t.shape[2]
    y = Act(y_t, xa.N, OH, OW, Cn)
    y.no_relu = not relu
    y.scale, y.shift, y.mean, y.invstd = scale, shift, mean, invstd
    y.bn_owner = (bn, sink)
    if training:
        if fold[0] is not True:              # (True: the convolution's last-arriving blocks finished the statistics themselves)
            _lib.call("dn_bn_finalize", partial.data_ptr(), prow, Cn, y.rows, _ptr(layer.m.bias.detach()) if layer.m.bias is not None else None,
                      bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                      bn.momentum if bn.momentum is not None else BN_MOMENTUM, bn.eps, y.mean.data_ptr(), y.invstd.data_ptr(),
                      y.scale.data_ptr(), y.shift.data_ptr(), nbt, _stream())
    else:
        _lib.call("dn_bn_eval_affine", Cn, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), bn.eps, y.scale.data_ptr(), y.shift.data_ptr(), _stream())

    def backward():
        if y.grad is None:
            return
        g = bn_backward(y, bn, sink, training)
        # g is now dL/d(conv output).  The conv bias sits in front of a BatchNorm: its gradient is identically zero in
        # exact arithmetic (BN removes the mean); the reference's value is rounding noise.  We store exact zeros.
        if layer.m.bias is not None:
            sink.put_zero(layer.m.bias)
        conv_wgrad(layer, [x_piece], g, (OH, OW), out=sink.dest(layer.m.weight), sink=sink)
        conv_dgrad(layer, g, xa.N, OH, OW, [x_piece], (xa.H, xa.W))
        y.grad = None

    tape.push(backward)
    return y


def block_pool(tape, y):
    """relu(bn(y)) then MaxPool2d(2,2) in one pass; the result is a plain activation (a skip tensor)."""
    dev = y.t.device
    p_t = torch.empty((y.N, y.H // 2, y.W // 2, y.C), dtype=torch.float32, device=dev)
    idx = torch.empty((y.N, y.H // 2, y.W // 2, y.C), dtype=torch.uint8, device=dev)
    hbm_call("dn::bn_relu_pool_fwd_kernel", y.rows * y.C * 4 * 1.25 + y.rows * y.C // 4, "dn_bn_relu_pool_fwd", y.t.data_ptr(),
             y.scale.data_ptr(), y.shift.data_ptr(), y.N, y.H, y.W, y.C, p_t.data_ptr(), idx.data_ptr(), _stream())
    p = Act(p_t, y.N, y.H // 2, y.W // 2, y.C)

    def backward():
        if p.grad is None:
            return
        nblk = _lib.load().dn_reduce_blocks(p.rows, y.C)
        y.partial = torch.empty((nblk, y.C, 2), dtype=torch.float32, device=dev)
        y.partial_rows = nblk
        y.grad = y.new_like()
        y.grad_is_dz = True
        if BN_MATERIALIZE_DZ:
            hbm_call("dn::colreduce_kernel<dn::PoolBwdOp, 4>", y.rows * y.C * 4 * 2.25 + y.rows * y.C // 4, "dn_bn_relu_pool_bwd",
                     p.grad.data_ptr(), idx.data_ptr(), y.t.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), y.N, y.H, y.W, y.C,
                     y.grad.data_ptr(), y.partial.data_ptr(), _stream())
        else:
            # sums only; the BatchNorm backward of the producing layer (block_conv_bn) expands (dpooled, idx) into dy in ITS pass:
            # the full-resolution dz (three quarters zeros) is never written nor read back
            hbm_call("dn::colreduce_kernel<dn::PoolBwdOp, 4>", y.rows * y.C * 4 * 1.25 + y.rows * y.C // 4, "dn_bn_relu_pool_bwd_sums",
                     p.grad.data_ptr(), idx.data_ptr(), y.t.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(), y.N, y.H, y.W, y.C,
                     y.partial.data_ptr(), _stream())
            y.pool_src = (p.grad, idx)
        p.grad = None

    tape.push(backward)
    return p


def block_conv_act(tape, sink, pieces, layer, act, p0=0.0, p1=0.0, out_hw=None, stat_bn=None):
    """conv / conv-transpose over virtually concatenated pieces + bias + activation in the epilogue.
    Reference: Conv2dBlock1 / ConvTranspose2dBlock1 / predict_disp (models/Disp_vgg_BN.py:40-70).
    `stat_bn`: a BatchNorm2d whose output the reference computes and DISCARDS (models/Disp_res_50.py:141-145: `bn1 =
    self.bn1(conv1); relu1 = self.relu(conv1)`): only its running statistics are updated, from the conv epilogue's sums."""
    a0 = pieces[0].act
    in_hw = (a0.H * (2 if pieces[0].up else 1), a0.W * (2 if pieces[0].up else 1))
    # (a DISPARITY head: alpha * sigmoid + beta with beta > 0, in training or eval; PoseExpNet's explainability masks are sigmoid heads
    #  with beta = 0 -- nobody reads 1 / mask, and it is inf where the mask underflows: ADVICE r4)
    want_recip = [] if (FUSE_RECIP and act == ACT_SIGMOID_AFFINE and layer.Cout == 1 and stat_bn is None and p1 > 0) else None
    y_t, partial, prow = conv_forward(layer, pieces, act, p0, p1, out_hw=out_hw, bn_stats=stat_bn is not None, recip=want_recip)
    OH, OW = y_t.shape[1], y_t.shape[2]
    y = Act(y_t, a0.N, OH, OW, layer.Cout)
    if want_recip:
        y.recip_t = want_recip[0]             # depth = 1 / disp, written by the head kernel: functional.reciprocal() hands it out
    if stat_bn is not None:
        Cn, dev = layer.Cout, y_t.device
        scratch = torch.empty((4, Cn), dtype=torch.float32, device=dev)
        _lib.call("dn_bn_finalize", partial.data_ptr(), prow, Cn, y.rows, _ptr(layer.m.bias.detach()) if layer.m.bias is not None else None,
                  stat_bn.weight.data_ptr(), stat_bn.bias.data_ptr(), stat_bn.running_mean.data_ptr(), stat_bn.running_var.data_ptr(),
                  stat_bn.momentum if stat_bn.momentum is not None else BN_MOMENTUM, stat_bn.eps, scratch[0].data_ptr(),
                  scratch[1].data_ptr(), scratch[2].data_ptr(), scratch[3].data_ptr(), _nbt_ptr(stat_bn), _stream())

    def backward():
        if y.grad is None:                   # an output no loss term reads (autograd hands None): no gradient, as in the reference
            sink.put_none(layer.m.weight)
            if layer.m.bias is not None:
                sink.put_none(layer.m.bias)
            return
        g, g_src = y.grad, None
        if y.grad_borrowed:                  # the framework's tensor (seed_grad): read it, write the result elsewhere
            g_src, g, y.grad_borrowed = g, torch.empty_like(y.grad), False
        finish_db, db_partial = act_bwd(g, y_t, act, p0, p1, y.rows, layer.Cout,
                                        out=sink.dest(layer.m.bias) if layer.m.bias is not None else None, defer=True, src=g_src)

        def bias_grad():
            db = finish_db()
            if layer.m.bias is not None:
                sink.put(layer.m.bias, db)
            return db if (layer.m.bias is None or sink.dest(layer.m.bias) is None) else None      # (a fresh tensor the caller's stream will read)

        conv_wgrad(layer, pieces, g, (OH, OW), out=sink.dest(layer.m.weight), sink=sink, first=(bias_grad, db_partial))
        conv_dgrad(layer, g, a0.N, OH, OW, pieces, in_hw)
        y.grad = None

    tape.push(backward, out=y)
    return y


def _not_on_tape(what):
    """Framework-side device work the launch tape cannot see: refuse while one is being recorded (the replay would silently skip it)."""
    if TAPE is not None and not TAPE.get("paused"):
        raise RuntimeError("launch tape: %s runs outside libdispnet_hip and would be missing from the replay" % what)


def seed_grad(act, g):
    """Seed an activation's gradient with one supplied by autograd (owned copy: the tape works in place)."""
    if act.planar:
        g = g.reshape(act.N, act.C, act.H, act.W)
        if act.grad is None:
            act.grad = g                 # read-only use by the producing block: no owned copy of a K-channel full-resolution map
            return
    else:
        g = g.reshape(act.N, act.H, act.W, act.C) if act.C == 1 else g.permute(0, 2, 3, 1)
    if act.grad is None:
        if act.seed_borrow and BORROW_SEED_GRADS and g.is_cuda and g.is_contiguous() and g.dtype == torch.float32:
            act.grad, act.grad_borrowed = g, True        # read once, out of place, by the producing block's activation backward
        elif g.is_cuda and g.is_contiguous() and g.dtype == torch.float32:
            act.grad = torch.empty_like(g)       # (this library's copy kernel: on the launch tape, which a framework clone would not be)
            _lib.call("dn_copy", g.data_ptr(), act.grad.data_ptr(), g.numel(), _stream())
        else:
            _not_on_tape("seed_grad of a non-contiguous gradient")
            act.grad = g.clone(memory_format=torch.contiguous_format)
    else:
        _not_on_tape("seed_grad into an existing gradient")
        act.grad.add_(g)


def block_bilinear_up2(tape, d, out_hw):
    """1-channel bilinear x2 (align_corners=False) cropped to out_hw -- models/DispNetS.py:120,126,132 + crop_like."""
    if d.C != 1:
        raise NotImplementedError("bilinear upsample is only needed for the 1-channel disparity")
    OH, OW = out_hw
    up_t = torch.empty((d.N, OH, OW, 1), dtype=torch.float32, device=d.t.device)
    _lib.call("dn_upsample2x_bilinear_fwd", d.t.data_ptr(), d.N, d.H, d.W, OH, OW, up_t.data_ptr(), _stream())
    up = Act(up_t, d.N, OH, OW, 1)

    def backward():
        if up.grad is None:
            return
        first = d.grad is None
        if first:
            d.grad = d.new_like()
        _lib.call("dn_upsample2x_bilinear_bwd", up.grad.data_ptr(), d.N, d.H, d.W, OH, OW, d.grad.data_ptr(),
                  0 if first else 1, _stream())
        up.grad = None

    tape.push(backward)
    return up


def ord_head_fusable(x, K):
    return (x.scale is None and x.strides == (x.H * x.W * x.C, x.W * x.C, x.C, 1)
            and bool(_lib.load().dn_ord_head_supported(x.C, x.H * x.W, K)))


def block_ord_head(tape, sink, x, conv, mask):
    """Dropout2d mask -> 1x1 conv (16 -> 2K) -> clamp -> pair softmax as ONE kernel (models/Disp_vgg_BN_DORN.py:112-114,191-227):
    returns (ord_c1 Act planar [N,K,H,W], decode_c Act planar int64 [N,1,H,W]); the 2K-channel logits never exist in HBM, and the
    backward recomputes them to go from d(ord_c1) straight to d(x), d(conv.weight), d(conv.bias)."""
    K = conv.out_channels // 2
    N, H, W = x.N, x.H, x.W
    dev = x.t.device
    w, b = conv.weight.detach(), conv.bias.detach()
    ord_t = torch.empty((N, K, H, W), dtype=torch.float32, device=dev)
    dec_t = torch.empty((N, 1, H, W), dtype=torch.int64, device=dev)
    mp = mask.data_ptr() if mask is not None else None
    px = N * H * W
    hbm_call("dn::ord_head_fwd_kernel", px * (16 + K + 2) * 4, "dn_ord_head_fwd", x.t.data_ptr(), mp, w.data_ptr(), b.data_ptr(), N, H * W, K,
             ord_t.data_ptr(), dec_t.data_ptr(), _stream())
    o = Act(ord_t, N, H, W, K)
    o.planar = True
    d = Act(dec_t, N, H, W, 1, needs_grad=False)
    d.planar = True

    def backward():
        if o.grad is None:
            return
        g = o.grad if o.grad.is_contiguous() else o.grad.contiguous()
        nblk = _lib.load().dn_ord_head_bwd_blocks(N, H * W)
        ws = torch.empty(nblk * (2 * K * 16 + 2 * K), dtype=torch.float32, device=dev)
        dw = sink.dest(conv.weight)
        if dw is None:
            dw = torch.empty_like(conv.weight, memory_format=torch.contiguous_format)
        db = sink.dest(conv.bias)
        if db is None:
            db = torch.empty_like(conv.bias)
        first = x.grad is None
        if x.needs_grad:
            if first:
                x.grad = x.new_like()
            dx = x.grad
        else:
            dx, first = x.new_like(), True
        hbm_call("dn::ord_head_bwd_kernel", px * (K + 48) * 4, "dn_ord_head_bwd", x.t.data_ptr(), mp, w.data_ptr(), b.data_ptr(), g.data_ptr(), N,
                 H * W, K, dx.data_ptr(), 0 if first else 1, ws.data_ptr(), dw.data_ptr(), db.data_ptr(), _stream())
        sink.put(conv.bias, db)
        sink.put(conv.weight, dw)
        o.grad = None

    tape.push(backward)
    return o, d


def block_channel_scale(tape, x, mask):
    """out[n,h,w,c] = x[n,h,w,c] * mask[n,c]  -- nn.Dropout2d in training mode with a precomputed keep/scale mask
    (models/Disp_vgg_BN_DORN.py:112,191).  The backward is the same kernel on the gradient."""
    out_t = x.new_like()
    _lib.call("dn_channel_scale", x.t.data_ptr(), mask.data_ptr(), x.N, x.H * x.W, x.C, out_t.data_ptr(), _stream())
    out = Act(out_t, x.N, x.H, x.W, x.C)

    def backward():
        if out.grad is None:
            return
        if x.grad is None:
            x.grad = x.new_like()
            _lib.call("dn_channel_scale", out.grad.data_ptr(), mask.data_ptr(), x.N, x.H * x.W, x.C, x.grad.data_ptr(), _stream())
        else:
            tmp = x.new_like()
            _lib.call("dn_channel_scale", out.grad.data_ptr(), mask.data_ptr(), x.N, x.H * x.W, x.C, tmp.data_ptr(), _stream())
            x.grad.add_(tmp)
        out.grad = None

    tape.push(backward)
    return out


def block_bn_add_relu(tape, y, r):
    """out = relu(bn(y) + identity): the tail of a ResNet bottleneck (models/Disp_res_50.py:229-247).  `y` carries a pending
    ReLU-less BatchNorm (block_conv_bn(..., relu=False)); `r` is either a plain activation (identity) or another ReLU-less
    BatchNorm output (the downsample branch).  One HBM pass forward, one backward (which also produces the BatchNorm
    backward's column sums for both branches)."""
    if r is not None and not y.no_relu:
        raise RuntimeError("block_bn_add_relu: y must come from block_conv_bn(relu=False)")
    ds = r is not None and r.no_relu
    if r is not None and not ds and r.scale is not None:
        raise RuntimeError("block_bn_add_relu: the identity branch must be a plain activation")
    dev = y.t.device
    out_t = y.new_like()
    _lib.call("dn_bn_add_relu_fwd", y.t.data_ptr(), y.scale.data_ptr(), y.shift.data_ptr(), r.t.data_ptr() if r is not None else None,
              r.scale.data_ptr() if ds else None, r.shift.data_ptr() if ds else None, y.rows, y.C, out_t.data_ptr(), _stream())
    out = Act(out_t, y.N, y.H, y.W, y.C)

    def backward():
        if out.grad is None:
            return
        if y.mean is None:
            raise NotImplementedError("backward through eval-mode BatchNorm (frozen statistics) is not implemented")
        nblk = _lib.load().dn_reduce_blocks(y.rows, y.C)
        partial = torch.empty((nblk, y.C, 4), dtype=torch.float32, device=dev)
        y.grad = y.new_like()
        y.grad_is_dz = True
        y.partial, y.partial_rows, y.partial_stride, y.partial_offset = partial, nblk, 4, 0
        dr, acc = None, 0
        if ds:
            r.grad = r.new_like()
            r.grad_is_dz = True
            r.partial, r.partial_rows, r.partial_stride, r.partial_offset = partial, nblk, 4, 2
            dr = r.grad
        elif r is not None and r.needs_grad:
            acc = 0 if r.grad is None else 1
            if r.grad is None:
                r.grad = r.new_like()
            dr = r.grad
        _lib.call("dn_bn_add_relu_bwd", out.grad.data_ptr(), out_t.data_ptr(), y.t.data_ptr(), y.mean.data_ptr(), y.invstd.data_ptr(),
                  r.t.data_ptr() if r is not None else None, r.mean.data_ptr() if ds else None, r.invstd.data_ptr() if ds else None, y.rows, y.C, y.grad.data_ptr(),
                  _ptr(dr), acc, partial.data_ptr(), _stream())
        out.grad = None

    tape.push(backward)
    return out


def block_bn_relu(tape, y):
    """Materialise relu(bn(y)) of a block_conv_bn result as a plain activation (needed when it leaves the engine as a tensor or
    feeds a max-pool other than the fused 2x2 one)."""
    return block_bn_add_relu(tape, y, None)


def block_maxpool3s2(tape, x, ceil_mode=False):
    """MaxPool2d(kernel_size=3, stride=2, padding=1[, ceil_mode=True]) of a plain activation (models/Disp_res_50.py:73; ASPP.py:138)."""
    if x.scale is not None:
        raise RuntimeError("block_maxpool3s2 expects a plain activation")
    cm = 1 if ceil_mode else 0
    lib = _lib.load()
    OH, OW = lib.dn_maxpool3s2_out(x.H, cm), lib.dn_maxpool3s2_out(x.W, cm)
    dev = x.t.device
    o_t = torch.empty((x.N, OH, OW, x.C), dtype=torch.float32, device=dev)
    idx = torch.empty((x.N, OH, OW, x.C), dtype=torch.uint8, device=dev)
    _lib.call("dn_maxpool3s2_fwd", x.t.data_ptr(), x.N, x.H, x.W, x.C, cm, o_t.data_ptr(), idx.data_ptr(), _stream())
    out = Act(o_t, x.N, OH, OW, x.C)

    def backward():
        if out.grad is None or not x.needs_grad:
            return
        first = x.grad is None
        if first:
            x.grad = x.new_like()
        _lib.call("dn_maxpool3s2_bwd", out.grad.data_ptr(), idx.data_ptr(), x.N, x.H, x.W, x.C, cm, x.grad.data_ptr(), 0 if first else 1, _stream())
        out.grad = None

    tape.push(backward)
    return out


def block_maxpool2(tape, x):
    """nn.MaxPool2d(2, 2) on a plain (already activated) tensor -- the BatchNorm-free VGG encoders (models/Disp_vgg.py:79-100,
    Disp_vgg_feature.py:138-142).  Odd extents drop their last row / column like ATen (floor)."""
    if x.scale is not None:
        raise RuntimeError("block_maxpool2 expects a plain activation (block_pool handles the pending-BatchNorm case)")
    if (x.H | x.W) & 1:
        raise NotImplementedError("2x2 max-pool of an odd-sized map (%dx%d)" % (x.H, x.W))
    dev = x.t.device
    o_t = torch.empty((x.N, x.H // 2, x.W // 2, x.C), dtype=torch.float32, device=dev)
    idx = torch.empty((x.N, x.H // 2, x.W // 2, x.C), dtype=torch.uint8, device=dev)
    hbm_call("dn::bn_relu_pool_fwd_kernel", x.rows * x.C * 4 * 1.25 + x.rows * x.C // 4, "dn_bn_relu_pool_fwd", x.t.data_ptr(), None, None,
             x.N, x.H, x.W, x.C, o_t.data_ptr(), idx.data_ptr(), _stream())
    out = Act(o_t, x.N, x.H // 2, x.W // 2, x.C)

    def backward():
        if out.grad is None or not x.needs_grad:
            return
        first = x.grad is None
        if first:
            x.grad = x.new_like()
        hbm_call("dn::maxpool2_bwd_kernel", x.rows * x.C * 4 * 1.25 + x.rows * x.C // 4, "dn_maxpool2_bwd", out.grad.data_ptr(), idx.data_ptr(),
                 x.N, x.H, x.W, x.C, x.grad.data_ptr(), 0 if first else 1, _stream())
        out.grad = None

    tape.push(backward)
    return out


def normalize_input(x, sub, div):
    """(x - sub) / div on the user's NCHW image (networks/vgg_encoder.py:80, resnet_encoder.py:89) -> plain NCHW tensor."""
    xc = x.contiguous()
    out = torch.empty_like(xc)
    _lib.call("dn_sub_div", xc.data_ptr(), xc.numel(), sub, div, out.data_ptr(), _stream())
    return out


def block_spatial_mean(tape, x, scale=1.0):
    """[N,H,W,C] -> [N,1,1,C] = scale * mean over H,W  (models/PoseExpNet.py:73-75)."""
    if x.scale is not None:
        raise RuntimeError("block_spatial_mean expects a plain activation")
    o_t = torch.empty((x.N, 1, 1, x.C), dtype=torch.float32, device=x.t.device)
    _lib.call("dn_spatial_mean_fwd", x.t.data_ptr(), x.N, x.H * x.W, x.C, scale, o_t.data_ptr(), _stream())
    out = Act(o_t, x.N, 1, 1, x.C)

    def backward():
        if out.grad is None:
            return
        if x.grad is not None:
            raise RuntimeError("block_spatial_mean: single-consumer input expected")
        x.grad = x.new_like()
        _lib.call("dn_spatial_mean_bwd", out.grad.data_ptr(), x.N, x.H * x.W, x.C, scale, x.grad.data_ptr(), _stream())
        out.grad = None

    tape.push(backward)
    return out


# ------------------------------------------------------------------------------------------------- FCRN / ASPP blocks
def block_bn_plain(tape, y):
    """Materialise bn(y) WITHOUT ReLU of a block_conv_bn(..., relu=False) result as a plain activation (FCRN: `x = self.bn2(self.conv2(x))`
    feeds the first up-projection, models/FCRN.py:236-239).  Backward: dz = d(out); its column sums (sum dz, sum dz * xhat) come from the
    ReLU-backward sums kernel run with an always-true mask (scale 0, shift 1), then the producer's BatchNorm backward applies them."""
    if not y.no_relu:
        raise RuntimeError("block_bn_plain: y must come from block_conv_bn(relu=False)")
    out_t = y.new_like()
    _lib.call("dn_bn_apply_fwd", y.t.data_ptr(), y.scale.data_ptr(), y.shift.data_ptr(), y.rows, y.C, out_t.data_ptr(), _stream())
    out = Act(out_t, y.N, y.H, y.W, y.C)

    def backward():
        if out.grad is None:
            return
        if y.mean is None:
            raise NotImplementedError("backward through eval-mode BatchNorm (frozen statistics) is not implemented")
        dev = y.t.device
        nblk = _lib.load().dn_reduce_blocks(y.rows, y.C)
        y.partial = torch.empty((nblk, y.C, 2), dtype=torch.float32, device=dev)
        y.partial_rows, y.partial_stride, y.partial_offset = nblk, 2, 0
        zero, one = torch.zeros(y.C, dtype=torch.float32, device=dev), torch.ones(y.C, dtype=torch.float32, device=dev)
        _lib.call("dn_bn_relu_bwd_sums", out.grad.data_ptr(), y.t.data_ptr(), zero.data_ptr(), one.data_ptr(), y.mean.data_ptr(),
                  y.invstd.data_ptr(), y.rows, y.C, y.partial.data_ptr(), _stream())
        y.grad = out.grad
        y.grad_is_dz = True
        out.grad = None

    tape.push(backward)
    return out


class _CompositeConvT(object):
    """Stand-in for the nn.ConvTranspose2d(6x6, stride 2, padding 2) that one branch of FCRN's up-projection amounts to; `.weight` is a
    plain tensor [Cin][Cout][6][6] rebuilt from the branch's four nn.Conv2d weights before every use (assemble / scatter_grad)."""
    kernel_size, stride, padding, output_padding, dilation, bias = (6, 6), (2, 2), (2, 2), (0, 0), (1, 1), None
    # `.weight` is scratch rebuilt in place before every use: a row in the batched re-lay table would re-lay STALE contents on the side
    # stream every step, mark the table dirty (a blocking host-to-device upload) and race the ordinary re-pack that follows
    no_batch_pack = True

    def __init__(self, convs):
        self.convs = convs                                  # phase order (0,0), (0,1), (1,0), (1,1)
        self.in_channels, self.out_channels = convs[0].in_channels, convs[0].out_channels
        self.weight = None

    def assemble(self):
        w0 = self.convs[0].weight
        if self.weight is None or self.weight.device != w0.device:
            self.weight = torch.zeros((self.in_channels, self.out_channels, 6, 6), dtype=torch.float32, device=w0.device)
        wt = self.weight
        for k, m in enumerate(self.convs):
            a, b = k >> 1, k & 1
            R, S = m.kernel_size
            # wt[ci][co][a + 4 - 2 kr][b + 4 - 2 ks] = w[co][ci][kr][ks]
            wt[:, :, a + 4 - 2 * (R - 1):a + 5:2, b + 4 - 2 * (S - 1):b + 5:2] = m.weight.detach().permute(1, 0, 2, 3).flip(2, 3)
        return wt

    def scatter_grad(self, dwt, sink):
        for k, m in enumerate(self.convs):
            a, b = k >> 1, k & 1
            R, S = m.kernel_size
            dw = dwt[:, :, a + 4 - 2 * (R - 1):a + 5:2, b + 4 - 2 * (S - 1):b + 5:2].flip(2, 3).permute(1, 0, 2, 3)
            dst = sink.dest(m.weight)
            if dst is not None:
                dst.copy_(dw)
                sink.put(m.weight, dst)
            else:
                sink.put(m.weight, dw.contiguous())


def block_upproject(tape, sink, x, up, rt, training):
    """FCRN's up-projection block (models/FCRN.py:52-124): per branch four convolutions (3x3, 2x3, 3x2, 2x2; one zero row above and one
    zero column to the left, the rest implied by output size = input size) whose results interleave into a 2H x 2W map -- phase (a, b) =
    (row parity, column parity) = conv*_1, conv*_2, conv*_3, conv*_4 --, BatchNorm over the interleaved map; branch 1: ReLU, conv3x3,
    BatchNorm; branch 2: BatchNorm only; sum, ReLU.  A branch's four convolutions ARE a ConvTranspose2d(6x6, stride 2, padding 2) on a
    composite weight (include/dispnet_hip.h, dn_phase_bias_add), so each branch is one run of the engine's four-phase transposed
    convolution + the four biases; the statistics of the interleaved maps are taken by dn_bn_stats_partial; bn1_1 + ReLU stays pending
    for conv3's loader; the tail is block_bn_add_relu.  `rt`: {"t1", "t2": ConvLayer over a _CompositeConvT, "conv3": ConvLayer}."""
    if x.scale is not None:
        raise RuntimeError("block_upproject expects a plain activation")
    N, H, W = x.N, x.H, x.W
    Cn = up.conv3.out_channels
    dev = x.t.device
    P = Piece
    maps = []
    for br in (1, 2):
        layer = rt["t%d" % br]
        layer.m.assemble()
        o_t, _, _ = conv_forward(layer, [P(x)])
        bias4 = torch.stack([m.bias.detach() for m in layer.m.convs]).contiguous()
        _lib.call("dn_phase_bias_add", o_t.data_ptr(), N, 2 * H, 2 * W, Cn, bias4.data_ptr(), _stream())
        o = Act(o_t, N, 2 * H, 2 * W, Cn)
        bn = up.bn1_1 if br == 1 else up.bn1_2
        partial, prow = None, 0
        if training:
            prow = _lib.load().dn_bn_stats_rows(o.rows)
            partial = torch.empty((prow, Cn, 2), dtype=torch.float32, device=dev)
            _lib.call("dn_bn_stats_partial", o_t.data_ptr(), o.rows, Cn, partial.data_ptr(), _stream())
        _bn_pending(o, bn, partial, prow, training)
        maps.append(o)
    o1, o2 = maps
    o2.no_relu = True

    def backward():
        # d(o1) arrives from conv3's input gradient (through the pending ReLU), d(o2) as dz from block_bn_add_relu
        for br, o in ((1, o1), (2, o2)):
            if o.grad is None:
                continue
            layer = rt["t%d" % br]
            bn = up.bn1_1 if br == 1 else up.bn1_2
            g = bn_backward(o, bn, sink, training)                      # dL/d(interleaved pre-BatchNorm map)
            lib = _lib.load()
            ws = torch.empty(lib.dn_phase_colsum_workspace_bytes(Cn) // 4, dtype=torch.float32, device=dev)
            db4 = torch.empty((4, Cn), dtype=torch.float32, device=dev)
            _lib.call("dn_phase_colsum", g.data_ptr(), N, H, W, Cn, ws.data_ptr(), db4.data_ptr(), _stream())
            for k, m in enumerate(layer.m.convs):
                dst = sink.dest(m.bias)
                if dst is not None:
                    dst.copy_(db4[k])
                sink.put(m.bias, dst if dst is not None else db4[k])
            layer.m.assemble()                                          # (the composite is shared scratch: rebuild for this branch's dgrad)
            dwt = conv_wgrad(layer, [P(x)], g, (2 * H, 2 * W))
            join_side_stream()                                          # the composite gradient may come from the weight-gradient side stream
            layer.m.scatter_grad(dwt, sink)
            conv_dgrad(layer, g, N, 2 * H, 2 * W, [P(x)], (H, W))
            o.grad = None

    tape.push(backward)
    y3 = block_conv_bn(tape, sink, P(o1), rt["conv3"], up.bn2, training, relu=False)
    return block_bn_add_relu(tape, y3, o2)


def block_sum_convs_act(tape, sink, x, layers, act, p0=0.0, p1=0.0):
    """act(sum_i conv_i(x)): the ASPP classifier (models/ASPP.py:107-123 -- four dilated 3x3 convolutions 2048 -> 1, summed, then
    10 * sigmoid + 0.01).  The first convolution writes the map, the others accumulate into it; every bias gets the same gradient."""
    if x.scale is not None:
        raise RuntimeError("block_sum_convs_act expects a plain activation")
    N, H, W = x.N, x.H, x.W
    Cn = layers[0].Cout
    dev = x.t.device
    pre = torch.empty((N, H, W, Cn), dtype=torch.float32, device=dev)
    dense = (H * W * Cn, W * Cn, Cn)
    for i, layer in enumerate(layers):
        conv_forward(layer, [Piece(x)], out_hw=(H, W), out_view=(pre, 0, dense, i > 0))
    out_t = torch.empty_like(pre)
    _lib.call("dn_act_fwd", pre.data_ptr(), pre.numel(), act, p0, p1, out_t.data_ptr(), _stream())
    out = Act(out_t, N, H, W, Cn)

    def backward():
        if out.grad is None:
            return
        g = out.grad
        db = act_bwd(g, out_t, act, p0, p1, N * H * W, Cn)
        for layer in layers:
            if layer.m.bias is not None:
                dst = sink.dest(layer.m.bias)
                if dst is not None:
                    dst.copy_(db)
                sink.put(layer.m.bias, dst if dst is not None else db.clone())
            conv_wgrad(layer, [Piece(x)], g, (H, W), out=sink.dest(layer.m.weight), sink=sink)
            conv_dgrad(layer, g, N, H, W, [Piece(x)], (H, W))
        out.grad = None

    tape.push(backward)
    return out


def block_resize_bilinear(tape, d, out_hw, align_corners=True):
    """F.interpolate(d, size=out_hw, mode='bilinear', align_corners=...) of a one-channel map (models/FCRN.py:253, models/ASPP.py:192)."""
    if d.C != 1 or d.scale is not None:
        raise NotImplementedError("block_resize_bilinear: one-channel plain maps only")
    OH, OW = out_hw
    ac = 1 if align_corners else 0
    o_t = torch.empty((d.N, OH, OW, 1), dtype=torch.float32, device=d.t.device)
    _lib.call("dn_resize_bilinear_fwd", d.t.data_ptr(), d.N, d.H, d.W, OH, OW, ac, o_t.data_ptr(), _stream())
    out = Act(o_t, d.N, OH, OW, 1)

    def backward():
        if out.grad is None:
            return
        first = d.grad is None
        if first:
            d.grad = d.new_like()
        _lib.call("dn_resize_bilinear_bwd", out.grad.data_ptr(), d.N, d.H, d.W, OH, OW, ac, d.grad.data_ptr(), 0 if first else 1, _stream())
        out.grad = None

    tape.push(backward)
    return out
