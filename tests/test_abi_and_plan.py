"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/dispnet_hip.h declares
(no compute calls without a GPU), the host-side conv planning (tap / phase tables for conv, strided conv,
conv-transpose and their gradients) describes the right arithmetic, and the product never reaches into oracle/."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build(only_library=True)
    from supervised_dispnet_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from supervised_dispnet_amd import _lib
    header = (ROOT / "include" / "dispnet_hip.h").read_text()
    declared = set(re.findall(r"\b(dn_[a-zA-Z0-9_]+)\s*\(", header))
    declared -= {"dn_conv_bn_partial_rows"} - set(_lib.SIGNATURES)   # keep set arithmetic explicit
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libdispnet_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes binding missing for %s" % name
    assert lib.dn_version() == _lib.EXPECTED_ABI
    assert isinstance(lib.dn_last_error(), bytes)


def test_integration_md_stub_matches_the_binding():
    """The reference-side ctypes stub printed in INTEGRATION.md section 1(b) must declare the argument lists the header has
    (round 2 shipped a 9-argument dn_masked_loss_fwd there against the header's 13)."""
    import ctypes as C  # noqa: F401 -- evaluated below
    from supervised_dispnet_amd import _lib
    text = (ROOT / "INTEGRATION.md").read_text()
    found = re.findall(r"lib\.(dn_[a-z0-9_]+)\.argtypes\s*=\s*(\[[^\]]*\])", text)
    assert found, "no argtypes lines in INTEGRATION.md"
    for name, lst in found:
        got = eval(lst, {"C": C})
        want = list(_lib.SIGNATURES[name][1])
        assert got == want, "%s: INTEGRATION.md declares %d arguments, the binding %d" % (name, len(got), len(want))
    m = re.search(r"lib\.dn_version\(\) == (\d+)", text)
    assert m and int(m.group(1)) == _lib.EXPECTED_ABI
    call = re.search(r"lib\.dn_masked_loss_fwd\((.*?)\),\s*\n\s*\"dn_masked_loss_fwd\"", text, re.S)
    assert call and call.group(1).count(",") + 1 == len(_lib.SIGNATURES["dn_masked_loss_fwd"][1])


def test_struct_layout_matches_header(lib):
    """ConvDesc mirrors dn_conv_desc: probe it by letting the library read fields back through the plan dump."""
    from supervised_dispnet_amd._lib import CONV_FWD, ConvDesc
    d = ConvDesc()
    d.kind, d.N, d.IH, d.IW, d.OH, d.OW, d.R, d.S, d.stride, d.pad = CONV_FWD, 2, 10, 12, 10, 12, 3, 3, 1, 1
    d.n_in = 2
    d.in_[0].C, d.in_[1].C = 64, 1
    d.n_out = 1
    d.out[0].C = 48
    d.out[0].data = 16   # any non-null value: the plan only checks presence
    buf = (C.c_int32 * 512)()
    n = lib.dn_debug_conv_plan(C.byref(d), 0, buf, 512)
    assert n > 0, lib.dn_last_error()
    nph, gh, gw, sy, osy, ntot, npad, bn, ndim0, d0, d1, m = list(buf[:12])
    assert (nph, gh, gw, sy, osy, ntot, npad, bn, ndim0, d0, d1, m) == (1, 10, 12, 1, 1, 48, 64, 64, 1, 48, 65, 240)
    ntaps, ooy, oox, nchunks = buf[12], buf[13], buf[14], buf[15]
    assert (ntaps, ooy, oox) == (9, 0, 0) and nchunks == (9 * 64 + 31) // 32 + 1
    assert lib.dn_conv_packed_weight_elems(C.byref(d)) == 64 * nchunks * 32


def _plan(lib, d, for_wgrad=0):
    buf = (C.c_int32 * 2048)()
    n = lib.dn_debug_conv_plan(C.byref(d), for_wgrad, buf, 2048)
    assert n > 0, lib.dn_last_error()
    v = list(buf[:n])
    head = dict(zip(("nphases", "GH", "GW", "sy", "osy", "Ntot", "Npad", "BN", "n_is_dim0", "D0", "D1", "M"), v[:12]))
    pos, phases = 12, []
    for _ in range(head["nphases"]):
        ntaps, ooy, oox, nchunks, woff = v[pos:pos + 5]
        pos += 5
        taps = [tuple(v[pos + 4 * t: pos + 4 * t + 4]) for t in range(ntaps)]
        pos += 4 * ntaps
        phases.append(dict(ntaps=ntaps, ooy=ooy, oox=oox, taps=taps))
    return head, phases


def _simulate(head, phases, x_nhwc, w, OH, OW):
    """What igemm_conv_kernel computes, restated with numpy from the dumped plan (no activation / bias)."""
    N, IH, IW, Cin = x_nhwc.shape
    P = 8
    xp = np.zeros((N, IH + 2 * P, IW + 2 * P, Cin), dtype=np.float64)
    xp[:, P:P + IH, P:P + IW] = x_nhwc
    out = np.zeros((N, OH, OW, head["Ntot"]), dtype=np.float64)
    gy, gx = np.arange(head["GH"]), np.arange(head["GW"])
    for ph in phases:
        acc = np.zeros((N, head["GH"], head["GW"], head["Ntot"]))
        for dy, dx, r, s in ph["taps"]:
            iy, ix = gy * head["sy"] + dy + P, gx * head["sy"] + dx + P
            patch = xp[:, iy][:, :, ix]
            wt = w[:, :, r, s] if head["n_is_dim0"] else w[:, :, r, s].T          # -> [Ntot][Cin]
            acc += patch @ wt.T
        oy, ox = gy * head["osy"] + ph["ooy"], gx * head["osy"] + ph["oox"]
        vy, vx = oy < OH, ox < OW
        out[:, oy[vy][:, None], ox[vx][None, :]] = acc[:, vy][:, :, vx]
    return out


CASES = [  # k, stride, pad, out_pad, H, W
    (3, 1, 1, 0, 7, 9), (7, 2, 3, 0, 12, 10), (5, 2, 2, 0, 9, 11), (3, 2, 1, 0, 7, 8), (1, 1, 0, 0, 5, 6), (1, 2, 0, 0, 6, 7),
    (4, 2, 1, 0, 5, 6), (3, 2, 1, 1, 5, 4),
]


@pytest.mark.parametrize("k,s,p,op,H,W", CASES)
def test_plan_describes_conv_and_its_input_gradient(lib, k, s, p, op, H, W):
    from supervised_dispnet_amd._lib import CONV_DGRAD, CONV_FWD, ConvDesc
    if op:
        pytest.skip("output_padding only applies to conv-transpose")
    N, Cin, Cout = 2, 5, 6
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, dtype=torch.float64)
    y = F.conv2d(x, w, stride=s, padding=p)
    OH, OW = y.shape[2:]
    d = ConvDesc()
    d.kind, d.N, d.IH, d.IW, d.OH, d.OW, d.R, d.S, d.stride, d.pad = CONV_FWD, N, H, W, OH, OW, k, k, s, p
    d.n_in, d.n_out = 1, 1
    d.in_[0].C, d.out[0].C, d.out[0].data = Cin, Cout, 16
    head, phases = _plan(lib, d)
    sim = _simulate(head, phases, x.detach().permute(0, 2, 3, 1).numpy(), w.numpy(), OH, OW)
    np.testing.assert_allclose(sim, y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    g = torch.randn_like(y)
    y.backward(g)
    d2 = ConvDesc()
    d2.kind, d2.N, d2.IH, d2.IW, d2.OH, d2.OW, d2.R, d2.S, d2.stride, d2.pad = CONV_DGRAD, N, OH, OW, H, W, k, k, s, p
    d2.n_in, d2.n_out = 1, 1
    d2.in_[0].C, d2.out[0].C, d2.out[0].data = Cout, Cin, 16
    head, phases = _plan(lib, d2)
    sim = _simulate(head, phases, g.permute(0, 2, 3, 1).numpy(), w.numpy(), H, W)
    np.testing.assert_allclose(sim, x.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)


@pytest.mark.parametrize("k,s,p,op,H,W", [c for c in CASES if c[1] == 2 and c[0] >= 3])
def test_plan_describes_conv_transpose_and_its_input_gradient(lib, k, s, p, op, H, W):
    from supervised_dispnet_amd._lib import CONVT_DGRAD, CONVT_FWD, ConvDesc
    N, Cin, Cout = 2, 5, 6
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cin, Cout, k, k, dtype=torch.float64)
    y_full = F.conv_transpose2d(x, w, stride=s, padding=p, output_padding=op)
    OH, OW = y_full.shape[2] - (1 if op else 0), y_full.shape[3]          # also exercise crop_like on one axis
    y = y_full[:, :, :OH, :OW]
    d = ConvDesc()
    d.kind, d.N, d.IH, d.IW, d.OH, d.OW, d.R, d.S, d.stride, d.pad = CONVT_FWD, N, H, W, OH, OW, k, k, s, p
    d.n_in, d.n_out = 1, 1
    d.in_[0].C, d.out[0].C, d.out[0].data = Cin, Cout, 16
    head, phases = _plan(lib, d)
    assert head["nphases"] == 4
    sim = _simulate(head, phases, x.detach().permute(0, 2, 3, 1).numpy(), w.numpy(), OH, OW)
    np.testing.assert_allclose(sim, y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    g = torch.randn_like(y)
    y.backward(g)
    d2 = ConvDesc()
    d2.kind, d2.N, d2.IH, d2.IW, d2.OH, d2.OW, d2.R, d2.S, d2.stride, d2.pad = CONVT_DGRAD, N, OH, OW, H, W, k, k, s, p
    d2.n_in, d2.n_out = 1, 1
    d2.in_[0].C, d2.out[0].C, d2.out[0].data = Cout, Cin, 16
    head, phases = _plan(lib, d2)
    sim = _simulate(head, phases, g.permute(0, 2, 3, 1).numpy(), w.numpy(), H, W)
    np.testing.assert_allclose(sim, x.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)


def _desc3x3(N, H, W, cins, cout, kind=None, strides_ok=True, ups=None):
    from supervised_dispnet_amd._lib import COMPUTE_F32, CONV_FWD, ConvDesc
    d = ConvDesc()
    d.compute = COMPUTE_F32                      # (the geometry tests below were written for the fp32 layouts; 0 = the library default = F32X3)
    d.kind = CONV_FWD if kind is None else kind
    d.N, d.IH, d.IW, d.OH, d.OW, d.R, d.S, d.stride, d.pad = N, H, W, H, W, 3, 3, 1, 1
    d.n_in = len(cins)
    for i, c in enumerate(cins):
        up = 1 if (ups and ups[i]) else 0
        h, w = H >> up, W >> up
        d.in_[i].data = 4096 * (i + 1)          # fake, 16-byte aligned: host planning only looks at presence / alignment
        d.in_[i].C, d.in_[i].up_shift = c, up
        d.in_[i].stride_c, d.in_[i].stride_w, d.in_[i].stride_h, d.in_[i].stride_n = 1, c, w * c, h * w * c
    d.n_out = 1
    d.out[0].data, d.out[0].C = 1 << 20, cout
    d.out[0].stride_w, d.out[0].stride_h, d.out[0].stride_n = cout, W * cout, H * W * cout
    return d


def test_weight_layout_follows_the_geometry(lib):
    """dn_conv_weight_layout (host only): Winograd for the 3x3/s1/p1 layers with 16-aligned channels, even extents and enough
    tiles -- including a decoder concat with the 1-channel upsampled disparity piece --, the implicit GEMM otherwise; the packed
    size is 16 positions x (K padded per operand to 16, + one chunk of prefetch slack) x (Cout padded to 64)."""
    L = lambda d: lib.dn_conv_weight_layout(C.byref(d))
    assert L(_desc3x3(32, 64, 208, (128,), 128)) == 1
    assert L(_desc3x3(32, 32, 104, (64, 128, 1), 64, ups=(0, 0, 1))) == 1
    assert L(_desc3x3(32, 128, 416, (3,), 64)) == 0            # first layer: 3 channels
    assert L(_desc3x3(32, 128, 416, (16, 1), 16, ups=(0, 1))) == 0   # 16 output channels: a 64-wide tile would be 3/4 padding
    assert L(_desc3x3(2, 4, 6, (256,), 256)) == 0              # 12 tiles: stays on the direct kernel
    assert L(_desc3x3(2, 15, 20, (256,), 256)) == 0            # odd extent
    d = _desc3x3(32, 32, 104, (64, 128, 1), 64, ups=(0, 0, 1))
    assert lib.dn_conv_packed_weight_elems(C.byref(d)) == 16 * (64 + 128 + 16 + 16) * 64
    from supervised_dispnet_amd._lib import CONV_DGRAD
    dh = _desc3x3(32, 32, 104, (1,), 64, kind=CONV_DGRAD)       # input gradient of a disparity head: its own kernel, igemm layout
    assert L(dh) == 0
    d5 = _desc3x3(32, 32, 104, (64,), 64)
    d5.R = d5.S = 5
    d5.pad = 2
    assert L(d5) == 0
    # the weight gradient: Winograd for 64-aligned layers, the thin kernel for the full-resolution 16 / 3-channel layers; both
    # report a workspace
    for dd in (_desc3x3(32, 64, 208, (128,), 128), _desc3x3(32, 128, 416, (16, 1), 16, ups=(0, 1)), _desc3x3(32, 128, 416, (3,), 64)):
        assert lib.dn_conv_wgrad_workspace_bytes(C.byref(dd)) > 0


def test_compute_field_selects_the_packed_layout(lib):
    """dn_conv_desc.compute (host only): the Winograd layers pack their weights per arithmetic -- 1 fp32 fragments (DN_COMPUTE_F32),
    2 bf16-rounded (DN_COMPUTE_BF16), 3 three exact bf16 pieces (DN_COMPUTE_F32X3, 6 bytes per weight = 1.5 floats); layers on the
    implicit-GEMM path ignore the field; 0 (a zeroed descriptor) is the library default = the three-piece arithmetic, unknown values mean fp32."""
    from supervised_dispnet_amd._lib import COMPUTE_BF16, COMPUTE_DEFAULT, COMPUTE_F32, COMPUTE_F32X3
    assert (COMPUTE_DEFAULT, COMPUTE_BF16, COMPUTE_F32X3, COMPUTE_F32) == (0, 1, 2, 3)
    L = lambda d: lib.dn_conv_weight_layout(C.byref(d))
    n32 = None
    for mode, layout in ((COMPUTE_F32, 1), (COMPUTE_BF16, 2), (COMPUTE_F32X3, 3), (77, 1), (COMPUTE_DEFAULT, 3)):
        d = _desc3x3(32, 64, 208, (128,), 128)
        d.compute = mode
        assert L(d) == layout
        n = lib.dn_conv_packed_weight_elems(C.byref(d))
        if n32 is None:
            n32 = n
        assert n == (n32 * 3 // 2 if layout == 3 else n32)
        first = _desc3x3(32, 128, 416, (3,), 64)
        first.compute = mode
        assert L(first) == 0


def test_bad_descriptors_are_rejected_not_crashing(lib):
    from supervised_dispnet_amd._lib import CONV_FWD, ConvDesc
    d = ConvDesc()
    d.kind = 17
    assert lib.dn_conv_packed_weight_elems(C.byref(d)) == -1
    assert b"kind" in lib.dn_last_error()
    d.kind, d.N, d.IH, d.IW, d.OH, d.OW, d.R, d.S, d.stride, d.pad = CONV_FWD, 1, 4, 4, 4, 4, 9, 9, 1, 4
    d.n_in = 1
    d.in_[0].C = 4
    assert lib.dn_conv_packed_weight_elems(C.byref(d)) == -1
    assert lib.dn_conv2d_fwd(None, None) != 0       # null descriptor -> status, no crash


def test_launch_tape_host_logic(lib):
    """dn_tape_*: the bookkeeping that needs no device -- one tape at a time, marks cut segments, pause / resume, a tape being recorded
    cannot be replayed, errors are statuses with a message (no launch happens here: an empty tape replays as a no-op)."""
    t = lib.dn_tape_begin()
    assert t
    assert not lib.dn_tape_begin() and b"another tape" in lib.dn_last_error()       # one at a time
    assert lib.dn_tape_segments(t) == 1 and lib.dn_tape_launches(t) == 0 and lib.dn_tape_fences(t) == 0
    assert lib.dn_tape_mark(t) == 1 and lib.dn_tape_mark(t) == 2
    assert lib.dn_tape_segments(t) == 3
    assert lib.dn_tape_replay(t, -1) != 0 and b"still being recorded" in lib.dn_last_error()
    assert lib.dn_tape_pause(t, 1) == 0
    assert lib.dn_tape_pause(t, 1) != 0                                             # already paused
    assert lib.dn_tape_fence(t, None, None) != 0                                    # not recording while paused
    assert lib.dn_tape_pause(t, 0) == 0
    assert lib.dn_tape_end(t) == 0
    assert lib.dn_tape_end(t) != 0                                                  # not being recorded any more
    assert lib.dn_tape_replay(t, 3) != 0 and b"segment" in lib.dn_last_error()
    t2 = lib.dn_tape_begin()                                                        # the slot is free again
    assert t2 and lib.dn_tape_end(t2) == 0
    lib.dn_tape_free(t2)
    lib.dn_tape_free(t)
    assert lib.dn_tape_segments(None) == -1 and lib.dn_tape_mark(None) == -1


def test_product_never_touches_the_oracle_and_has_no_cpu_path():
    pkg = ROOT / "supervised_dispnet_amd"
    for f in list(pkg.rglob("*.py")) + [ROOT / "train.py", ROOT / "test_disp.py"]:
        if not f.exists():
            continue
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), "%s imports the oracle" % f
        assert "from oracle" not in src and "import oracle" not in src, "%s references the oracle" % f
    import supervised_dispnet_amd.models as models
    net = models.Disp_vgg_BN(with_classifier=False)
    with pytest.raises(RuntimeError, match="HIP path only"):
        net(torch.zeros(1, 3, 64, 96))


def test_state_dict_layout_matches_reference():
    import supervised_dispnet_amd.models as models
    from oracle import nets as ON
    net = models.Disp_vgg_BN(with_classifier=False)
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    want = {k: tuple(v.shape) for k, v in ON.disp_vgg_bn_state_dict().items()}
    assert ours == want
    full = models.Disp_vgg_BN()
    assert len(full.state_dict()) == 125 and sum(p.numel() for p in full.parameters()) == 143516012
    dn = models.DispNetS()
    assert {k: tuple(v.shape) for k, v in dn.state_dict().items()} == {k: tuple(v.shape) for k, v in ON.dispnets_state_dict().items()}
    assert net.alpha == 10 and net.beta == 0.01 and models.Disp_vgg_BN("nyu", with_classifier=False).beta == 0.1
    assert net.only_train_dec is False
