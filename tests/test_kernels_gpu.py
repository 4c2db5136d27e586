"""GPU parity of the individual HIP operators (through the C ABI) against plain PyTorch fp32 on the CPU.

Inputs are pre-rounded to fp16 so that the only differences left are fp32 accumulation order and the
fp16 rounding of stored results: tolerance = 2e-3 of the output's RMS (+ 2e-3 relative).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _h(t):
    return t.half().float()


def _close(got, ref, what, tol=2e-3):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    err = (got - ref).abs()
    lim = tol * rms + tol * ref.abs()
    bad = (err > lim)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max():.3e} rms {rms:.3e}"


def _join():
    """Weight gradients are issued on the engine's side stream (ops.side_run): join before reading them."""
    from fmri_hip.ops import join_side
    join_side()


def _nhwc16(x):   # NCHW fp32 cpu -> NHWC fp16 cuda, channels padded to 8
    from fmri_hip.ops import images_to_nhwc
    return images_to_nhwc(x.to(DEV))


def _from_nhwc(x16, C):
    from fmri_hip.ops import nhwc_to_images
    return nhwc_to_images(x16, C).cpu()


class _G:
    """Minimal FlatGroup stand-in for single-layer tests."""

    def __init__(self, tensors):
        self.views = {k: v.to(DEV).contiguous() for k, v in tensors.items()}
        self.grads = {k: torch.zeros_like(v) for k, v in self.views.items()}
        self.version = 0
        self.device = torch.device(DEV)


def test_image_layout_roundtrip():
    x = _h(torch.randn(3, 3, 10, 7))
    x16 = _nhwc16(x)
    assert x16.shape == (3, 10, 7, 8)
    assert torch.equal(_from_nhwc(x16, 3), x)
    assert (x16[..., 3:] == 0).all()


CONV_CASES = [
    # cin, cout, stride, H, W, N
    (3, 32, 1, 12, 12, 3),       # discriminator conv0 (stride_gan=1), K=75
    (3, 64, 2, 16, 16, 2),       # encoder conv0
    (32, 128, 2, 16, 16, 3),     # two taps per K-step
    (64, 128, 2, 12, 12, 2),
    (128, 256, 2, 8, 8, 5),
    (32, 3, 1, 16, 16, 2),       # decoder last conv, N=3 outputs
    (64, 128, 2, 25, 25, 2),     # odd size (100-px config: 25 -> 13)
    (128, 256, 2, 13, 13, 3),    # 13 -> 7
    (3, 32, 2, 20, 20, 2),       # stride_gan = 2
    (32, 128, 2, 25, 25, 2),     # data gradient = 128 -> 32 transposed conv on odd sizes (igemm_tc32, masked tiles)
]


@pytest.mark.parametrize("cin,cout,stride,H,W,N", CONV_CASES)
def test_conv_forward_dgrad_wgrad(cin, cout, stride, H, W, N):
    from fmri_hip.ops import ConvLayer, ACT_NONE
    torch.manual_seed(cin * 1000 + cout + H)
    w = _h(torch.randn(cout, cin, 5, 5) * 0.1)
    b = torch.randn(cout) * 0.1
    x = _h(torch.randn(N, cin, H, W))
    g = _G({"w": w, "b": b})
    layer = ConvLayer(g, "w", "b", "conv", cin, cout, 5, stride, 2)
    x16 = _nhwc16(x)
    y16 = layer.forward(x16, ACT_NONE)
    ref = F.conv2d(x, w, b, stride, 2)
    _close(_from_nhwc(y16, cout), ref, "conv fwd")
    if cout % 8:
        assert (y16[..., cout:] == 0).all()
    # cotangent
    dy = _h(torch.randn_like(ref))
    dy16 = _nhwc16(dy)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, 2).backward(dy)
    dx16 = layer.dgrad(dy16, H, W)
    _close(_from_nhwc(dx16, cin), xr.grad, "conv dgrad")
    layer.wgrad(x16, dy16, 4.0)          # scale 4 -> grad accumulates dW/4
    _join()
    _close(g.grads["w"].cpu() * 4.0, wr.grad, "conv wgrad", tol=3e-3)


C5_CASES = [
    # cin, cout, H, W, N          conv k5 s2 p2 without bias -> csrc/igemm_c5.hip
    (32, 128, 64, 64, 3),         # discriminator.conv.1: one 32-channel sub-chunk, 8 x 16 tiles
    (64, 128, 32, 32, 2),         # encoder.conv.1: two sub-chunks
    (128, 256, 16, 16, 5),        # 8 x 8 tiles of two images, odd image count, two channel blocks
    (256, 256, 16, 16, 2),        # eight sub-chunks
    (128, 256, 25, 25, 2),        # odd size 25 -> 13 (100-px config): partial tiles, odd last input row / column
    (128, 256, 13, 13, 3),        # 13 -> 7: partial 8 x 8 tiles
    (64, 128, 50, 50, 1),         # 50 -> 25: two tiles per row, four per column
    (32, 128, 20, 36, 2),         # rectangular: 10 x 18 outputs
    (96, 128, 12, 12, 2),         # three sub-chunks (odd count: ring stage parity)
]


@pytest.mark.parametrize("cin,cout,H,W,N", C5_CASES)
def test_conv_stride2_window_kernel(cin, cout, H, W, N):
    """csrc/igemm_c5.hip (window-resident stride-2 convolution) against the fp32 convolution, and against the tap-list
    kernel it replaces (FMRI_C5 is read once per process, so that comparison goes through the label only)."""
    from fmri_hip.ops import ConvLayer, ACT_NONE, igemm_route, MODE_CONV
    torch.manual_seed(cin * 7 + cout + H * 3 + W + N)
    w = _h(torch.randn(cout, cin, 5, 5) * 0.05)
    x = _h(torch.randn(N, cin, H, W))
    g = _G({"w": w})
    layer = ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
    Ho, Wo = layer.out_hw(H, W)
    assert "igemm_c5" in igemm_route(N, H, W, cin, Ho, Wo, cout, cout, 5, 2, 2, MODE_CONV, ACT_NONE, False, 1, 128,
                                     layer.pw_f.buf.numel())
    y16 = layer.forward(_nhwc16(x), ACT_NONE)
    ref = F.conv2d(x, w, None, 2, 2)
    _close(_from_nhwc(y16, cout), ref, "conv s2 (window kernel)")


def test_conv0_wgrad_large_batch_routes_to_narrow_kernel():
    """discriminator.conv.0-shaped weight gradient with enough 8x8 tiles (>= 32768) to take the wave-private window
    kernel (csrc/wgrad_narrow.hip); the small-batch cases above stay on the generic kernel."""
    from fmri_hip.ops import ConvLayer
    torch.manual_seed(5)
    N, H = 512, 64
    w = _h(torch.randn(32, 3, 5, 5) * 0.1)
    x = _h(torch.randn(N, 3, H, H))
    dy = _h(torch.randn(N, 32, H, H) * 0.1)
    g = _G({"w": w})
    layer = ConvLayer(g, "w", None, "conv", 3, 32, 5, 1, 2)
    layer.wgrad(_nhwc16(x), _nhwc16(dy), 2.0)
    _join()
    ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=1, padding=2)
    _close(g.grads["w"].cpu() * 2.0, ref, "conv0 wgrad (narrow kernel)", tol=3e-3)


def test_conv0_bias_gradient_from_the_narrow_wgrad_kernel():
    """wgrad(..., bias_too=True): the bias gradient is the spare column of csrc/wgrad_narrow.hip (taps x ones)."""
    from fmri_hip.ops import ConvLayer
    torch.manual_seed(6)
    N, H = 512, 64
    w = _h(torch.randn(32, 3, 5, 5) * 0.1)
    x = _h(torch.randn(N, 3, H, H))
    dy = _h(torch.randn(N, 32, H, H) * 0.1)
    g = _G({"w": w, "b": torch.zeros(32)})
    layer = ConvLayer(g, "w", "b", "conv", 3, 32, 5, 1, 2)
    layer.wgrad(_nhwc16(x), _nhwc16(dy), 2.0, bias_too=True)
    _join()
    ref = dy.double().sum((0, 2, 3)).float()
    _close(g.grads["b"].cpu() * 2.0, ref, "conv0 bias gradient (narrow kernel column)", tol=3e-3)
    refw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=1, padding=2)
    _close(g.grads["w"].cpu() * 2.0, refw, "conv0 wgrad (narrow kernel)", tol=3e-3)
    # small batch: the generic kernel has no such column, the layer reduces dy itself
    g2 = _G({"w": w, "b": torch.zeros(32)})
    layer2 = ConvLayer(g2, "w", "b", "conv", 3, 32, 5, 1, 2)
    layer2.wgrad(_nhwc16(x[:3]), _nhwc16(dy[:3]), 1.0, bias_too=True)
    _join()
    _close(g2.grads["b"].cpu(), dy[:3].double().sum((0, 2, 3)).float(), "conv0 bias gradient (reduction)", tol=3e-3)


@pytest.mark.parametrize("H,N", [(32, 3), (25, 2)])
def test_dgrad_epilogue_relu_backward(H, N):
    """fmri_epilogue.act_y: the 128 -> 32 channel data gradient (csrc/igemm_tc32.hip) masks with the ReLU of the layer
    below; bit-identical to the plain data gradient followed by fmri_act_bwd."""
    from fmri_hip.ops import ConvLayer, act_backward, ACT_RELU
    torch.manual_seed(H + N)
    w = _h(torch.randn(128, 32, 5, 5) * 0.05)
    g = _G({"w": w})
    layer = ConvLayer(g, "w", None, "conv", 32, 128, 5, 2, 2)
    Ho, Wo = layer.out_hw(H, H)
    dy = (torch.randn(N, Ho, Wo, 128, device=DEV) * 0.5).half()
    y0 = torch.relu(torch.randn(N, H, H, 32, device=DEV)).half()
    plain = layer.dgrad(dy, H, H)
    assert not layer.act_applied
    masked = layer.dgrad(dy, H, H, relu_y=y0)
    assert layer.act_applied
    assert torch.equal(masked, act_backward(y0, plain, ACT_RELU))


DECONV_CASES = [
    # cin, cout, H, out_pad, N
    (256, 256, 8, 1, 3),
    (256, 128, 16, 1, 2),
    (128, 32, 16, 1, 2),
    (256, 256, 13, 0, 2),        # 100-px config first block: 13 -> 25
    (128, 64, 25, 1, 2),         # 25 -> 50
    (128, 32, 13, 0, 3),         # igemm_tc32 with partial tiles and output_padding = 0
    (256, 256, 7, 0, 9),         # igemm_tc5w<8>: 7 -> 13, four images per tile, the last tile holds one
    (256, 128, 8, 1, 8),         # igemm_tc5w<8>: two full tiles
]


@pytest.mark.parametrize("cin,cout,H,op,N", DECONV_CASES)
def test_deconv_forward_dgrad_wgrad(cin, cout, H, op, N):
    from fmri_hip.ops import ConvLayer, ACT_NONE
    torch.manual_seed(cin + cout + H)
    w = _h(torch.randn(cin, cout, 5, 5) * 0.05)
    x = _h(torch.randn(N, cin, H, H))
    g = _G({"w": w})
    layer = ConvLayer(g, "w", None, "deconv", cin, cout, 5, 2, 2, op)
    x16 = _nhwc16(x)
    y16 = layer.forward(x16, ACT_NONE)
    ref = F.conv_transpose2d(x, w, None, 2, 2, output_padding=op)
    assert y16.shape[1] == ref.shape[2]
    _close(_from_nhwc(y16, cout), ref, "deconv fwd")
    dy = _h(torch.randn_like(ref))
    dy16 = _nhwc16(dy)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv_transpose2d(xr, wr, None, 2, 2, output_padding=op).backward(dy)
    dx16 = layer.dgrad(dy16, H, H)
    _close(_from_nhwc(dx16, cin), xr.grad, "deconv dgrad")
    layer.wgrad(x16, dy16, 1.0)
    _join()
    _close(g.grads["w"].cpu(), wr.grad, "deconv wgrad", tol=3e-3)


EPI_STAT_CASES = [
    # kind, cin, cout, stride, H, out_pad, N, groups, expect the epilogue
    ("conv", 64, 128, 2, 16, 0, 4, 1, True),        # igemm_c5, two images per tile
    ("conv", 32, 128, 2, 32, 0, 8, 1, True),        # igemm_c5, 8 x 16 tiles, one sub-chunk
    ("conv", 3, 64, 2, 16, 0, 4, 1, True),          # generic kernel: 2 K-steps, 64-channel tile
    ("conv", 128, 256, 2, 32, 0, 200, 1, True),     # igemm_c5, many tiles, two channel blocks
    ("conv", 128, 256, 2, 13, 0, 3, 1, True),       # igemm_c5, partial tiles, odd image count
    ("conv", 64, 128, 2, 50, 0, 2, 2, True),        # igemm_c5, two BatchNorm batches of one image
    ("conv", 256, 256, 2, 16, 0, 6, 3, True),       # igemm_c5, three batches of two images (= one tile each)
    ("conv", 256, 256, 2, 16, 0, 3, 3, False),      # a two-image tile would hold two batches: no epilogue
    ("deconv", 256, 128, 2, 16, 1, 4, 2, True),     # igemm_tc5, double-buffered window, two BatchNorm batches
    ("deconv", 256, 256, 2, 8, 1, 4, 2, True),      # igemm_tc5, two images per tile
    ("deconv", 256, 256, 2, 8, 1, 8, 2, True),      # igemm_tc5w<8>, four images per tile = one tile per BatchNorm batch
    ("deconv", 256, 128, 2, 7, 0, 12, 1, True),     # igemm_tc5w<8>, 7 -> 13, three tiles
    ("deconv", 256, 256, 2, 8, 1, 6, 2, False),     # a tile would hold images of two batches: no epilogue
    ("deconv", 256, 256, 2, 13, 0, 3, 1, True),     # partial tiles, output_padding = 0
    ("deconv", 128, 64, 2, 25, 1, 2, 2, True),      # 64-channel tile
    ("deconv", 256, 128, 2, 16, 1, 6, 1, True),     # igemm_tc5w<16>, one class per block (few tiles), statistics
    ("deconv", 128, 128, 2, 32, 1, 12, 3, True),    # the same with four tiles per image, three BatchNorm batches
    ("deconv", 256, 128, 2, 16, 1, 160, 1, True),   # igemm_tc5w<16>, four classes per block (many tiles)
]


@pytest.mark.parametrize("kind,cin,cout,stride,H,op,N,groups,expect", EPI_STAT_CASES)
def test_conv_epilogue_batchnorm_statistics(kind, cin, cout, stride, H, op, N, groups, expect):
    """fmri_igemm_ep: the per-block rows of sum x / sum x^2 the contraction's epilogue writes fold to the statistics of
    the stored output, per BatchNorm batch, and BatchNorm.forward gives the same result from them as from its own pass."""
    from fmri_hip.ops import ConvLayer, BatchNorm
    torch.manual_seed(cin + cout + H + N)
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = _G({"w": _h(torch.randn(*shape) * 0.05), "bn.weight": torch.rand(cout) + 0.5, "bn.bias": torch.randn(cout) * 0.1})
    g.bufs = {"bn.running_mean": torch.zeros(cout, device=DEV), "bn.running_var": torch.ones(cout, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    layer = ConvLayer(g, "w", None, kind, cin, cout, 5, stride if kind == "conv" else 2, 2, op)
    x16 = torch.randn(N, H, H, layer.cinp, device=DEV).half()
    if layer.cinp != cin:
        x16[..., cin:] = 0
    y16 = layer.forward(x16, bn_groups=groups)
    B = N // groups
    assert (layer.take_stats(0) is not None) == expect
    # the stored output itself (the one-class-per-block tc5w instantiation once stored wrong dwords with statistics on:
    # the gfx950 store-data hazard)
    xr, wr = x16[..., :cin].float().permute(0, 3, 1, 2), g.views["w"].float()
    ref = (F.conv2d(xr, wr, None, stride, 2) if kind == "conv" else
           F.conv_transpose2d(xr, wr, None, 2, 2, output_padding=op)).permute(0, 2, 3, 1)
    _close(y16[..., :cout], ref, "stored output")
    if not expect:
        return
    bn = BatchNorm(g, "bn.", layer.coutp) if layer.coutp == cout else None
    for gi in range(groups):
        part = layer.take_stats(gi)
        yg = y16[gi * B:(gi + 1) * B].float().reshape(-1, layer.coutp)
        got = part.double().sum(0).cpu()
        ref = torch.stack([yg.double().sum(0), (yg.double() ** 2).sum(0)]).cpu()
        scale = (yg.double() ** 2).sum(0).sqrt().cpu().clamp_min(1e-6)
        assert ((got[0] - ref[0]).abs() <= 2e-5 * scale * np.sqrt(yg.shape[0])).all(), "sum x"
        assert ((got[1] - ref[1]).abs() <= 2e-5 * ref[1].abs() + 1e-6).all(), "sum x^2"
        if bn is not None:
            a, sva = bn.forward(y16[gi * B:(gi + 1) * B], True, 0, stat_acc=part)
            b, svb = bn.forward(y16[gi * B:(gi + 1) * B], True, 0)
            assert torch.allclose(sva.mean, svb.mean, rtol=1e-5, atol=1e-6)
            assert torch.allclose(sva.rstd, svb.rstd, rtol=1e-5, atol=1e-6)
            assert ((a.float() - b.float()).abs() <= 2e-3 * b.float().abs().clamp_min(1.0)).all()   # <= 1 fp16 ulp
    # deterministic: a second run writes bit-identical rows
    first = layer._stat_part[:groups, :layer._stat_rows].clone()
    layer.forward(x16, bn_groups=groups)
    assert torch.equal(first, layer._stat_part[:groups, :layer._stat_rows])


EPI_BWD_CASES = [
    # kind, cin, cout, H (layer input), out_pad, N, groups
    ("conv", 128, 256, 16, 0, 4, 1),        # conv dgrad = igemm_tc5<128>, two images per tile
    ("conv", 128, 256, 32, 0, 4, 2),        # igemm_tc5<128>, double-buffered window, two cotangent streams
    ("conv", 64, 128, 32, 0, 6, 3),         # igemm_tc5<64>, three groups (decoder entries)
    ("conv", 128, 256, 26, 0, 3, 1),        # partial tiles
    ("deconv", 256, 128, 16, 1, 4, 2),      # deconv dgrad = stride-2 convolution: igemm_c5<16> (the wide form has no BnBwdEpi)
    ("deconv", 128, 64, 16, 1, 128, 2),     # ... 64 cotangent channels, persistent blocks over 2 tiles
    ("deconv", 256, 256, 8, 1, 8, 2),       # igemm_c5<8>: two images per tile, two tiles per statistics group
    ("deconv", 256, 256, 8, 1, 4, 2),       # two images per group = one tile per group
    ("deconv", 128, 32, 32, 1, 6, 1),       # one 32-channel sub-chunk per tile (13 K-steps), four tiles per image
    ("deconv", 128, 32, 13, 0, 3, 3),       # partial tiles, three groups of one image
]


@pytest.mark.parametrize("kind,cin,cout,H,op,N,groups", EPI_BWD_CASES)
def test_dgrad_epilogue_batchnorm_backward(kind, cin, cout, H, op, N, groups):
    """fmri_epilogue.bn_x: the data gradient's epilogue applies the ReLU mask of the BatchNorm block in front of the
    layer and emits its backward sums; BatchNorm.backward / backward2 from those rows equals the separate reduction."""
    from fmri_hip.ops import ConvLayer, BatchNorm
    torch.manual_seed(cin + cout + H + N)
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = _G({"w": _h(torch.randn(*shape) * 0.05), "bn.weight": torch.rand(cin) + 0.5, "bn.bias": torch.randn(cin) * 0.3})
    g.bufs = {"bn.running_mean": torch.zeros(cin, device=DEV), "bn.running_var": torch.ones(cin, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    layer = ConvLayer(g, "w", None, kind, cin, cout, 5, 2, 2, op)
    bn = BatchNorm(g, "bn.", cin)
    B = N // groups
    # forward of the BatchNorm block in front of the layer: `groups` calls over B images each, two of them on the SAME
    # saved tensor when groups >= 2 (cotangent streams), the last on its own
    nfw = 1 if groups < 3 else 2
    raw = torch.randn(nfw * B, H, H, cin, device=DEV).half()
    svs = [bn.forward(raw[i * B:(i + 1) * B], True, 0)[1] for i in range(nfw)]
    fwd_of = [0] * groups if groups < 3 else [0, 0, 1]
    Ho, Wo = layer.out_hw(H, H)
    dy = (torch.randn(N, Ho, Wo, layer.coutp, device=DEV) * 0.5).half()
    grp = [(fwd_of[e] * B, svs[fwd_of[e]]) for e in range(groups)]
    d_plain = layer.dgrad(dy, H, H)
    d_mask = layer.dgrad(dy, H, H, bn_bwd=dict(bn=bn, x=raw, groups=grp))
    stat = layer.take_bwd_stats()
    assert stat is not None
    rows = lambda t, e: t[e * B:(e + 1) * B]
    e = 0
    while e < groups:
        xg = rows(raw, fwd_of[e])
        if e + 1 < groups and fwd_of[e + 1] == fwd_of[e]:
            a, sa = bn.backward2(xg, d_plain[e * B:(e + 2) * B], svs[fwd_of[e]], True)
            b, sb = bn.backward2(xg, d_mask[e * B:(e + 2) * B], svs[fwd_of[e]], True, stat=stat, stat_group=e)
            e += 2
        else:
            a, sa = bn.backward(xg, rows(d_plain, e), svs[fwd_of[e]], True)
            b, sb = bn.backward(xg, rows(d_mask, e), svs[fwd_of[e]], True, stat=stat, stat_group=e)
            e += 1
        ref = sa.double().cpu()
        tol = 2e-4 * ref.abs().max() + 1e-4
        assert ((sb.double().cpu() - ref).abs() <= tol).all(), (float((sb.double().cpu() - ref).abs().max()), float(tol))
        assert ((a.float() - b.float()).abs() <= 2e-3 * a.float().abs().clamp_min(float(a.float().abs().mean()))).all()
    # the masked output is the plain one with the ReLU-off elements zeroed
    gamma, beta = g.views["bn.weight"], g.views["bn.bias"]
    for e in range(groups):
        sv = svs[fwd_of[e]]
        xh = (rows(raw, fwd_of[e]).float() - sv.mean) * sv.rstd
        on = (xh * gamma + beta) > 0
        assert torch.equal(rows(d_mask, e), torch.where(on, rows(d_plain, e), torch.zeros_like(rows(d_plain, e))))


def test_bn_backward_epilogue_request_is_never_answered_with_forward_statistics():
    """fmri_igemm_ep with ``bn_x`` set (BatchNorm-BACKWARD epilogue) and a row capacity the kernel that has this epilogue
    cannot meet (igemm_c5<16>: 4 rows) but the wide kernel's FORWARD-statistics rows would (igemm_c5w<16>: 2 rows): the
    call must come back with *ep_done = 0, no row written and the plain data gradient in ``out`` -- never with
    sum x / sum x^2 rows that the caller would fold as (sum g, sum g*xhat).  (Round-3 advisor finding: the wide kernels
    were gated on the already-cleared copy of the request.)"""
    from fmri_hip import ops
    from fmri_hip.ops import ConvLayer, BatchNorm
    torch.manual_seed(5)
    cin, cout, H, N, G = 256, 128, 16, 4, 2
    g = _G({"w": _h(torch.randn(cin, cout, 5, 5) * 0.05), "bn.weight": torch.rand(cin) + 0.5, "bn.bias": torch.randn(cin) * 0.3})
    g.bufs = {"bn.running_mean": torch.zeros(cin, device=DEV), "bn.running_var": torch.ones(cin, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    layer = ConvLayer(g, "w", None, "deconv", cin, cout, 5, 2, 2, 1)
    bn = BatchNorm(g, "bn.", cin)
    B = N // G
    raw = torch.randn(B, H, H, cin, device=DEV).half()
    sv = bn.forward(raw, True, 0)[1]
    Ho, Wo = layer.out_hw(H, H)
    dy = (torch.randn(N, Ho, Wo, layer.coutp, device=DEV) * 0.5).half()
    plain = layer.dgrad(dy, H, H)
    cap = 3
    part = torch.full((G, cap, 2, cin), float("nan"), dtype=torch.float32, device=DEV)
    out = torch.empty_like(plain)
    gamma, beta, _, _ = bn._params()
    r = ops.run_igemm(dy, layer.pw_d, out, None, N, Ho, Wo, layer.coutp, H, H, layer.cinp, layer.cin, 5, 2, 2,
                      ops.MODE_CONV, ops.ACT_NONE, False, 1, 0, layer.t_in, stats=(part, cap, B),
                      bn_bwd=dict(x=raw, gamma=gamma, beta=beta, relu=True, groups=[(0, sv)] * G))
    torch.cuda.synchronize()
    assert r == 0, r
    assert bool(torch.isnan(part).all()), "statistics rows were written although the request was declined"
    assert torch.equal(out, plain)
    # with room for its rows the epilogue is honoured
    part = torch.full((G, 8, 2, cin), float("nan"), dtype=torch.float32, device=DEV)
    r = ops.run_igemm(dy, layer.pw_d, out, None, N, Ho, Wo, layer.coutp, H, H, layer.cinp, layer.cin, 5, 2, 2,
                      ops.MODE_CONV, ops.ACT_NONE, False, 1, 0, layer.t_in, stats=(part, 8, B),
                      bn_bwd=dict(x=raw, gamma=gamma, beta=beta, relu=True, groups=[(0, sv)] * G))
    torch.cuda.synchronize()
    assert r == 4, r
    assert not bool(torch.isnan(part[:, :4]).any())


def test_conv_bias_relu_tanh_epilogues():
    from fmri_hip.ops import ConvLayer, ACT_RELU, ACT_TANH
    torch.manual_seed(5)
    w = _h(torch.randn(32, 3, 5, 5) * 0.2)
    b = torch.randn(32) * 0.2
    x = _h(torch.randn(2, 3, 9, 11))
    g = _G({"w": w, "b": b})
    layer = ConvLayer(g, "w", "b", "conv", 3, 32, 5, 1, 2)
    x16 = _nhwc16(x)
    _close(_from_nhwc(layer.forward(x16, ACT_RELU), 32), F.relu(F.conv2d(x, w, b, 1, 2)), "bias+relu")
    _close(_from_nhwc(layer.forward(x16, ACT_TANH), 32), torch.tanh(F.conv2d(x, w, b, 1, 2)), "bias+tanh")


DENSE_CASES = [
    # M, K, N, in_perm, out_perm
    (6, 1024, 256, None, None),          # fused heads
    (12, 512, 1, None, None),            # discriminator fc.3
    (5, 4096, 1024, None, None),         # cognitive encoder fc1 (split-K)
    (5, 3620, 1024, None, None),         # BOLD5000-shaped voxel count (not a multiple of 8)
    (4, 16384, 1024, (256, 64), None),   # encoder fc.0 with (C,H,W)->(H,W,C) flatten
    (9, 16384, 512, (256, 64), None),    # discriminator fc.0
    (4, 128, 16384, None, (256, 64)),    # decoder fc.0
    (3, 128, 512, None, None),           # WAE latent discriminator layer
]


@pytest.mark.parametrize("M,K,N,in_perm,out_perm", DENSE_CASES)
def test_dense_forward_dgrad_wgrad(M, K, N, in_perm, out_perm):
    from fmri_hip.ops import DenseLayer, ACT_NONE, pad8, rows_to_f16
    torch.manual_seed(M + K + N)
    w = _h(torch.randn(N, K) / np.sqrt(K))
    # the only output-permuted layer of the model (decoder.fc.0) has no bias
    b = torch.zeros(N) if out_perm else torch.randn(N) * 0.1
    x = _h(torch.randn(M, K))
    g = _G({"w": w, "b": b})
    layer = DenseLayer(g, "w", "b", K, N, in_perm=in_perm, out_perm=out_perm)

    def to_engine_in(t):          # reference (C,HW) feature order -> engine (HW,C)
        if in_perm:
            C, HW = in_perm
            return t.reshape(-1, C, HW).transpose(1, 2).reshape(-1, K)
        return t

    def from_engine_out(t):
        if out_perm:
            C, HW = out_perm
            return t.reshape(-1, HW, C).transpose(1, 2).reshape(-1, N)
        return t

    def to_engine_out(t):
        if out_perm:
            C, HW = out_perm
            return t.reshape(-1, C, HW).transpose(1, 2).reshape(-1, N)
        return t

    def from_engine_in(t):
        if in_perm:
            C, HW = in_perm
            return t.reshape(-1, HW, C).transpose(1, 2).reshape(-1, K)
        return t

    x16 = rows_to_f16(to_engine_in(x).contiguous().to(DEV))
    ref = F.linear(x, w, b)
    o16, o32 = layer.forward(x16, ACT_NONE, want16=True, want32=True)
    _close(from_engine_out(o32.cpu()), ref, "dense fwd fp32", tol=1e-3)
    _close(from_engine_out(o16[:, :N].float().cpu()), ref, "dense fwd fp16")
    dy = _h(torch.randn(M, N))
    dy16 = rows_to_f16(to_engine_out(dy).contiguous().to(DEV))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.linear(xr, wr, None).backward(dy)
    dx16, _ = layer.dgrad(dy16)
    _close(from_engine_in(dx16[:, :K].float().cpu()), xr.grad, "dense dgrad")
    _, dx32 = layer.dgrad(dy16, want32=True)
    _close(from_engine_in(dx32.cpu()), xr.grad, "dense dgrad fp32", tol=1e-3)
    layer.wgrad(x16, dy16, 2.0)
    _join()
    _close(g.grads["w"].cpu() * 2.0, wr.grad, "dense wgrad", tol=3e-3)
    if not out_perm:
        layer.bias_grad(dy16, 2.0)
        _close(g.grads["b"].cpu() * 2.0, dy.sum(0), "dense bias grad")


@pytest.mark.parametrize("case", ["window (stride 2, 128 rows)", "narrow (5x5 stride 1, 32 x 8)", "generic (dense)"])
def test_gated_weight_gradient_launch(case):
    """fmri_wgrad_if: with the device flag at 0 the launch leaves its output untouched (whatever kernel the geometry is
    routed to), with the flag at 1 it is fmri_wgrad."""
    from fmri_hip import ops
    torch.manual_seed(3)
    if case.startswith("window"):
        N, A, Bc, k, s, pad, Hq, Yc = 8, 128, 64, 5, 2, 2, 16, 8
    elif case.startswith("narrow"):
        N, A, Bc, k, s, pad, Hq, Yc = 2048, 32, 8, 5, 1, 2, 16, 16          # >= 32768 8x8 tiles: the narrow kernel
    else:
        N, A, Bc, k, s, pad, Hq, Yc = 64, 256, 128, 1, 1, 0, 1, 1
    P = torch.randn(N, Yc, Yc, A, device=DEV).half()
    Q = torch.randn(N, Hq, Hq, Bc, device=DEV).half()
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    was = ops._DET["on"]
    ops.set_deterministic(True)           # slab outputs (plain stores): two launches give the same bits
    try:
        ref, _ = ops.run_wgrad(P, Q, N, Yc, Yc, A, Hq, Hq, Bc, k, s, pad)
        on, _ = ops.run_wgrad(P, Q, N, Yc, Yc, A, Hq, Hq, Bc, k, s, pad, gate=flag)
        hold = {}
        ops.run_wgrad(P, Q, N, Yc, Yc, A, Hq, Hq, Bc, k, s, pad, hold=hold)          # allocates the persistent buffer
        buf = [v for kk, v in hold.items() if kk != "busy"][0]
        buf.fill_(7.0)
        hold["busy"] = False
        flag.zero_()
        off, _ = ops.run_wgrad(P, Q, N, Yc, Yc, A, Hq, Hq, Bc, k, s, pad, hold=hold, gate=flag)
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic(was)
    cols = k * k * Bc                      # (the columns behind them are padding nobody writes or reads)
    assert float(ref[..., :cols].abs().max()) > 0
    assert torch.equal(ref[..., :cols], on[..., :cols]), "flag = 1 differs from the unconditional launch"
    assert off.data_ptr() == buf.data_ptr() and bool((off == 7.0).all()), "flag = 0 wrote into the output"


@pytest.mark.parametrize("R,C", [(1, 8), (100, 130), (64, 64), (129, 72), (512, 16384)])
def test_transpose_f16(R, C):
    """fmri_transpose_f16 against torch on ragged shapes; the padding of the destination stays untouched."""
    from fmri_hip import lib
    rows, lds, ldd = (R + 63) // 64 * 64, (C + 7) // 8 * 8 + 8, (R + 7) // 8 * 8 + 16
    src = torch.zeros(rows, lds, dtype=torch.float16, device=DEV)
    src[:R, :C] = torch.randn(R, C, device=DEV).half()
    dst = torch.full((C + 3, ldd), 7.0, dtype=torch.float16, device=DEV)
    lib.call("fmri_transpose_f16", src.data_ptr(), dst.data_ptr(), R, C, rows, lds, ldd)
    torch.cuda.synchronize()
    assert torch.equal(dst[:C, :R], src[:R, :C].t())
    r8 = (R + 7) // 8 * 8
    assert bool((dst[:C, R:r8] == 0).all()) and bool((dst[:C, r8:] == 7.0).all()) and bool((dst[C:] == 7.0).all())


@pytest.mark.parametrize("kind,cin,cout", [("conv", 3, 64), ("conv", 64, 128), ("conv", 128, 256), ("conv", 32, 128),
                                           ("deconv", 256, 256), ("deconv", 256, 128), ("deconv", 128, 32)])
def test_conv_class_blocks_are_tap_transposes(kind, cin, cout):
    """The four parity-class blocks of a stride-2 layer's fp16 weight made by fmri_transpose_f16_batch from the
    single-block copy (one transpose per tap) == the blocks fmri_pack_weight makes from the fp32 master, bit for bit,
    padding included."""
    from fmri_hip import lib
    from fmri_hip.ops import ConvLayer
    torch.manual_seed(cin + cout)
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = _G({"w": torch.randn(*shape) * 0.05})
    layer = ConvLayer(g, "w", None, kind, cin, cout, 5, 2, 2, 1 if kind == "deconv" else 0)
    cls = layer.pw_d if kind == "conv" else layer.pw_f
    assert getattr(cls, "taps_of", None) is not None and len(cls.specs) == 4
    got = cls.get().clone()
    ref = torch.zeros_like(got)
    for item in cls._items():
        lib.call("fmri_pack_weight", item[0], ref.data_ptr() + (item[1] - cls.buf.data_ptr()), *item[2:])
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,K,N,in_perm,out_perm", DENSE_CASES)
def test_dense_second_orientation_is_the_transpose(M, K, N, in_perm, out_perm):
    """The data-gradient copy of a dense weight made by fmri_transpose_f16 from the forward copy == the one
    fmri_pack_weight makes from the fp32 master (bit for bit, padding included)."""
    from fmri_hip import lib, ops
    from fmri_hip.ops import DenseLayer
    torch.manual_seed(K + N)
    g = _G({"w": torch.randn(N, K) / np.sqrt(K), "b": torch.zeros(N)})
    layer = DenseLayer(g, "w", "b", K, N, in_perm=in_perm, out_perm=out_perm)
    if getattr(layer.pw_d, "transpose_of", None) is None:
        pytest.skip("this layer's two orientations differ in padding (K % 8 != 0): packed from the master")
    got = layer.pw_d.get().clone()
    ref = torch.zeros_like(got)
    for item in layer.pw_d._items():
        lib.call("fmri_pack_weight", item[0], ref.data_ptr(), *item[2:])
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,C", [(3 * 64 * 64, 32), (600, 128), (2 * 13 * 13, 256), (7, 1024), (5, 16384), (12, 512)])
def test_batchnorm_forward_backward(M, C):
    from fmri_hip.ops import BatchNorm
    torch.manual_seed(M + C)
    x = _h(torch.randn(M, C) * 1.5 + 0.3)
    gamma = 1 + 0.2 * torch.randn(C)
    beta = 0.1 * torch.randn(C)

    class G2(_G):
        pass
    g = G2({"bn.weight": gamma, "bn.bias": beta})
    g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    bn = BatchNorm(g, "bn.", C)
    x16 = x.half().to(DEV)
    y16, sv = bn.forward(x16, relu=True, updates=2)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.9, 1e-5))
    F.batch_norm(x, rm, rv, gamma, beta, True, 0.9, 1e-5)      # second running-stat update
    _close(y16.float().cpu(), yr, "bn fwd")
    _close(g.bufs["bn.running_mean"].cpu(), rm, "running mean", tol=1e-3)
    _close(g.bufs["bn.running_var"].cpu(), rv, "running var", tol=1e-3)
    assert int(g.bufs["bn.num_batches_tracked"]) == 2
    dy = _h(torch.randn(M, C))
    yr.backward(dy)
    dx16, _ = bn.backward(x16, dy.half().to(DEV), sv, relu=True, param_scale=8.0)
    _close(dx16.float().cpu(), xr.grad, "bn dx", tol=3e-3)
    _close(g.grads["bn.weight"].cpu() * 8.0, gr.grad, "bn dgamma", tol=3e-3)
    _close(g.grads["bn.bias"].cpu() * 8.0, br.grad, "bn dbeta", tol=3e-3)


@pytest.mark.parametrize("M,C", [(3 * 32 * 32, 128), (600, 256), (37, 64)])
def test_batchnorm_backward_two_streams(M, C):
    """``BatchNorm.backward2`` (two stacked cotangent streams, the forward tensor read once) against autograd on the two
    cotangents separately; gamma / beta gradients come from stream A only."""
    from fmri_hip.ops import BatchNorm
    torch.manual_seed(M * 3 + C)
    x = _h(torch.randn(M, C) * 1.2 - 0.2)
    gamma = 1 + 0.2 * torch.randn(C)
    beta = 0.1 * torch.randn(C)
    g = _G({"bn.weight": gamma, "bn.bias": beta})
    g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    bn = BatchNorm(g, "bn.", C)
    x16 = x.half().to(DEV)
    _, sv = bn.forward(x16, relu=True, updates=0)
    dya, dyb = _h(torch.randn(M, C)), _h(torch.randn(M, C) * 0.5)
    ref = []
    for dy in (dya, dyb):
        xr = x.clone().requires_grad_(True)
        gr = gamma.clone().requires_grad_(True)
        br = beta.clone().requires_grad_(True)
        F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.9, 1e-5)).backward(dy)
        ref.append((xr.grad, gr.grad, br.grad))
    dy2 = torch.cat([dya, dyb], 0).half().to(DEV)
    dx2, sums = bn.backward2(x16, dy2, sv, relu=True, param_scale=4.0)
    _close(dx2[:M].float().cpu(), ref[0][0], "bn dx stream A", tol=3e-3)
    _close(dx2[M:].float().cpu(), ref[1][0], "bn dx stream B", tol=3e-3)
    _close(g.grads["bn.weight"].cpu() * 4.0, ref[0][1], "bn dgamma (A only)", tol=3e-3)
    _close(g.grads["bn.bias"].cpu() * 4.0, ref[0][2], "bn dbeta (A only)", tol=3e-3)
    assert sums.shape == (4, C)
    # parameter gradients from stream B instead
    for v in g.grads.values():
        v.zero_()
    bn.backward2(x16, dy2, sv, relu=True, param_scale=2.0, param_stream=1)
    _close(g.grads["bn.weight"].cpu() * 2.0, ref[1][1], "bn dgamma (B only)", tol=3e-3)
    _close(g.grads["bn.bias"].cpu() * 2.0, ref[1][2], "bn dbeta (B only)", tol=3e-3)


def test_batchnorm_permuted_features():
    """decoder.fc.1: reference vectors in (C,HW) order, engine rows in (HW,C) order."""
    from fmri_hip.ops import BatchNorm
    C0, HW, M = 16, 9, 6
    C = C0 * HW
    torch.manual_seed(0)
    x = _h(torch.randn(M, C))                       # reference order
    gamma = 1 + 0.2 * torch.randn(C)
    beta = 0.1 * torch.randn(C)
    g = _G({"bn.weight": gamma, "bn.bias": beta})
    g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    bn = BatchNorm(g, "bn.", C, perm=(C0, HW))
    xe = x.reshape(M, C0, HW).transpose(1, 2).reshape(M, C).contiguous()
    y16, sv = bn.forward(xe.half().to(DEV), relu=True, updates=1)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rm, rv, gr, beta, True, 0.9, 1e-5))
    got = y16.float().cpu().reshape(M, HW, C0).transpose(1, 2).reshape(M, C)
    _close(got, yr, "perm bn fwd")
    _close(g.bufs["bn.running_var"].cpu(), rv, "perm running var", tol=1e-3)
    dy = _h(torch.randn(M, C))
    yr.backward(dy)
    dye = dy.reshape(M, C0, HW).transpose(1, 2).reshape(M, C).contiguous()
    bn.backward(xe.half().to(DEV), dye.half().to(DEV), sv, True, param_scale=1.0)
    _close(g.grads["bn.weight"].cpu(), gr.grad, "perm dgamma", tol=3e-3)


def test_loss_kernels():
    from fmri_hip import lib
    P = lib.ptr
    torch.manual_seed(1)
    B, Z = 5, 128
    head = torch.randn(B, 2 * Z) * 0.5
    eps = torch.randn(B, Z)
    mu, lv = head[:, :Z], head[:, Z:]
    z_ref = eps * torch.exp(0.5 * lv) + mu
    kl_ref = (-0.5 * torch.sum(-lv.exp() - mu.pow(2) + lv + 1, 1))
    hd, ed = head.to(DEV), eps.to(DEV)
    z16 = torch.empty(B, Z, dtype=torch.float16, device=DEV)
    klr = torch.zeros(B, device=DEV)
    klt = torch.zeros(1, device=DEV)
    lib.call("fmri_latent_fwd", P(hd), P(ed), B, Z, Z, P(z16), P(klr), P(klt), 1)
    _close(z16.float().cpu(), z_ref, "z")
    _close(klr.cpu(), kl_ref, "kl rows", tol=1e-4)
    assert abs(klt.item() - kl_ref.sum().item()) < 1e-3 * abs(kl_ref.sum().item())
    # backward
    dz = torch.randn(B, Z)
    hr = head.clone().requires_grad_(True)
    zr = eps * torch.exp(0.5 * hr[:, Z:]) + hr[:, :Z]
    loss = (zr * dz).sum() + (-0.5 * torch.sum(-hr[:, Z:].exp() - hr[:, :Z].pow(2) + hr[:, Z:] + 1, 1)).sum()
    loss.backward()
    dh16 = torch.empty(B, 2 * Z, dtype=torch.float16, device=DEV)
    dh32 = torch.empty(B, 2 * Z, device=DEV)
    lib.call("fmri_latent_bwd", P(hd), P(ed), P(dz.to(DEV)), Z, 1.0, 1.0, None, B, Z, 16.0, P(dh16), P(dh32), 1)
    _close(dh32.cpu(), hr.grad, "dhead32", tol=1e-4)
    _close(dh16.float().cpu() / 16.0, hr.grad, "dhead16")
    # device-side normalisation factor n: dz carries n, the KL term is multiplied by n inside the kernel
    nrm = torch.tensor([3.0], device=DEV)
    lib.call("fmri_latent_bwd", P(hd), P(ed), P((dz * 3.0).to(DEV)), Z, 1.0, 1.0, P(nrm), B, Z, 1.0, None, P(dh32), 1)
    _close(dh32.cpu() / 3.0, hr.grad, "dhead32 normalised", tol=1e-4)
    # GAN head
    logit = torch.randn(3 * B) * 2
    p_ref = torch.sigmoid(logit)
    lr_ = logit.clone().requires_grad_(True)
    pr = torch.sigmoid(lr_)
    bo = -torch.log(pr[:B] + 1e-3)
    bp = -torch.log(1 - pr[B:2 * B] + 1e-3)
    bs = -torch.log(1 - pr[2 * B:] + 1e-3)
    (bo.sum() + bp.sum() + bs.sum()).backward()
    scal = torch.zeros(16, device=DEV)
    prob = torch.empty(3 * B, device=DEV)
    dl16 = torch.empty(3 * B, 8, dtype=torch.float16, device=DEV)
    lgd = logit.to(DEV)
    lib.call("fmri_gan_head", P(lgd), 1, B, P(prob), P(scal))
    _close(prob.cpu(), p_ref, "prob", tol=1e-5)
    _close(scal[:3].cpu(), torch.stack([bo.sum(), bp.sum(), bs.sum()]).detach(), "bce sums", tol=1e-5)
    assert abs(scal[9].item() - lr_.grad.pow(2).sum().item()) < 1e-4 * lr_.grad.pow(2).sum().item()
    nrm = torch.tensor([0.5], device=DEV)
    lib.call("fmri_gan_head_bwd", P(lgd), 1, B, P(dl16), 8, 32.0, P(nrm))
    _close(dl16[:, 0].float().cpu() / 16.0, lr_.grad, "dlogit")
    assert (dl16[:, 1:] == 0).all()
    # feature mse
    Fd = 16384
    feat = _h(torch.randn(3 * B, Fd))
    fr = feat.clone().requires_grad_(True)
    mse = torch.sum(0.5 * (fr[:B] - fr[B:2 * B]) ** 2, 1)
    mse.sum().backward()
    f16 = feat.half().to(DEV)
    rows = torch.zeros(B, device=DEV)
    tot = torch.zeros(1, device=DEV)
    df = torch.empty_like(f16)
    lib.call("fmri_feat_mse", P(f16), B, Fd, P(rows), P(tot))
    _close(rows.cpu(), mse.detach(), "mse rows", tol=1e-4)
    nrm = torch.tensor([2.0], device=DEV)
    lib.call("fmri_feat_mse_bwd", P(f16), B, Fd, P(df), 4.0, P(nrm))
    _close(df.float().cpu() / 8.0, fr.grad, "dfeat")
    # gate + stream normalisation factors
    scal = torch.tensor([0.4 * B, 0.2 * B, 0.5 * B, 7.0, 11.0, 3.0, 0, 0, 0, 12.0] + [0.0] * 6, device=DEV)
    flags = torch.zeros(2, dtype=torch.int32, device=DEV)
    lib.call("fmri_compose_gate", P(scal), P(flags), float(B), 100.0, 1e-6, 0.68, 0.35, 1, -1, -1)
    s = scal.cpu()
    assert flags.tolist() == [0, 1]                     # bce_pred mean 0.2 < 0.33 -> discriminator paused
    assert abs(s[6] - 18.0) < 1e-5 and abs(s[7] - 1.1 * B) < 1e-5
    assert abs(s[8] - (1e-6 * 11.0 - (1 - 1e-6) * 1.1 * B)) < 1e-5
    na, nb = 1 / np.sqrt(12.0 / (3 * B)), 1 / np.sqrt(2 * 11.0 / (B * 100.0))
    assert abs(s[10] - na) < 1e-4 * na and abs(s[11] - nb) < 1e-4 * nb and abs(s[12] - na / nb) < 1e-4 * na / nb


def test_optimizers_match_torch():
    from fmri_hip import lib
    P = lib.ptr
    torch.manual_seed(3)
    n = 10007
    p0 = torch.randn(n)
    grads = [torch.randn(n) * (10.0 ** np.random.RandomState(i).uniform(-6, 0)) for i in range(3)]
    # RMSprop
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.RMSprop([pt], lr=1e-4, alpha=0.9, eps=1e-8)
    pd, sq = p0.to(DEV).clone(), torch.zeros(n, device=DEV)
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    for g in grads:
        pt.grad = g.clone()
        opt.step()
        # gradient stored with a device-side factor 4 that the kernel divides out again
        four = torch.tensor([4.0], device=DEV)
        lib.call("fmri_rmsprop", P(pd), P((g * 4.0).to(DEV)), P(sq), n, 1e-4, 0.9, 1e-8, 1.0, P(four), 0.0, P(flag))
    assert torch.allclose(pd.cpu(), pt.detach(), rtol=1e-6, atol=1e-7)
    flag.zero_()
    before = pd.clone()
    lib.call("fmri_rmsprop", P(pd), P(grads[0].to(DEV)), P(sq), n, 1e-4, 0.9, 1e-8, 1.0, None, 0.0, P(flag))
    assert torch.equal(pd, before)                       # gated off
    # Adam(0.5, 0.999)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4, betas=(0.5, 0.999))
    pd, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for t, g in enumerate(grads, 1):
        pt.grad = g.clone()
        opt.step()
        lib.call("fmri_adam", P(pd), P(g.to(DEV)), P(m), P(v), n, 1e-4, 0.5, 0.999, 1e-8, 1 - 0.5 ** t,
                 float(np.sqrt(1 - 0.999 ** t)), 1.0, None, 0.0, None)
    assert torch.allclose(pd.cpu(), pt.detach(), rtol=2e-6, atol=1e-7)


ROUTE_VARIANTS = {
    # the round-2 kernels behind the 8-wave loader / compute forms (they still own the 64-channel tiles, the
    # BatchNorm-backward epilogue and the statistics groups the wide forms decline)
    "narrower": dict(FMRI_C5W="off", FMRI_TC5W="off"),
    # every specialised kernel off: the tap-list kernel (csrc/igemm.hip) takes all geometries
    "generic": dict(FMRI_C5="off", FMRI_TC5="off", FMRI_TC32="off", FMRI_NARROW="off"),
}


@pytest.mark.parametrize("route", sorted(ROUTE_VARIANTS))
def test_igemm_routing_variants(route):
    """Every geometry has a second kernel behind the one the default routing picks (fmri_igemm_ep falls through to it when
    the first declines): the conv / deconv parity cases are re-run with the first choice switched off.  The switches are
    read once per process, hence the child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **ROUTE_VARIANTS[route])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "conv_forward or deconv_forward or epilogues", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] sizes (B = 256: discriminator batch 3B = 768, decoder batch 2B = 512).  The CPU reference
# cannot run these in seconds, so the full-size launches are pinned through size-independent properties:
#   * per-image independence of a convolution: three sampled images of the full-batch result equal the CPU fp32
#     convolution of those three images alone (same tolerance as the small cases);
#   * adjointness, with y = layer(x):  <y, y> = <dgrad(y), x> = <wgrad(x, y), w>  -- one scalar that ties the forward,
#     data-gradient and weight-gradient kernels of a layer together at any size (fp64 dot products on the device;
#     the cotangent is y itself so that the product is a sum of squares and rounding noise averages out).
FULL_SIZE = [
    # kind, cin, cout, stride, H, out_pad, N
    ("conv", 3, 32, 1, 64, 0, 768),       # discriminator.conv.0
    ("conv", 32, 128, 2, 64, 0, 768),     # discriminator.conv.1  (dgrad: igemm_tc32)
    ("conv", 128, 256, 2, 32, 0, 768),    # discriminator.conv.2  (256x256 tile; dgrad: window-resident kernel)
    ("conv", 256, 256, 2, 16, 0, 768),    # discriminator.conv.3
    ("conv", 3, 64, 2, 64, 0, 256),       # encoder.conv.0
    ("conv", 64, 128, 2, 32, 0, 256),     # encoder.conv.1
    ("conv", 128, 256, 2, 16, 0, 256),    # encoder.conv.2
    ("deconv", 256, 256, 2, 8, 1, 512),   # decoder.conv.0
    ("deconv", 256, 128, 2, 16, 1, 512),  # decoder.conv.1
    ("deconv", 128, 32, 2, 32, 1, 512),   # decoder.conv.2        (igemm_tc32)
    ("conv", 32, 3, 1, 64, 0, 512),       # decoder.conv.3        (igemm_narrow)
    # the 128-px geometry of BASELINE configs[4] (ArchConfig.px128: fc_input 16, stride_gan 2) at its per-GPU batch 128
    # (1024 over 8 GPUs); the discriminator behind conv.0 has the 64-px shapes above
    ("conv", 3, 64, 2, 128, 0, 128),      # encoder.conv.0 @128 px
    ("conv", 64, 128, 2, 64, 0, 128),     # encoder.conv.1 @128 px
    ("conv", 128, 256, 2, 32, 0, 128),    # encoder.conv.2 @128 px
    ("deconv", 256, 256, 2, 16, 1, 128),  # decoder.conv.0 @128 px
    ("deconv", 256, 128, 2, 32, 1, 128),  # decoder.conv.1 @128 px
    ("deconv", 128, 32, 2, 64, 1, 128),   # decoder.conv.2 @128 px
    ("conv", 32, 3, 1, 128, 0, 128),      # decoder.conv.3 @128 px
    ("conv", 3, 32, 2, 128, 0, 384),      # discriminator.conv.0 @128 px (stride_gan = 2), 3 x 128 images
]


@pytest.mark.parametrize("kind,cin,cout,stride,H,op,N", FULL_SIZE)
def test_full_size_layers_sampled_images_and_adjoint_identities(kind, cin, cout, stride, H, op, N):
    from fmri_hip.ops import ConvLayer, ACT_NONE, images_to_nhwc, nhwc_to_images
    torch.manual_seed(cin * 7 + cout + H + N)
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    w = _h(torch.randn(shape) * (1.0 / (5.0 * cin ** 0.5)))
    g = _G({"w": w})
    layer = ConvLayer(g, "w", None, kind, cin, cout, 5, stride, 2, op)
    x = torch.randn(N, cin, H, H, device=DEV).half().float()
    x16 = images_to_nhwc(x)
    y16 = layer.forward(x16, ACT_NONE)
    Ho = y16.shape[1]
    # (1) sampled images against the CPU convolution
    pick = [0, N // 2 + 1, N - 1]
    xs = x[pick].cpu()
    if kind == "conv":
        ref = F.conv2d(xs, w, None, stride, 2)
    else:
        ref = F.conv_transpose2d(xs, w, None, 2, 2, output_padding=op)
    assert ref.shape[2] == Ho
    _close(nhwc_to_images(y16[pick].contiguous(), cout).cpu(), ref, f"{kind} fwd, sampled images")
    # (2) adjoint identities with the cotangent y
    yy = y16[..., :cout].double().pow(2).sum().item()
    dx16 = layer.dgrad(y16, H, H)
    lhs_d = (dx16[..., :cin].double() * x16[..., :cin].double()).sum().item()
    layer.wgrad(x16, y16, 1.0)
    _join()
    lhs_w = (g.grads["w"].double() * g.views["w"].double()).sum().item()
    assert yy > 0
    # dgrad result is stored in fp16 (relative 2^-11 per element, random sign); the weight gradient is fp32
    assert abs(lhs_d - yy) < 1e-3 * yy, (lhs_d, yy)
    assert abs(lhs_w - yy) < 1e-3 * yy, (lhs_w, yy)
    # (3) sampled images of the data gradient against the CPU (per-image independence holds for dgrad too)
    ys = nhwc_to_images(y16[pick].contiguous(), cout).cpu()
    xr = xs.clone().requires_grad_(True)
    if kind == "conv":
        F.conv2d(xr, w, None, stride, 2).backward(ys)
    else:
        F.conv_transpose2d(xr, w, None, 2, 2, output_padding=op).backward(ys)
    _close(nhwc_to_images(dx16[pick].contiguous(), cin).cpu(), xr.grad, f"{kind} dgrad, sampled images")


AFFINE_CASES = [
    # kind, cin, cout, H, out_pad, N, expect the epilogue
    ("conv", 128, 256, 32, 0, 3, True),       # igemm_c5w<16>
    ("conv", 256, 256, 16, 0, 5, True),       # igemm_c5w<8>, partial last tile
    ("conv", 32, 128, 20, 0, 2, True),        # one sub-chunk, partial tiles
    ("deconv", 256, 128, 16, 1, 2, True),     # igemm_tc5w<16>
    ("deconv", 256, 256, 8, 1, 6, True),      # igemm_tc5w<8>
    ("deconv", 128, 32, 16, 1, 2, False),     # igemm_tc32: no such epilogue -> plain output, bit clear
]


@pytest.mark.parametrize("kind,cin,cout,H,op,N,expect", AFFINE_CASES)
def test_conv_epilogue_eval_batchnorm(kind, cin, cout, H, op, N, expect):
    """fmri_epilogue.aff_*: an eval-mode BatchNorm (+ ReLU) folded into the convolution in front of it equals the
    convolution followed by BatchNorm.forward_eval (one fp16 rounding less: compared at 2e-3); kernels without the
    epilogue leave the plain output and say so."""
    from fmri_hip.ops import ConvLayer, BatchNorm
    torch.manual_seed(cin + 3 * cout + H + N)
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = _G({"w": _h(torch.randn(*shape) * 0.05), "bn.weight": torch.rand(cout) + 0.5, "bn.bias": torch.randn(cout) * 0.2})
    g.bufs = {"bn.running_mean": torch.randn(cout, device=DEV) * 0.3, "bn.running_var": torch.rand(cout, device=DEV) + 0.5,
              "bn.num_batches_tracked": torch.ones((), dtype=torch.int64, device=DEV)}
    layer = ConvLayer(g, "w", None, kind, cin, cout, 5, 2, 2, op)
    bn = BatchNorm(g, "bn.", layer.coutp)
    x16 = torch.randn(N, H, H, layer.cinp, device=DEV).half()
    raw = layer.forward(x16)
    ref = bn.forward_eval(raw, relu=True)
    fused = layer.forward(x16, affine=bn.eval_affine())
    assert layer.aff_applied == expect
    if not expect:
        assert torch.equal(fused, raw)
        return
    err = (fused.float() - ref.float()).abs()
    lim = 2e-3 * ref.float().abs() + 2e-3 * float(ref.float().pow(2).mean().sqrt())
    assert (err <= lim).all(), float(err.max())
    assert (fused >= 0).all()


def test_output_stores_survive_memory_contention():
    """gfx950 store-data hazard, timing-dependent form (csrc/common.h FMRI_STORE_FENCE): the loader
    waves' `buffer_store_dwordx4 ..., sN offen` followed by a VALU write of the first data register stored a wrong first
    dword in 268 of 400 launches of igemm_tc5w<8,0,solo> -- but only while a kernel on another stream kept the memory
    system busy, so no single-stream parity test could see it.  Here the data gradient of discriminator.conv.3 at the
    size that showed it (and the statistics forward of decoder.conv.0, the other solo instantiation) runs 120 times
    beside a weight-gradient kernel on a second stream; every output must equal the idle-GPU result bit for bit."""
    from fmri_hip.ops import ConvLayer
    from fmri_hip import ops
    torch.manual_seed(3)
    side = torch.cuda.Stream()
    wl = ConvLayer(_G({"w": _h(torch.randn(256, 256, 5, 5) * 0.05)}), "w", None, "conv", 256, 256, 5, 2, 2)
    wx = torch.randn(96, 16, 16, 256, device=DEV).half()
    wdy = torch.randn(96, 8, 8, 256, device=DEV).half()
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    try:
        for kind, cin, cout, H, N, what in (("conv", 256, 256, 16, 40, "dgrad"), ("deconv", 256, 256, 8, 24, "fwd")):
            shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
            L = ConvLayer(_G({"w": _h(torch.randn(*shape) * 0.05)}), "w", None, kind, cin, cout, 5, 2, 2,
                          1 if kind == "deconv" else 0)
            Ho, Wo = L.out_hw(H, H)
            if what == "dgrad":
                inp = (torch.randn(N, Ho, Wo, L.coutp, device=DEV) * 0.5).half()
                run = lambda out: L.dgrad(inp, H, H, out=out)
                ref = torch.empty(N, H, H, L.cinp, dtype=torch.float16, device=DEV)
            else:
                inp = torch.randn(N, H, H, L.cinp, device=DEV).half()
                run = lambda out: L.forward(inp, out=out, bn_groups=1)
                ref = torch.empty(N, Ho, Wo, L.coutp, dtype=torch.float16, device=DEV)
            run(ref)
            torch.cuda.synchronize()
            ring = [torch.empty_like(ref) for _ in range(4)]
            bad = 0
            for it in range(120):
                with torch.cuda.stream(side):
                    wl._wgrad(wx, wdy, 1.0)
                run(ring[it % 4])
                if it % 4 == 3:
                    torch.cuda.synchronize()
                    bad += sum(int(not torch.equal(o, ref)) for o in ring)
            assert bad == 0, (kind, what, bad)
    finally:
        ops._SIDE["on"] = side_was
