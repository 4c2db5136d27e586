"""The engine's FLOP counter must reproduce the per-image constants the measurement contract is priced in
(E = 253.6, D = 862.7, S = 875.3 MFLOP forward at 64 px; Stage-I step 13.81 GFLOP / image;
128 px: E = 1012.9, D = 3450.9, S = 875.3, step 31.62 GFLOP / image)."""
import pytest

from fmri_hip.params import ArchConfig, forward_flops, stage1_step_flops, stage2_step_flops


def test_px64_constants():
    f = forward_flops(ArchConfig.px64())
    assert round(f["E"] / 1e6, 1) == 253.6
    assert round(f["D"] / 1e6, 1) == 862.7
    assert round(f["S"] / 1e6, 1) == 875.3
    assert round(f["C"] / 1e6, 1) == 8.9
    assert round(f["W"] / 1e6, 1) == 1.7
    assert round(stage1_step_flops(ArchConfig.px64()) / 1e9, 2) == 13.81


def test_px128_constants():
    f = forward_flops(ArchConfig.px128())
    assert round(f["E"] / 1e6, 1) == 1012.9
    assert round(f["D"] / 1e6, 1) == 3450.9
    assert round(f["S"] / 1e6, 1) == 875.3
    assert round(stage1_step_flops(ArchConfig.px128()) / 1e9, 2) == 31.62


def test_px100_as_shipped():
    f = forward_flops(ArchConfig.px100())
    assert round(f["E"] / 1e6, 1) == 647.6
    assert round(f["D"] / 1e6, 1) == 2742.1


def test_bench_constant_matches_counter():
    import bench
    assert bench.FLOP_PER_IMAGE == pytest.approx(stage1_step_flops(ArchConfig.px64()), rel=1e-3)


def test_stage2_constant():
    assert round(stage2_step_flops(ArchConfig.px64(), 4096) / 1e9, 2) == 11.61


def test_kernel_routing_of_the_headline_layers():
    """fmri_igemm_route -- the library's own statement of csrc/api.hip's routing, host code that needs no GPU (bench.py
    prices kernel families by it): the stride-2 layers of the B = 256 Stage-I step go to the 8-wave loader / compute kernels
    of round 3, the 32-channel and 64-channel-tile ones stay where they were, and a request for the BatchNorm-backward
    epilogue is never routed to a kernel without one."""
    from fmri_hip import lib
    from fmri_hip.ops import igemm_route as route, MODE_CONV, MODE_TCONV2, ACT_NONE, ACT_RELU

    def tconv_elems(ci, rows_pad):
        return max(g["w_off"] + rows_pad * g["kpad"] for g in (lib.tconv_class(5, 2, cy, cx, ci, rows_pad)
                                                                for cy in range(2) for cx in range(2)))
    conv = lambda N, H, ci, co, **kw: route(N, H, H, ci, H // 2, H // 2, co, co, 5, 2, 2, MODE_CONV, ACT_NONE, False, 1, 128,
                                            co * lib.kpad(25, ci), **kw)
    tconv = lambda N, H, ci, co, tile=128, **kw: route(N, H, H, ci, 2 * H, 2 * H, co, co, 5, 2, 2, MODE_TCONV2, ACT_NONE,
                                                       False, 1, tile, tconv_elems(ci, max(co, tile)), **kw)
    assert conv(768, 32, 128, 256) == "fmri::igemm_c5w_kernel<16,0>"     # discriminator.conv.2 forward
    assert conv(768, 32, 128, 256, stat_rows_cap=4096) == "fmri::igemm_c5w_kernel<16,1>"     # ... with BatchNorm statistics
    assert conv(768, 64, 32, 128).startswith("fmri::igemm_c5w_kernel<16")   # discriminator.conv.1 forward (one sub-chunk)
    assert conv(768, 16, 256, 256) == "fmri::igemm_c5w_kernel<8,0>"      # discriminator.conv.3 forward (8 x 8 outputs)
    assert tconv(1536, 16, 256, 128) == "fmri::igemm_tc5w_kernel<16,0,false>"    # discriminator.conv.2 data gradient
    assert tconv(1536, 8, 256, 256) == "fmri::igemm_tc5w_kernel<8,0,false>"      # discriminator.conv.3 data gradient
    assert tconv(256, 8, 256, 128) == "fmri::igemm_tc5w_kernel<8,0,true>"        # encoder.conv.2 data gradient: one class per block
    assert tconv(512, 16, 256, 128, stat_rows_cap=4096) == "fmri::igemm_tc5w_kernel<16,1,false>"   # decoder.conv.1 forward
    assert tconv(256, 16, 128, 64, 64).startswith("fmri::igemm_tc5_kernel<64")   # encoder.conv.1 data gradient
    assert tconv(512, 32, 128, 32, 32) == "fmri::igemm_tc32_kernel<false>"   # decoder.conv.2 forward
    assert tconv(1536, 32, 128, 32, 32, want_act_y=True) == "fmri::igemm_tc32_kernel<true>"   # discriminator.conv.1 dgrad + ReLU mask
    # the BatchNorm-backward epilogue exists in the narrower kernels only (advisor finding of round 3: the request must
    # decide, not the row capacity)
    assert conv(512, 32, 128, 256, stat_rows_cap=4096, want_bn_bwd=True).startswith("fmri::igemm_c5_kernel<16,2")
    assert conv(512, 32, 128, 256, stat_rows_cap=1, want_bn_bwd=True).startswith("fmri::igemm_c5_kernel<16,0")
    assert tconv(512, 16, 256, 128, stat_rows_cap=4096, want_bn_bwd=True).startswith("fmri::igemm_tc5_kernel<128")
    # small-channel stride-1 layers and the dense layers
    n = lambda N, ci, co, mode=MODE_CONV, act=ACT_NONE, bias=False: route(N, 64, 64, ci, 64, 64, co, min(co, 3) if co == 8 else co,
                                                                          5, 1, 2, mode, act, False, 1, 32, 32 * lib.kpad(25, ci),
                                                                          has_bias=bias)
    assert n(768, 8, 32, act=ACT_RELU, bias=True) == "fmri::igemm_narrow_kernel<8,2,false>"      # discriminator.conv.0
    assert n(512, 32, 8).startswith("fmri::igemm_narrow_kernel<32,1")                             # decoder.conv.3
    dense = route(256, 1, 1, 16384, 1, 1, 1024, 1024, 1, 1, 0, MODE_CONV, ACT_NONE, True, 8, 64, 1024 * 16384)
    assert dense == "fmri::igemm_kernel<128,64,2,2,true,true>", dense
