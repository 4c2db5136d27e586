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
    """ops.igemm_kernel_label mirrors csrc/api.hip's routing (bench.py prices kernel families by it): the stride-2 layers of
    the B = 256 Stage-I step go to the 8-wave loader / compute kernels of round 3, the 32-channel and 64-channel-tile ones
    stay where they were."""
    from fmri_hip.ops import igemm_kernel_label as lab, MODE_CONV, MODE_TCONV2
    conv = lambda N, H, ci, co: lab(N, H, H, ci, H // 2, H // 2, co, co, 5, 2, 2, MODE_CONV, False, 1, 128)
    tconv = lambda N, H, ci, co, tile=128: lab(N, H, H, ci, 2 * H, 2 * H, co, co, 5, 2, 2, MODE_TCONV2, False, 1, tile)
    assert conv(768, 32, 128, 256) == "fmri::igemm_c5w_kernel"           # discriminator.conv.2 forward
    assert conv(768, 64, 32, 128) == "fmri::igemm_c5w_kernel"            # discriminator.conv.1 forward (one sub-chunk)
    assert conv(768, 16, 256, 256) == "fmri::igemm_c5w_kernel"           # discriminator.conv.3 forward (8 x 8 outputs)
    assert tconv(1536, 16, 256, 128) == "fmri::igemm_tc5w_kernel"        # discriminator.conv.2 data gradient
    assert tconv(1536, 8, 256, 256) == "fmri::igemm_tc5w_kernel"         # discriminator.conv.3 data gradient (8 x 8 grid)
    assert tconv(512, 16, 256, 128) == "fmri::igemm_tc5w_kernel"         # decoder.conv.1 forward
    assert tconv(256, 16, 128, 64, 64).startswith("fmri::igemm_tc5_kernel<64")   # encoder.conv.1 data gradient
    assert tconv(512, 32, 128, 32) == "fmri::igemm_tc32_kernel"          # decoder.conv.2 forward
