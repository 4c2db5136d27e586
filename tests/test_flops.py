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
