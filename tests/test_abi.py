"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol include/fmri_hip.h
declares, and its host-side index arithmetic is exact.  No kernels are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    from fmri_hip import build, lib as L
    build.build(verbose=False)
    return L.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "fmri_hip.h")).read()
    declared = set(re.findall(r"\b(fmri_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fmri_err", "fmri_act", "fmri_mode"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/fmri_hip.h but not exported"
    from fmri_hip import lib as L
    assert set(L.EXPORTS) == declared


def test_version_and_error_strings(lib):
    assert lib.fmri_version() >= 100
    assert lib.fmri_last_error_string(0) == b"ok"
    assert b"bad argument" in lib.fmri_last_error_string(-1)


def test_fastdiv_is_exact(lib):
    rs = np.random.RandomState(0)
    divisors = [1, 2, 3, 5, 7, 8, 13, 25, 50, 64, 100, 169, 625, 4096, 10000, 65537, 2 ** 31 - 1]
    for d in divisors:
        ns = np.concatenate([np.arange(0, 2000), rs.randint(0, 2 ** 32, 2000, dtype=np.uint64),
                             np.array([2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1, d, d - 1, 2 * d, 2 * d - 1])])
        for n in ns:
            n = int(n) & 0xFFFFFFFF
            assert lib.fmri_test_fastdiv(n, d) == n // d, (n, d)


def test_tconv_class_geometry(lib):
    """k=5, pad=2 stride-2 transposed conv: classes have 3x3, 3x2, 2x3, 2x2 taps (25 in total)."""
    from fmri_hip import lib as L
    taps = 0
    off = 0
    for cy in range(2):
        for cx in range(2):
            g = L.tconv_class(5, 2, cy, cx, 128, 256)
            assert (g["py"], g["px"]) == (cy, cx)
            assert g["th"] == (3 if cy == 0 else 2) and g["tw"] == (3 if cx == 0 else 2)
            assert g["kpad"] == g["th"] * g["tw"] * 128 and g["kpad"] % 64 == 0
            assert g["w_off"] == off
            off += 256 * g["kpad"]
            taps += g["th"] * g["tw"]
    assert taps == 25
    assert L.kpad(25, 8) == 256 and L.kpad(1, 4096) == 4096


def test_argument_validation_without_gpu(lib):
    """Bad shapes are rejected on the host before anything is enqueued."""
    from fmri_hip import lib as L
    z = ctypes.c_void_p(16)
    # Ci not a multiple of 8
    assert lib.fmri_igemm(z, z, z, None, z, 1, 4, 4, 3, 4, 4, 8, 8, 5, 1, 2, 0, 0, 0, 1, 0, 32, None) == -1
    # unsupported tile
    assert lib.fmri_igemm(z, z, z, None, z, 1, 4, 4, 8, 4, 4, 8, 8, 5, 1, 2, 0, 0, 0, 1, 0, 48, None) == -2
    # split-K needs fp32 slabs
    assert lib.fmri_igemm(z, z, z, None, z, 1, 4, 4, 8, 4, 4, 8, 8, 5, 1, 2, 0, 0, 0, 2, 0, 32, None) == -1
    assert L.load().fmri_bn_stats(None, 4, 8, None, None, 0, None) == -1
    assert L.load().fmri_bn_ws_floats(786432, 128) >= 2 * 128


def test_latent_range_scale_and_its_argument_checks(lib):
    """Host side of the range-safe latent path (fmri_latent_fwd_ranged): the scale a batch maximum maps to is
    the LARGEST power of two <= 1 that brings it under the cap -- 1 inside the cap (a healthy batch is stored unchanged)
    and for a non-finite maximum (inf / NaN rows are carried on, not hidden) -- and bad arguments are refused before
    anything is enqueued."""
    import math
    cap = 256.0
    f = lambda m: lib.fmri_latent_range_scale(ctypes.c_float(m), ctypes.c_float(cap))
    for m in (0.0, 1e-30, 1.0, 255.99, 256.0):
        assert f(m) == 1.0, m
    for m in (256.0001, 300.0, 511.9, 512.0, 512.1, 65504.0, 9.4e4, 1e9, 2e20):
        s = f(m)
        assert s < 1.0 and math.log2(s) == int(math.log2(s)), (m, s)
        assert np.float32(m) * np.float32(s) <= cap < np.float32(m) * np.float32(s) * 2, (m, s)
    assert f(1e38) == 2.0 ** -60                       # capped: eps * s^2 must stay a normal fp32 number
    assert f(float("inf")) == 1.0 and f(float("nan")) == 1.0
    z = ctypes.c_void_p(16)
    L = lib
    # phase out of range, missing scratch / maximum, zp < Z, non-positive cap in phase 2
    assert L.fmri_latent_fwd_ranged(z, z, 4, 128, 128, z, None, None, 1, z, z, z, cap, 0, None) == -1
    assert L.fmri_latent_fwd_ranged(z, z, 4, 128, 128, z, None, None, 1, None, z, z, cap, 3, None) == -1
    assert L.fmri_latent_fwd_ranged(z, z, 4, 128, 128, z, None, None, 1, z, None, z, cap, 3, None) == -1
    assert L.fmri_latent_fwd_ranged(z, z, 4, 128, 120, z, None, None, 1, z, z, z, cap, 3, None) == -1
    assert L.fmri_latent_fwd_ranged(z, z, 4, 128, 128, z, None, None, 1, z, z, z, 0.0, 2, None) == -1
    assert L.fmri_latent_fwd_ranged(None, z, 4, 128, 128, z, None, None, 1, z, z, z, cap, 1, None) == -1     # phase 1 needs the heads
    assert L.fmri_rows_absmax(None, 16, z, None) == -1
    assert L.fmri_sumsq_f64(z, 16, ctypes.c_void_p(12), 1, None) == -1                                      # misaligned double


def _scanner():
    import importlib.util
    spec = importlib.util.spec_from_file_location("scan_store_hazard", os.path.join(ROOT, "tools", "scan_store_hazard.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    return scan


def test_apply_table_rows_without_gpu(lib):
    """fmri_apply_entry_fill (host code): which layout maps the one-launch update takes, how many blocks a row occupies,
    and what it refuses -- for the parameter shapes of the model (models/vae_gan.py)."""
    n = lib.fmri_apply_entry_bytes()
    assert 64 <= n <= 256
    buf = ctypes.create_string_buffer(n)
    P = lambda v: ctypes.c_void_p(v)
    w, sq, g, src, pk = 0x10000, 0x20000, 0x30000, 0x40000, 0x50000

    def fill(sa, sta, sb, stb, A, TA, B, KW, TH, TW, ld, kpad, flat=0, py=0, px=0, step=1, nsl=1, gsrc=src, pkp=pk):
        return lib.fmri_apply_entry_fill(buf, P(gsrc), P(w), P(sq), P(g), P(pkp), sa, sta, sb, stb, A, TA, B, KW, py, px,
                                         step, TH, TW, ld, kpad, nsl, A * TA * ld, 0, 1.0, flat, 0)
    # Conv2d(128 -> 256, k5): rows co, taps contiguous in the reference layout: one block per (row, 64 input channels)
    assert fill(128 * 25, 0, 25, 1, 256, 1, 128, 5, 5, 5, 25 * 128, 25 * 128) == 256 * 2
    # Conv2d(3 -> 32): fewer than 64 channels -> 32-wide tiles
    assert fill(3 * 25, 0, 25, 1, 32, 1, 3, 5, 5, 5, 256, 256) == 32
    # Linear(16384 -> 1024) behind the (C,H,W) flatten: 64 'taps' (positions) of 256 channels
    assert fill(16384, 0, 64, 1, 1024, 1, 256, 64, 1, 64, 16384, 16384) == 1024 * 4
    # plain Linear(1024 -> 256): row-contiguous, 1024-element chunks
    assert fill(1024, 0, 1, 0, 256, 1, 1024, 1, 1, 1, 1024, 1024) == 256
    # a flat segment of 1-D parameters
    assert fill(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, flat=5000, gsrc=0, pkp=0) == 5
    # a tap SUBSET (one parity class of a transposed convolution) is not a gradient layout: refused (0 = keep the
    # separate launches), as is a transposed dense map
    assert fill(25, 0, 128 * 25, 1, 128, 1, 256, 5, 2, 2, 4 * 256, 4 * 256, py=1, px=1, step=2) == 0
    assert fill(1, 0, 1024, 0, 1024, 1, 256, 1, 1, 1, 256, 256) == 0
    # bad arguments: leading dimension too small for the taps, missing gradient source, fp16 copy narrower than the taps
    assert fill(128 * 25, 0, 25, 1, 256, 1, 128, 5, 5, 5, 100, 25 * 128) < 0
    assert fill(128 * 25, 0, 25, 1, 256, 1, 128, 5, 5, 5, 25 * 128, 25 * 128, gsrc=0) < 0
    assert fill(128 * 25, 0, 25, 1, 256, 1, 128, 5, 5, 5, 25 * 128, 64) < 0
    # launches are refused without a table / with an unknown mode (no GPU touched)
    assert lib.fmri_apply_batch(None, 1, 1, 1, None, 0.9, 1e-8, 1.0, None, 0.0, None, 0, None) < 0
    assert lib.fmri_apply_batch(P(0x1000), 1, 1, 7, None, 0.9, 1e-8, 1.0, None, 0.0, None, 0, None) < 0
    assert lib.fmri_apply_batch(P(0x1000), 1, 1, 1, None, 0.9, 1e-8, 1.0, None, 0.0, None, 0, None) < 0      # mode 1 needs lr
    assert lib.fmri_transpose_f16(P(0x1000), P(0x2000), 8, 8, 8, 12, 8, None) < 0                             # ld not a multiple of 8


def test_no_store_data_hazard_in_the_code_objects(lib):
    """gfx950: a VMEM store of more than 64 bits followed within two issue slots by a VALU write of one of its data
    registers can store the NEW value (measured twice: DESIGN section 6; csrc/common.h FMRI_STORE_FENCE).  The compiler pads
    no wait states behind stores with an SGPR offset, so the built code objects are scanned: every wide store must have two
    wait states in front of such a write.  (Also a build step of fmri_hip/build.py.)"""
    scan = _scanner()
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    from fmri_hip import build
    hits, nstores, nkern, nobj = scan.scan(build.LIB)
    assert nobj >= 10 and nstores > 500, (nobj, nstores)       # the scan saw the kernels
    assert not hits, hits


def test_store_hazard_scanner_recognises_the_measured_patterns():
    """The instruction pairs of the two broken builds are hits -- round 3's packed-fp32 writers directly behind the store
    (profiles/r03_store_hazard.txt) and round 4's single-register `v_cndmask_b32` into the FIRST data register
    (profiles/r04_store_hazard.txt) -- and so is a writer with only ONE wait state in between; the harmless neighbours
    (a writer of OTHER registers, `s_nop 1` or two instructions in between, a 64-bit store, a VMEM load into the data
    registers) are not."""
    scan = _scanner()
    dis = """
0000000000001000 <_ZN4fmri6brokenEv>:
\tbuffer_store_dwordx4 v[14:17], v95, s[24:27], s20 offen  // 000000001000: E07C1000 14065F0E
\tv_pk_add_f32 v[14:15], v[56:57], v[52:53]                // 000000001008: D3B2400E 1802693
\tbuffer_store_dwordx4 v[6:9], v97, s[24:27], s22 offen
\tv_pk_mul_f32 v[6:7], v[70:71], v[70:71]
\tbuffer_store_dwordx4 v[24:27], v28, s[20:23], s15 offen
\tv_cndmask_b32_e64 v24, v64, v80, s[2:3]
\tbuffer_store_dwordx4 v[50:53], v54, s[52:55], s93 offen
\ts_cmp_gt_i32 s12, 5
\tv_cvt_f32_f16_sdwa v50, v46 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1
\tglobal_store_dwordx4 v[92:93], v[34:37], off
\ts_nop 0
\tv_lshl_add_u64 v[36:37], s[28:29], 0, v[82:83]
0000000000002000 <_ZN4fmri4fineEv>:
\tbuffer_store_dwordx4 v[26:29], v90, s[24:27], s28 offen
\tv_pk_add_f32 v[32:33], v[40:41], v[32:33]
\tbuffer_store_dwordx4 v[10:13], v96, s[24:27], s21 offen
\ts_nop 1
\tv_pk_add_f32 v[10:11], v[64:65], v[52:53]
\tbuffer_store_dwordx4 v[44:47], v52, s[64:67], s70 offen
\ts_cselect_b64 s[8:9], -1, 0
\ts_mul_i32 s19, s14, 20
\tv_cndmask_b32_e32 v44, v66, v65, vcc
\tbuffer_store_dwordx2 v[2:3], v96, s[24:27], s21 offen
\tv_pk_add_f32 v[2:3], v[64:65], v[52:53]
\tglobal_store_dwordx4 v[82:83], v[34:37], off
\tglobal_load_dwordx4 v[34:37], v[84:85], off
"""
    hits, nstores, nkern = scan.scan_text(dis)
    assert nstores == 9 and nkern == 2
    assert [h[0] for h in hits] == ["_ZN4fmri6brokenEv"] * 5, hits
    assert [h[3] for h in hits] == [0, 0, 0, 1, 1], hits             # wait states seen in front of the writer
    assert "v[14:15]" in hits[0][2] and "v[6:7]" in hits[1][2] and "v_cndmask_b32_e64 v24" in hits[2][2]
    assert "v_cvt_f32_f16_sdwa v50" in hits[3][2] and "v_lshl_add_u64" in hits[4][2]
    # a stricter requirement also reports the pairs with two wait states in between
    assert len(scan.scan_text(dis, need=3)[0]) == 7
