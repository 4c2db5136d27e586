"""ctypes binding of libfmri_hip.so (C ABI: include/fmri_hip.h).

There is no fallback: if the shared library is missing or a kernel returns an error the caller gets a
RuntimeError.  PyTorch is used only as the allocator / stream provider (tensor.data_ptr(),
torch.cuda.current_stream()).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FMRI_LIB_PATH: another build of the same library (A/B timing of two builds in one gpurun call)
LIB_PATH = os.environ.get("FMRI_LIB_PATH") or os.path.join(_HERE, "libfmri_hip.so")

_i, _f, _l, _p = C.c_int, C.c_float, C.c_int64, C.c_void_p

# name -> argtypes (restype is int unless noted)
_SIGS = {
    "fmri_tconv_class": [_i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                         C.POINTER(_i), C.POINTER(_l)],
    "fmri_kpad": [_i, _i],
    "fmri_pack_weight": [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fmri_pack_entry_bytes": [],
    "fmri_apply_entry_bytes": [],
    "fmri_transpose_entry_bytes": [],
    "fmri_transpose_entry_fill": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i],
    "fmri_transpose_f16_batch": [_p, _i, _i, _p],
    "fmri_pack_entry_fill": [_p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "fmri_pack_weight_batch": [_p, _i, _i, _p],
    "fmri_apply_entry_fill": [_p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l,
                              _i, _f, _l, _i],
    "fmri_transpose_f16": [_p, _p, _i, _i, _i, _i, _i, _p],
    "fmri_apply_batch": [_p, _i, _i, _i, _p, _f, _f, _f, _p, _f, _p, _i, _p],
    "fmri_ingest_u8": [_p, _i, _i, _i, _i, _p, _p, _f, _f, _f, _f, _f, _f, _p, _p, _p],
    "fmri_crop_resize_u8": [_p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _i, _p, _p],
    "fmri_pcc": [_p, _p, _l, _p, _p, _p],
    "fmri_ssim": [_p, _p, _i, _i, _i, _p, _p, _p, _p],
    "fmri_unpack_grad": [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _l, _p],
    "fmri_igemm": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _p],
    "fmri_igemm_ep": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _l, _p, _p,
                      _p],
    "fmri_igemm_route": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _i,
                         C.c_char_p, _i],
    "fmri_wgrad_slabs": [_i, _i, _i, _i, _i, _i],
    "fmri_wgrad_narrow_blocks": [_i, _i, _i],
    "fmri_set_deterministic": [_i],
    "fmri_get_deterministic": [],
    "fmri_wgrad": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fmri_wgrad_if": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fmri_nchw_to_nhwc": [_p, _p, _i, _i, _i, _i, _p],
    "fmri_nhwc_to_nchw": [_p, _p, _i, _i, _i, _i, _f, _p],
    "fmri_rows_f32_to_f16": [_p, _p, _i, _i, _i, _f, _p],
    "fmri_rows_f16_to_f32": [_p, _p, _i, _i, _i, _f, _p],
    "fmri_reduce_slabs": [_p, _i, _l, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "fmri_permute_chw": [_p, _p, _i, _i, _i, _f, _i, _p],
    "fmri_bn_stats": [_p, _i, _i, _p, _p, _l, _p],
    "fmri_bn_finalize": [_p, _i, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "fmri_bn_stats_finalize": [_p, _i, _i, _p, _p, _l, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "fmri_bn_cols_fwd": [_p, _p, _i, _i, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "fmri_bn_cols_fwd_s": [_p, _p, _i, _i, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "fmri_bn_finalize_s": [_p, _i, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "fmri_latent_fwd_ranged": [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _f, _i, _p],
    "fmri_rows_absmax": [_p, _l, _p, _p],
    "fmri_bn_cols_bwd": [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _i, _p, _p, _p, _f, _i, _p],
    "fmri_bn_fold_finalize": [_p, _i, _i, _p, _p, _f, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "fmri_bn_fold": [_p, _i, _i, _p, _p, _p],
    "fmri_bn_bwd_fold": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p],
    "fmri_bn_apply": [_p, _p, _i, _i, _p, _p, _i, _p],
    "fmri_bn_bwd_reduce": [_p, _p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _l, _p, _p, _f, _p],
    "fmri_bn_bwd_apply": [_p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _i, _p, _p],
    "fmri_bn_bwd_reduce2": [_p, _p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _l, _p, _p, _f, _i, _p],
    "fmri_bn_bwd_apply2": [_p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _i, _p, _p],
    "fmri_act_bwd": [_p, _p, _p, _i, _i, _i, _p, _p, _l, _p, _i, _f, _p],
    "fmri_colsum_acc": [_p, _i, _i, _i, _l, _l, _f, _p, _p],
    "fmri_colsum_rows": [_p, _i, _i, _p, _p, _l, _p, _i, _f, _p],
    "fmri_latent_fwd": [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p],
    "fmri_latent_bwd": [_p, _p, _p, _i, _f, _f, _p, _i, _i, _f, _p, _p, _i, _p],
    "fmri_feat_mse": [_p, _i, _i, _p, _p, _p],
    "fmri_feat_mse_bwd": [_p, _i, _i, _p, _f, _p, _p],
    "fmri_pixel_sq": [_p, _p, _l, _i, _i, _p, _p, _f, _p],
    "fmri_gan_head": [_p, _i, _i, _p, _p, _p],
    "fmri_gan_head_bwd": [_p, _i, _i, _p, _i, _f, _p, _p],
    "fmri_gan_head_parts": [_p, _i, _i, _p, _p, _i, _p],
    "fmri_gan_head_bwd_parts": [_p, _i, _i, _p, _i, _f, _p, _i, _p],
    "fmri_wae_logloss": [_p, _i, _i, _i, _f, _p, _p, _p, _i, _f, _p],
    "fmri_mlp_fwd": [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "fmri_mlp_bwd": [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _p],
    "fmri_compose_gate": [_p, _p, _f, _f, _f, _f, _f, _i, _i, _i, _p],
    "fmri_axpby_f16": [_p, _p, _p, _l, _f, _f, _p, _p],
    "fmri_axpby2_f16": [_p, _p, _p, _l, _f, _f, _p, _p, _p],
    "fmri_compose_gate_dev": [_p, _p, _f, _f, _f, _p, _i, _i, _i, _i, _p],
    "fmri_counter_inc": [_p, _p],
    "fmri_rmsprop_dev": [_p, _p, _p, _l, _p, _f, _f, _f, _p, _f, _p, _p],
    "fmri_adam_dev": [_p, _p, _p, _p, _l, _p, _f, _f, _f, _p, _f, _p, _f, _p, _p],
    "fmri_sumsq": [_p, _l, _p, _p],
    "fmri_renorm": [_p, _p, _l, _f, _p, _f, _p, _p, _p],
    "fmri_sumsq_f64": [_p, _l, _p, _i, _p],
    "fmri_renorm_f64": [_p, _p, _l, _f, _p, _f, _p, _p, _p],
    "fmri_rmsprop": [_p, _p, _p, _l, _f, _f, _f, _f, _p, _f, _p, _p],
    "fmri_adam": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _f, _p, _f, _p, _p],
}

class Epilogue(C.Structure):
    """``fmri_epilogue`` of include/fmri_hip.h (BatchNorm statistics out of a contraction's epilogue)."""
    _fields_ = [("stat_part", C.c_void_p), ("stat_rows_cap", C.c_int32), ("stat_group_n", C.c_int32),
                ("bn_x", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("bn_mean", C.c_void_p * 4), ("bn_rstd", C.c_void_p * 4), ("bn_x_img0", C.c_int32 * 4),
                ("bn_relu", C.c_int32), ("reserved", C.c_int32), ("act_y", C.c_void_p),
                ("aff_scale", C.c_void_p), ("aff_shift", C.c_void_p), ("aff_relu", C.c_int32), ("reserved2", C.c_int32)]


EP_ACT_APPLIED = 0x40000000
EP_AFFINE_APPLIED = 0x20000000


EXPORTS = sorted(list(_SIGS) + ["fmri_version", "fmri_last_error_string", "fmri_test_fastdiv", "fmri_bn_ws_floats",
                             "fmri_bn_fold_scratch_floats", "fmri_resize_coeffs", "fmri_latent_range_scale"])

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m fmri_hip.build` (hipcc --offload-arch=gfx950). "
            "The engine has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    lib.fmri_version.restype = _i
    lib.fmri_last_error_string.restype = C.c_char_p
    lib.fmri_last_error_string.argtypes = [_i]
    lib.fmri_test_fastdiv.restype = C.c_uint32
    lib.fmri_test_fastdiv.argtypes = [C.c_uint32, C.c_uint32]
    lib.fmri_bn_ws_floats.restype = _l
    lib.fmri_bn_ws_floats.argtypes = [_i, _i]
    lib.fmri_bn_fold_scratch_floats.restype = _i
    lib.fmri_bn_fold_scratch_floats.argtypes = [_i]
    lib.fmri_resize_coeffs.restype = _i
    lib.fmri_resize_coeffs.argtypes = [_i, _i, _p, _p, _i]
    lib.fmri_latent_range_scale.restype = _f
    lib.fmri_latent_range_scale.argtypes = [_f, _f]
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current stream on the current device (the fast private accessor when torch has it:
    ``torch.cuda.current_stream()`` costs ~8 us per call, which adds up over ~370 launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(code, what=""):
    if code != 0:
        msg = load().fmri_last_error_string(code).decode()
        raise RuntimeError(f"fmri_hip: {what} failed: {msg} ({code})")


# bench.py sets PROFILE to a list: every entry-point call is then bracketed by HIP events on the stream it is launched
# on and recorded as (entry point, note, start event, end event).  ``note(...)`` attaches what the caller knows about the
# NEXT call -- the kernel the library routes the geometry to, its algorithmic FLOPs and bytes -- so that bench.py can
# group launches into kernel families and price them against their rooflines.
PROFILE = None
_NOTE = None


def note(**kw):
    global _NOTE
    if PROFILE is not None:
        _NOTE = kw


def call(name, *args):
    """Invoke an entry point on torch's current stream and raise on error."""
    global _NOTE
    lib = load()
    if PROFILE is None:
        code = getattr(lib, name)(*args, stream())
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        code = getattr(lib, name)(*args, stream())
        e1.record()
        PROFILE.append((name, _NOTE, e0, e1))
        _NOTE = None
    if code != 0:
        check(code, name)


def tconv_class(k, pad, cy, cx, ci, rows_pad):
    lib = load()
    py, px, th, tw, kpad = _i(), _i(), _i(), _i(), _i()
    woff = _l()
    check(lib.fmri_tconv_class(k, pad, cy, cx, ci, rows_pad, py, px, th, tw, kpad, woff), "fmri_tconv_class")
    return dict(py=py.value, px=px.value, th=th.value, tw=tw.value, kpad=kpad.value, w_off=woff.value)


def kpad(taps, ci):
    return load().fmri_kpad(taps, ci)
