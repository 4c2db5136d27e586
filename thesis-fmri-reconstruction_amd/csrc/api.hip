// extern "C" surface of libfmri_hip.so (declared in include/fmri_hip.h).
// Host-side geometry derivation + argument validation; all device work is enqueued on the caller's stream.
#include "../../include/fmri_hip.h"
#include "kernels.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <cstdlib>

using namespace fmri;

namespace {
inline hipStream_t S(void* s) { return (hipStream_t)s; }
inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct TClass { int py, px, th, tw, dy0, dx0, kpad; int64_t w_off; };

// parity classes of a k x k stride-2 pad-p transposed convolution (see csrc/igemm.hip header)
void tconv_classes(int k, int pad, int ci, int rows_pad, TClass out[4]) {
    int64_t off = 0;
    for (int cy = 0; cy < 2; ++cy)
        for (int cx = 0; cx < 2; ++cx) {
            TClass& c = out[cy * 2 + cx];
            c.py = (cy + pad) & 1;
            c.px = (cx + pad) & 1;
            c.th = (k - c.py + 1) / 2;
            c.tw = (k - c.px + 1) / 2;
            c.dy0 = (cy + pad - c.py) / 2;
            c.dx0 = (cx + pad - c.px) / 2;
            c.kpad = pad_to(c.th * c.tw * ci, 64);
            c.w_off = off;
            off += (int64_t)rows_pad * c.kpad;
        }
}
}  // namespace

// ---- routing probe (kernels.h): thread-local, so that fmri_igemm_route is re-entrant like every other entry point
namespace {
thread_local bool t_probe_on = false;
thread_local char t_probe_name[128];
}  // namespace
namespace fmri {
bool route_probe(const char* fmt, ...) {
    if (!t_probe_on) return false;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_probe_name, sizeof(t_probe_name), fmt, ap);
    va_end(ap);
    return true;
}
}  // namespace fmri

extern "C" {

/* ---- batch ingest (tail of the image transforms of train_vgan_stage1.py:162-170, data_loader.py:186-217,374-401) ---- */
int fmri_ingest_u8(const uint8_t* src, int N, int H, int W, int C, const int* flip_dev, const int* shift_dev,
                   float mean0, float mean1, float mean2, float std0, float std1, float std2, void* dst16,
                   float* dst32, void* stream) {
    if (!src || N < 1 || H < 1 || W < 1 || (C != 1 && C != 3) || (!dst16 && !dst32) || std0 == 0.f || std1 == 0.f ||
        std2 == 0.f)
        return FMRI_E_BADARG;
    const float m[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
    return ingest_u8_launch(src, N, H, W, C, flip_dev, shift_dev, m, sd, (half_t*)dst16, dst32, S(stream));
}

/* ---- head of the image transforms: CenterCrop + Resize (train_vgan_stage1.py:162-165) --------------------------- */
int fmri_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize_cap) {
    if (in_size < 1 || out_size < 1 || !bounds || !coef || ksize_cap < 1) return FMRI_E_BADARG;
    if ((double)in_size / out_size > 30.0) return FMRI_E_UNSUPPORTED;       /* 64 taps per output pixel at most */
    const int r = resize_coeffs(in_size, out_size, bounds, coef, ksize_cap);
    return r < 0 ? FMRI_E_WORKSPACE : r;
}
int fmri_crop_resize_u8(const uint8_t* pool, const int64_t* offsets_dev, const int32_t* dims_dev, int N, int crop, int size,
                        const int32_t* hb_dev, const int32_t* hk_dev, int hks, const int32_t* vb_dev, const int32_t* vk_dev,
                        int vks, int vcount_max, uint8_t* out, void* stream) {
    if (!pool || !offsets_dev || !dims_dev || N < 1 || crop < 1 || size < 1 || !out || hks < 0 || vks < 0) return FMRI_E_BADARG;
    if ((hks > 0 && (!hb_dev || !hk_dev)) || (vks > 0 && (!vb_dev || !vk_dev)) || vcount_max < 1 || vcount_max > vks + (vks == 0))
        return FMRI_E_BADARG;
    if ((hks == 0 || vks == 0) && crop != size) return FMRI_E_BADARG;
    return crop_resize_u8_launch(pool, offsets_dev, dims_dev, N, crop, size, hb_dev, hk_dev, hks, vb_dev, vk_dev, vks, vcount_max,
                                 out, S(stream));
}

/* ---- evaluation metrics (train/train_utils.py:267-292, :295-420) ---- */
int fmri_pcc(const float* pred, const float* truth, int64_t n, double* ws5, float* out, void* stream) {
    if (!pred || !truth || !ws5 || !out || n < 2) return FMRI_E_BADARG;
    return pcc_launch(pred, truth, n, ws5, out, S(stream));
}
int fmri_ssim(const float* img1, const float* img2, int planes, int H, int W, double* ws2, float* ssim,
              float* contrast, void* stream) {
    if (!img1 || !img2 || !ws2 || planes < 1 || H < 1 || W < 1 || (!ssim && !contrast)) return FMRI_E_BADARG;
    return ssim_launch(img1, img2, planes, H, W, ws2, ssim, contrast, S(stream));
}

int fmri_version(void) { return 100; }

const char* fmri_last_error_string(int code) {
    switch (code) {
        case FMRI_OK: return "ok";
        case FMRI_E_BADARG: return "bad argument (shape/alignment constraint violated)";
        case FMRI_E_UNSUPPORTED: return "unsupported configuration";
        case FMRI_E_LAUNCH: return "HIP kernel launch failed";
        case FMRI_E_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

uint32_t fmri_test_fastdiv(uint32_t n, uint32_t d) { return fd_div(n, make_fastdiv(d)); }

int fmri_kpad(int taps, int ci) { return pad_to(taps * ci, 64); }

int fmri_tconv_class(int k, int pad, int cy, int cx, int ci, int rows_pad, int* py, int* px, int* th, int* tw,
                     int* kpad, int64_t* w_off) {
    if (cy < 0 || cy > 1 || cx < 0 || cx > 1 || k < 1 || ci < 1) return FMRI_E_BADARG;
    TClass c[4];
    tconv_classes(k, pad, ci, rows_pad, c);
    const TClass& t = c[cy * 2 + cx];
    *py = t.py; *px = t.px; *th = t.th; *tw = t.tw; *kpad = t.kpad; *w_off = t.w_off;
    return FMRI_OK;
}

int fmri_pack_weight(const float* src, void* dst, int64_t sa, int64_t sta, int64_t sb, int64_t stb, int A, int TA,
                     int B, int KW, int py, int px, int step, int TH, int TW, int rows_pad, int kpad, void* stream) {
    if (!src || !dst || A < 1 || TA < 1 || B < 1 || TH < 1 || TW < 1) return FMRI_E_BADARG;
    PackArgs p;
    p.src = src; p.dst = (half_t*)dst; p.sa = sa; p.sta = sta; p.sb = sb; p.stb = stb;
    p.A = A; p.TA = TA; p.B = B; p.Bp = pad_to(B, 8);
    p.KW = KW; p.py = py; p.px = px; p.step = step; p.TH = TH; p.TW = TW;
    p.rows_pad = rows_pad; p.kpad = kpad;
    if (rows_pad < TA * A || kpad < TH * TW * p.Bp) return FMRI_E_BADARG;
    return pack_weight_launch(p, S(stream));
}

int fmri_pack_entry_bytes(void) { return (int)sizeof(PackEntry); }

// Fills one host-side table row; returns the number of blocks the row occupies (tile_begin of the next row =
// tile_begin + that), 0 if this weight/orientation is not eligible for the batched kernel, < 0 on bad arguments.
int fmri_pack_entry_fill(void* host_entry, const float* src, void* dst, int64_t sa, int64_t sta, int64_t sb,
                         int64_t stb, int A, int TA, int B, int KW, int py, int px, int step, int TH, int TW,
                         int rows_pad, int kpad, int tile_begin) {
    if (!host_entry || !src || !dst || A < 1 || TA < 1 || B < 1 || TH < 1 || TW < 1) return FMRI_E_BADARG;
    PackEntry e;
    memset(&e, 0, sizeof(e));
    PackArgs& p = e.p;
    p.src = src; p.dst = (half_t*)dst; p.sa = sa; p.sta = sta; p.sb = sb; p.stb = stb;
    p.A = A; p.TA = TA; p.B = B; p.Bp = pad_to(B, 8);
    p.KW = KW; p.py = py; p.px = px; p.step = step; p.TH = TH; p.TW = TW;
    p.rows_pad = rows_pad; p.kpad = kpad;
    if (rows_pad < TA * A || kpad < TH * TW * p.Bp) return FMRI_E_BADARG;
    const int tiles = pack_tile_count(p, &e.run);
    e.tile_begin = tile_begin;
    memcpy(host_entry, &e, sizeof(e));
    return tiles;
}

int fmri_pack_weight_batch(const void* table_dev, int n, int total_tiles, void* stream) {
    if (!table_dev || n < 0 || total_tiles < 0) return FMRI_E_BADARG;
    return pack_batch_launch((const PackEntry*)table_dev, n, total_tiles, S(stream));
}

int fmri_transpose_f16(const void* src, void* dst, int R, int C, int src_rows, int ld_src, int ld_dst, void* stream) {
    if (!src || !dst || R < 1 || C < 1 || src_rows < R || ld_src < C || ld_dst < R || (ld_src & 7) || (ld_dst & 7) ||
        ld_dst < ((R + 7) & ~7))
        return FMRI_E_BADARG;
    return transpose_f16_launch((const half_t*)src, (half_t*)dst, R, C, src_rows, ld_src, ld_dst, S(stream));
}

int fmri_transpose_entry_bytes(void) { return (int)sizeof(TransposeEntry); }

int fmri_transpose_entry_fill(void* host_entry, const void* src, void* dst, int R, int C, int src_rows, int width,
                              int ld_src, int ld_dst, int tile_begin) {
    if (!host_entry || !src || !dst || R < 1 || C < 1 || src_rows < R || width < C || ld_src < width || (ld_src & 7) ||
        (ld_dst & 7) || ld_dst < ((R + 7) & ~7) || tile_begin < 0 || (((uintptr_t)src | (uintptr_t)dst) & 15))
        return FMRI_E_BADARG;
    TransposeEntry e;
    memset(&e, 0, sizeof(e));
    e.src = (const half_t*)src; e.dst = (half_t*)dst; e.R = R; e.C = C; e.Rbuf = src_rows; e.width = width;
    e.lds = ld_src; e.ldd = ld_dst; e.tile_begin = tile_begin;
    memcpy(host_entry, &e, sizeof(e));
    return ((R + 63) / 64) * ((C + 63) / 64);
}

int fmri_transpose_f16_batch(const void* table_dev, int n, int total_tiles, void* stream) {
    if (!table_dev || n < 0 || total_tiles < 0) return FMRI_E_BADARG;
    return transpose_batch_launch((const TransposeEntry*)table_dev, n, total_tiles, S(stream));
}

int fmri_apply_entry_bytes(void) { return (int)sizeof(ApplyEntry); }

int fmri_apply_entry_fill(void* host_entry, const float* gsrc, float* w, float* sq, float* grad, void* pk, int64_t sa,
                          int64_t sta, int64_t sb, int64_t stb, int A, int TA, int B, int KW, int py, int px, int step,
                          int TH, int TW, int ld, int kpad, int nslabs, int64_t slab_stride, int clear, float scale,
                          int64_t flat_n, int tile_begin) {
    if (!host_entry || !w || !sq || tile_begin < 0) return FMRI_E_BADARG;
    ApplyEntry e;
    memset(&e, 0, sizeof(e));
    e.w = w; e.sq = sq; e.grad = grad; e.tile_begin = tile_begin;
    if (flat_n > 0) {
        if (!grad) return FMRI_E_BADARG;
        e.kind = 2; e.n = flat_n;
    } else {
        if (!gsrc || A < 1 || TA < 1 || B < 1 || TH < 1 || TW < 1 || nslabs < 1) return FMRI_E_BADARG;
        e.gsrc = gsrc; e.pk = (half_t*)pk; e.sa = sa; e.sta = sta; e.sb = sb; e.slab_stride = slab_stride;
        e.A = A; e.TA = TA; e.B = B; e.Bp = pad_to(B, 8); e.ld = ld; e.kpad = kpad; e.nslabs = nslabs;
        e.clear = clear ? 1 : 0; e.scale = scale;
        if (ld < TH * TW * e.Bp || (pk && kpad < TH * TW * e.Bp)) return FMRI_E_BADARG;
    }
    const int tiles = apply_entry_tiles(e, TH, TW, KW, py, px, step, stb);
    memcpy(host_entry, &e, sizeof(e));
    return tiles;
}

int fmri_apply_batch(const void* table_dev, int n, int total_tiles, int mode, const float* lr_dev, float alpha, float eps,
                     float gscale, const float* gdev, float clamp, const int* flag, int gated, void* stream) {
    if (!table_dev || n < 0 || total_tiles < 0 || mode < 0 || mode > 3 || ((mode == 1 || mode == 3) && !lr_dev)) return FMRI_E_BADARG;
    ApplyOpt o;
    o.lr_dev = lr_dev; o.gdev = gdev; o.flag = flag; o.alpha = alpha; o.eps = eps; o.gscale = gscale; o.clamp = clamp;
    o.mode = mode; o.gated = gated ? 1 : 0;
    return apply_batch_launch((const ApplyEntry*)table_dev, n, total_tiles, o, S(stream));
}

int fmri_unpack_grad(const float* src, float* dst, int64_t sa, int64_t sta, int64_t sb, int64_t stb, int A, int TA,
                     int B, int KW, int py, int px, int step, int TH, int TW, int ld, float scale, int accumulate,
                     int nslabs, int64_t slab_stride, void* stream) {
    if (!src || !dst || A < 1 || TA < 1 || B < 1 || nslabs < 1) return FMRI_E_BADARG;
    UnpackArgs p;
    p.src = src; p.dst = dst; p.sa = sa; p.sta = sta; p.sb = sb; p.stb = stb;
    p.A = A; p.TA = TA; p.B = B; p.Bp = pad_to(B, 8);
    p.KW = KW; p.py = py; p.px = px; p.step = step; p.TH = TH; p.TW = TW;
    p.ld = ld; p.scale = scale; p.accumulate = accumulate; p.nslabs = nslabs; p.slab_stride = slab_stride;
    if (ld < TH * TW * p.Bp) return FMRI_E_BADARG;
    return unpack_grad_launch(p, S(stream));
}

int fmri_igemm(const void* in, const void* w, void* out, const float* bias, const void* zero16, int N, int Hi, int Wi,
               int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad, int mode, int act,
               int out_f32, int splits, int64_t slab_stride, int bn_tile, void* stream) {
    return fmri_igemm_ep(in, w, out, bias, zero16, N, Hi, Wi, Ci, Ho, Wo, CoStore, Co, k, stride, pad, mode, act, out_f32,
                         splits, slab_stride, bn_tile, 0, nullptr, nullptr, stream);
}

int fmri_igemm_route(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad,
                     int mode, int act, int out_f32, int splits, int bn_tile, int64_t w_elems, int has_bias,
                     int stat_rows_cap, int stat_group_n, int want_bn_bwd, int want_act_y, int want_affine, char* name_out,
                     int cap) {
    if (!name_out || cap < 2) return FMRI_E_BADARG;
    name_out[0] = 0;
    // the routing decisions read sizes, flags and NULL-ness only: every pointer is a non-NULL dummy that is never
    // dereferenced, every launcher stops at route_probe()
    void* const dummy = (void*)(uintptr_t)64;
    fmri_epilogue e;
    memset(&e, 0, sizeof(e));
    if (stat_rows_cap > 0) { e.stat_part = (float*)dummy; e.stat_rows_cap = stat_rows_cap; e.stat_group_n = stat_group_n; }
    if (want_bn_bwd) {
        e.bn_x = dummy; e.bn_gamma = (const float*)dummy; e.bn_beta = (const float*)dummy; e.bn_relu = 1;
        for (int i = 0; i < 4; ++i) { e.bn_mean[i] = (const float*)dummy; e.bn_rstd[i] = (const float*)dummy; }
    }
    if (want_act_y) e.act_y = dummy;
    if (want_affine) { e.aff_scale = (const float*)dummy; e.aff_shift = (const float*)dummy; }
    const bool any_ep = stat_rows_cap > 0 || want_bn_bwd || want_act_y || want_affine;
    int done = 0;
    t_probe_name[0] = 0;
    t_probe_on = true;
    const int r = fmri_igemm_ep(dummy, dummy, dummy, has_bias ? (const float*)dummy : nullptr, dummy, N, Hi, Wi, Ci, Ho, Wo,
                                CoStore, Co, k, stride, pad, mode, act, out_f32, splits, 0, bn_tile, w_elems,
                                any_ep ? &e : nullptr, &done, nullptr);
    t_probe_on = false;
    if (r != FMRI_OK) return r;
    snprintf(name_out, (size_t)cap, "%s", t_probe_name[0] ? t_probe_name : "none");
    return FMRI_OK;
}

int fmri_igemm_ep(const void* in, const void* w, void* out, const float* bias, const void* zero16, int N, int Hi,
                  int Wi, int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad, int mode, int act,
                  int out_f32, int splits, int64_t slab_stride, int bn_tile, int64_t w_elems, const fmri_epilogue* ep,
                  int* ep_done, void* stream) {
    if (ep_done) *ep_done = 0;
    if (!in || !w || !out || !zero16) return FMRI_E_BADARG;
    StatEpi se;
    se.part = nullptr; se.rows_cap = 0; se.C = CoStore; se.group_n = 0;
    se.tpg[0] = se.tpg[1] = se.tpg[2] = se.tpg[3] = 0;
    BnBwdEpi bb;
    memset(&bb, 0, sizeof(bb));
    AffEpi aff;
    memset(&aff, 0, sizeof(aff));
    if (ep && ep->aff_scale) {
        if (!ep->aff_shift || ep->stat_part || bias || act != FMRI_ACT_NONE || out_f32) return FMRI_E_BADARG;
        aff.scale = ep->aff_scale; aff.shift = ep->aff_shift; aff.relu = ep->aff_relu ? 1 : 0;
    }
    if (ep && ep->stat_part) {
        if (ep->stat_rows_cap < 1 || ep->stat_group_n < 0 || out_f32 || (ep->stat_group_n > 0 && N % ep->stat_group_n))
            return FMRI_E_BADARG;
        se.part = ep->stat_part; se.rows_cap = ep->stat_rows_cap; se.group_n = ep->stat_group_n;
        if (ep->bn_x) {
            // BatchNorm backward: <= 4 groups, every used group fully described; no bias / activation
            const int groups = se.group_n > 0 ? N / se.group_n : 1;
            if (groups > 4 || !ep->bn_gamma || !ep->bn_beta || bias || act != FMRI_ACT_NONE || Co != CoStore)
                return FMRI_E_BADARG;
            bb.x = (const half_t*)ep->bn_x; bb.gamma = ep->bn_gamma; bb.beta = ep->bn_beta; bb.relu = ep->bn_relu;
            for (int i = 0; i < groups; ++i) {
                if (!ep->bn_mean[i] || !ep->bn_rstd[i] || ep->bn_x_img0[i] < 0) return FMRI_E_BADARG;
                bb.mean[i] = ep->bn_mean[i]; bb.rstd[i] = ep->bn_rstd[i]; bb.x_img0[i] = ep->bn_x_img0[i];
            }
        }
    } else if (ep && ep->bn_x) {
        return FMRI_E_BADARG;
    }
    if (N < 1 || Ci < 8 || (Ci & 7) || CoStore < 4 || (CoStore & 3) || Co < 1 || Co > CoStore) return FMRI_E_BADARG;
    if (bn_tile != 32 && bn_tile != 64 && bn_tile != 128) return FMRI_E_UNSUPPORTED;
    if (splits < 1) return FMRI_E_BADARG;
    if (splits > 1 && !out_f32) return FMRI_E_BADARG;
    const int copad = pad_to(Co, bn_tile);
    if (copad < CoStore) return FMRI_E_BADARG;   // every stored channel must be covered by a tile
    IgemmArgs a;
    a.in = (const half_t*)in; a.w = (const half_t*)w; a.out = out; a.bias = bias; a.zero = (const half_t*)zero16;
    a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.CoStore = CoStore; a.Co = Co;
    a.act = act; a.splits = splits; a.slab_stride = slab_stride;
    a.st = se; a.st.part = nullptr;
    a.bb = bb;
    a.fdCi = make_fastdiv((uint32_t)Ci);
    a.fdCpt = make_fastdiv((uint32_t)(Ci >= 64 ? Ci / 64 : 1));
    int maxM = 0;
    auto set_class = [&](IgemmClass& c, int Yc, int Xc, int oy0, int ox0, int TH, int TW, int dy0, int dx0, int dstep,
                         int kpad, int64_t w_off) -> bool {
        c.Yc = Yc; c.Xc = Xc; c.oy0 = oy0; c.ox0 = ox0; c.T = TH * TW; c.TW = TW;
        c.dy0 = dy0; c.dx0 = dx0; c.dstep = dstep;
        const int64_t M = (int64_t)N * Yc * Xc;
        if (M > 0x7fffff00LL) return false;
        c.M = (int)(M > 0 ? M : 0);
        c.Kpad = kpad; c.ksteps = kpad / 64; c.w_off = w_off;
        c.fdX = make_fastdiv((uint32_t)(Xc > 0 ? Xc : 1));
        c.fdYX = make_fastdiv((uint32_t)(Yc * Xc > 0 ? Yc * Xc : 1));
        c.fdTW = make_fastdiv((uint32_t)TW);
        if (c.M > maxM) maxM = c.M;
        return true;
    };
    if (mode == FMRI_CONV || mode == FMRI_CONV_FLIP) {
        if (mode == FMRI_CONV_FLIP && stride != 1) return FMRI_E_UNSUPPORTED;
        a.s = stride; a.os = 1; a.ncls = 1;
        const int d0 = mode == FMRI_CONV ? -pad : pad;
        const int ds = mode == FMRI_CONV ? 1 : -1;
        if (!set_class(a.cls[0], Ho, Wo, 0, 0, k, k, d0, d0, ds, pad_to(k * k * Ci, 64), 0)) return FMRI_E_BADARG;
        for (int i = 1; i < 4; ++i) a.cls[i] = a.cls[0];
    } else if (mode == FMRI_TCONV2) {
        if (stride != 2) return FMRI_E_UNSUPPORTED;
        a.s = 1; a.os = 2; a.ncls = 4;
        TClass tc[4];
        tconv_classes(k, pad, Ci, copad, tc);
        for (int cy = 0; cy < 2; ++cy)
            for (int cx = 0; cx < 2; ++cx) {
                const TClass& t = tc[cy * 2 + cx];
                const int Yc = (Ho - cy + 1) / 2, Xc = (Wo - cx + 1) / 2;
                if (!set_class(a.cls[cy * 2 + cx], Yc, Xc, cy, cx, t.th, t.tw, t.dy0, t.dx0, -1, t.kpad, t.w_off))
                    return FMRI_E_BADARG;
            }
    } else {
        return FMRI_E_UNSUPPORTED;
    }
    // no empty split: every split must own >= 1 K-step in every class
    for (int i = 0; i < a.ncls; ++i) {
        const int per = (a.cls[i].ksteps + splits - 1) / splits;
        if (per * (splits - 1) >= a.cls[i].ksteps && splits > 1) return FMRI_E_BADARG;
    }
    if (maxM == 0) return FMRI_OK;
    // 5x5 stride-1 convolutions between 3(8)- and 32-channel maps -> register-resident-weight kernel
    // (csrc/igemm_narrow.hip); FMRI_NARROW=off disables
    static const char* nar_env = getenv("FMRI_NARROW");
    static const bool no_narrow = nar_env && !strcmp(nar_env, "off");
    if (!no_narrow && (mode == FMRI_CONV || mode == FMRI_CONV_FLIP) && stride == 1 && k == 5 && pad == 2 &&
        (Ci == 8 || Ci == 32) && Co <= 32 && !out_f32 && splits == 1 && Hi == Ho && Wi == Wo &&
        (int64_t)N * Hi * Wi * 32 < 0x7fffffffLL) {
        const int co_tiles = Co <= 16 ? 1 : 2;
        if ((Ci == 32 && co_tiles == 1) || Ci == 8) {
            NarrowArgs q;
            q.in = a.in; q.w = a.w + a.cls[0].w_off; q.out = (half_t*)out; q.bias = bias;
            q.N = N; q.H = Hi; q.W = Wi; q.CoStore = CoStore; q.Co = Co; q.Kpad = a.cls[0].Kpad; q.act = act;
            q.tiles_y = (Hi + 15) / 16; q.tiles_x = (Wi + 15) / 16;
            q.ntiles = N * q.tiles_y * q.tiles_x;
            if (copad >= co_tiles * 16 && q.Kpad >= (Ci == 32 ? 800 : 224)) {
                const int r = igemm_narrow_launch(q, Ci, co_tiles, mode == FMRI_CONV_FLIP, S(stream));
                if (r != E_UNSUPPORTED) return r;
            }
        }
    }
    // stride-2 convolution k5 p2, Ci % 32 == 0, 128-channel tiles, no bias / activation -> window-resident kernel
    // (csrc/igemm_c5.hip); FMRI_C5=off disables
    static const char* c5_env = getenv("FMRI_C5");
    static const bool no_c5 = c5_env && !strcmp(c5_env, "off");
    if (!no_c5 && mode == FMRI_CONV && stride == 2 && k == 5 && pad == 2 && (Ci & 31) == 0 && bn_tile == 128 &&
        (copad & 127) == 0 && !out_f32 && splits == 1 && !bias && act == FMRI_ACT_NONE &&
        Ho == (Hi - 1) / 2 + 1 && Wo == (Wi - 1) / 2 + 1 && a.cls[0].Kpad >= 25 * Ci &&
        (int64_t)N * Hi * Wi * Ci * 2 < 0x7fffffffLL && (int64_t)copad * a.cls[0].Kpad * 2 < 0xffffffffLL &&
        (w_elems == 0 || w_elems >= (int64_t)copad * a.cls[0].Kpad)) {
        C5Args q;
        q.in = a.in; q.w = a.w + a.cls[0].w_off; q.out = (half_t*)out;
        q.N = N; q.Hi = Hi; q.Wi = Wi; q.Ci = Ci; q.Ho = Ho; q.Wo = Wo; q.CoStore = CoStore; q.Co = Co;
        q.Kpad = a.cls[0].Kpad; q.nsub = Ci / 32;
        q.pw16 = Wo > 8 ? 1 : 0;
        const int ipb = q.pw16 ? 1 : 2;
        q.tiles_x = q.pw16 ? (Wo + 15) / 16 : 1;
        q.tiles_y = (Ho + 7) / 8;
        q.ntiles = ((N + ipb - 1) / ipb) * q.tiles_y * q.tiles_x;
        q.in_bytes = (uint32_t)((int64_t)N * Hi * Wi * Ci * 2);
        q.w_bytes = (uint32_t)((int64_t)copad * q.Kpad * 2);
        q.fdTPI = make_fastdiv((uint32_t)(q.tiles_y * q.tiles_x));
        q.fdTX = make_fastdiv((uint32_t)q.tiles_x);
        q.st = se;
        // persistent blocks: tpb consecutive tiles each when that makes whole rounds of 2 blocks per CU
        const int ncol = copad / 128;
        q.tpb = (q.ntiles * ncol) / 512;      // whole rounds only: fewer, longer blocks leave CUs idle
        if (q.tpb < 1) q.tpb = 1;
        // statistics: one row per block; statistics groups must not share a tile, nor a block
        const int tpg5 = se.group_n > 0 ? (se.group_n / ipb) * q.tiles_y * q.tiles_x : q.ntiles;
        if (se.part && se.group_n > 0) {
            if (se.group_n % ipb) q.st.part = nullptr;
            else while (tpg5 % q.tpb) --q.tpb;
        }
        if (se.part) {
            q.st.tpg[0] = (tpg5 + q.tpb - 1) / q.tpb;
            if (q.st.tpg[0] > se.rows_cap) q.st.part = nullptr;
        }
        q.bb = bb;
        if (!q.st.part) q.bb.x = nullptr;       // no statistics rows: plain output (*ep_done = 0 tells the caller)
        memset(&q.aff, 0, sizeof(q.aff));
        // wide form (csrc/igemm_c5w.hip): 16 x 16-pixel tiles of one image or 8 x 8-pixel tiles of four, one 8-wave block per
        // CU, loader / compute waves; FMRI_C5W=off disables
        static const char* c5w_env = getenv("FMRI_C5W");
        static const bool no_c5w = c5w_env && !strcmp(c5w_env, "off");
        const int ipbw = q.pw16 ? 1 : 4;
        // (gated on the caller's REQUEST for the BatchNorm-backward epilogue, bb.x: the wide kernel has none, and q.bb.x is
        // also cleared when the narrow form's rows did not fit -- the wide form, needing fewer rows, would then emit
        // FORWARD statistics rows that the caller reads as (sum g, sum g*xhat))
        if (!no_c5w && !bb.x && (q.pw16 ? Ho > 8 : true) &&
            !(se.part && se.group_n > 0 && (se.group_n % ipbw))) {
            C5Args w = q;
            const int ph = q.pw16 ? 16 : 8;
            w.tiles_y = (Ho + ph - 1) / ph;
            w.ntiles = ((N + ipbw - 1) / ipbw) * w.tiles_y * w.tiles_x;
            w.fdTPI = make_fastdiv((uint32_t)(w.tiles_y * w.tiles_x));
            w.st = se;
            const int tpgw = se.group_n > 0 ? (se.group_n / ipbw) * w.tiles_y * w.tiles_x : w.ntiles;
            // whole rounds of one block per CU, statistics groups not sharing a block
            w.tpb = 1;
            for (int t = (w.ntiles * ncol) / 256; t > 1; --t)
                if (tpgw % t == 0 && w.ntiles % t == 0 && ((w.ntiles / t) * ncol) % 256 == 0) { w.tpb = t; break; }
            if (se.part) {
                w.st.tpg[0] = (tpgw + w.tpb - 1) / w.tpb;
                if (w.st.tpg[0] > se.rows_cap) w.st.part = nullptr;
            }
            if (w.st.part || !se.part) {
                w.aff = aff;
                const int r = igemm_c5w_launch(w, copad, S(stream));
                if (r == OK && ep_done && w.st.part) *ep_done = w.st.tpg[0];
                if (r == OK && ep_done && w.aff.scale) *ep_done |= FMRI_EP_AFFINE_APPLIED;
                if (r != E_UNSUPPORTED) return r;
            }
        }
        const int r = igemm_c5_launch(q, copad, S(stream));
        if (r == OK && ep_done && q.st.part) *ep_done = q.st.tpg[0];
        if (r != E_UNSUPPORTED) return r;
    }
    // stride-2 transposed convolution k5 p2, Ci % 128 == 0, >= 64 output channels, no bias / activation: all four parity
    // classes per block (csrc/igemm_tc5.hip); FMRI_TC5=off disables
    static const char* tc5_env = getenv("FMRI_TC5");
    static const bool no_tc5 = tc5_env && !strcmp(tc5_env, "off");
    if (!no_tc5 && mode == FMRI_TCONV2 && k == 5 && pad == 2 && (Ci & 127) == 0 && bn_tile >= 64 && !out_f32 &&
        splits == 1 && !bias && act == FMRI_ACT_NONE && w_elems > 0 && w_elems * 2 < 0xffffffffLL &&
        (int64_t)N * Hi * Wi * Ci * 2 < 0x7fffffffLL) {
        Tc5Args q;
        q.in = a.in; q.w = a.w; q.out = (half_t*)out; q.bias = nullptr;
        q.N = N; q.Hi = Hi; q.Wi = Wi; q.Ci = Ci; q.Ho = Ho; q.Wo = Wo; q.CoStore = CoStore; q.Co = Co;
        q.act = act; q.nchunks = Ci / 64;
        bool ok = true;
        for (int i = 0; i < 4; ++i) {
            const IgemmClass& s = a.cls[i];
            const int th = s.T / s.TW;
            if (th != ((i >> 1) ? 2 : 3) || s.TW != ((i & 1) ? 2 : 3) || s.dy0 != 1 || s.dx0 != 1 || s.dstep != -1 ||
                s.oy0 != (i >> 1) || s.ox0 != (i & 1) || s.Kpad < s.T * Ci)
                ok = false;
            q.cls[i].Yc = s.Yc; q.cls[i].Xc = s.Xc; q.cls[i].Kpad = s.Kpad; q.cls[i].pad0 = 0; q.cls[i].w_off = s.w_off;
            if ((s.w_off + (int64_t)copad * s.Kpad) > w_elems) ok = false;
        }
        const int Yc0 = a.cls[0].Yc, Xc0 = a.cls[0].Xc;          // class (0, 0) has the largest grid
        q.pw_log2 = Xc0 > 8 ? 4 : 3;
        q.ph_log2 = (q.pw_log2 == 3 && Yc0 > 8) ? 4 : 3;
        q.PH = 1 << q.ph_log2;
        q.IPB = 128 >> (q.pw_log2 + q.ph_log2);
        q.IH = q.PH + 2;
        q.IW = (1 << q.pw_log2) + 2;
        q.tiles_x = (Xc0 + (1 << q.pw_log2) - 1) >> q.pw_log2;
        q.tiles_y = (Yc0 + q.PH - 1) >> q.ph_log2;
        q.ntiles = Yc0 > 0 && Xc0 > 0 ? ((N + q.IPB - 1) / q.IPB) * q.tiles_y * q.tiles_x : 0;
        q.nslice = (q.IPB * q.IH * q.IW * 8 + 255) / 256;
        // whole class grid (and input) inside one 8 x 8 tile per image: the window's halo is all padding -> dense form
        if (q.IPB == 2 && q.tiles_x == 1 && q.tiles_y == 1 && Hi <= 8 && Wi <= 8) q.nslice = 4;
        q.in_bytes = (uint32_t)((int64_t)N * Hi * Wi * Ci * 2);
        q.w_bytes = (uint32_t)(w_elems * 2);
        q.fdTPI = make_fastdiv((uint32_t)(q.tiles_y * q.tiles_x));
        q.fdTX = make_fastdiv((uint32_t)q.tiles_x);
        q.fdIHW = make_fastdiv((uint32_t)(q.IH * q.IW));
        q.fdIW = make_fastdiv((uint32_t)q.IW);
        q.st = se;
        q.bb = bb;
        memset(&q.aff, 0, sizeof(q.aff));
        q.solo = 0; q.pad_solo = 0;
        // statistics: one row per tile; groups must not share a tile
        const int tpi5 = q.tiles_y * q.tiles_x;
        if (se.part) {
            if (se.group_n > 0 && (se.group_n % q.IPB)) q.st.part = nullptr;
            q.st.tpg[0] = se.group_n > 0 ? (se.group_n / q.IPB) * tpi5 : q.ntiles;
            if (q.st.tpg[0] > se.rows_cap) q.st.part = nullptr;
        }
        if (!q.st.part) q.bb.x = nullptr;       // no statistics rows: plain output (*ep_done = 0 tells the caller)
        // wide form (csrc/igemm_tc5w.hip): 16 x 16-position tiles, one 8-wave block per CU, loader / compute waves;
        // FMRI_TC5W=off disables
        static const char* tc5w_env = getenv("FMRI_TC5W");
        static const bool no_tc5w = tc5w_env && !strcmp(tc5w_env, "off");
        const bool wide16 = Xc0 > 8 && Yc0 > 8;
        const bool wide8 = Xc0 <= 8 && Yc0 <= 8 && Hi <= 8 && Wi <= 8 && !(se.part && se.group_n > 0 && (se.group_n & 3));
        if (ok && !no_tc5w && bn_tile == 128 && (wide16 || wide8) && !bb.x && !(q.nchunks & 1)) {      // bb.x: as above
            Tc5Args w = q;
            if (wide16) {
                w.pw_log2 = 4; w.ph_log2 = 4; w.PH = 16; w.IPB = 1; w.IH = 18; w.IW = 18; w.nslice = 11;
                w.tiles_x = (Xc0 + 15) / 16;
                w.tiles_y = (Yc0 + 15) / 16;
                w.ntiles = N * w.tiles_y * w.tiles_x;
            } else {
                w.pw_log2 = 3; w.ph_log2 = 3; w.PH = 8; w.IPB = 4; w.IH = 10; w.IW = 10; w.nslice = 13;
                w.tiles_x = w.tiles_y = 1;
                w.ntiles = (N + 3) / 4;
            }
            w.fdTPI = make_fastdiv((uint32_t)(w.tiles_y * w.tiles_x));
            w.fdTX = make_fastdiv((uint32_t)w.tiles_x);
            w.st = se;
            // few tiles (at most 128 tile x column-block pairs): one parity class per block, grid.z = 4
            w.solo = w.ntiles * (copad / 128) <= 128 ? 1 : 0;
            w.pad_solo = 0;
            if (se.part) {
                w.st.tpg[0] = (se.group_n > 0 ? (se.group_n / w.IPB) * w.tiles_y * w.tiles_x : w.ntiles) * (w.solo ? 4 : 1);
                if (w.st.tpg[0] > se.rows_cap) w.st.part = nullptr;
            }
            if (w.st.part || !se.part) {
                w.aff = aff;
                const int r = igemm_tc5w_launch(w, copad, S(stream));
                if (r == OK && ep_done && w.st.part) *ep_done = w.st.tpg[0];
                if (r == OK && ep_done && w.aff.scale) *ep_done |= FMRI_EP_AFFINE_APPLIED;
                if (r != E_UNSUPPORTED) return r;
            }
        }
        if (ok && q.ntiles > 0) {
            const int r = igemm_tc5_launch(q, bn_tile, copad, S(stream));
            if (r == OK && ep_done && q.st.part) *ep_done = q.st.tpg[0];
            if (r != E_UNSUPPORTED) return r;
        }
    }
    // stride-2 transposed convolution 128 -> <= 32 channels -> persistent register-resident-weight kernel
    // (csrc/igemm_tc32.hip); FMRI_TC32=off disables
    static const char* tc_env = getenv("FMRI_TC32");
    static const bool no_tc32 = tc_env && !strcmp(tc_env, "off");
    if (!no_tc32 && mode == FMRI_TCONV2 && Ci == 128 && CoStore <= 32 && !out_f32 && splits == 1 &&
        (int64_t)N * Hi * Wi * 128 < 0x7fffffffLL) {
        Tc32Args q;
        q.in = a.in; q.w = a.w; q.out = (half_t*)out; q.bias = bias;
        q.N = N; q.Hi = Hi; q.Wi = Wi; q.Ho = Ho; q.Wo = Wo; q.CoStore = CoStore; q.Co = Co; q.act = act;
        // the kernel hard-wires the k5 p2 class geometry: parity-0 classes have 3 taps from input offset +1 downwards,
        // parity-1 classes 2 taps from +1 downwards
        bool ok = copad >= 32 && k == 5 && pad == 2;
        for (int i = 0; i < 4 && ok; ++i) {
            const IgemmClass& s = a.cls[i];
            Tc32Class& d = q.cls[i];
            const int th = s.T / s.TW;
            if (th != ((i >> 1) ? 2 : 3) || s.TW != ((i & 1) ? 2 : 3) || s.dy0 != 1 || s.dx0 != 1 || s.dstep != -1 ||
                s.oy0 != (i >> 1) || s.ox0 != (i & 1) || s.Kpad < s.T * 128)
                ok = false;
            d.Yc = s.Yc; d.Xc = s.Xc; d.Kpad = s.Kpad; d.pad0 = 0; d.w_off = s.w_off;
        }
        q.tiles_y = (a.cls[0].Yc + 7) / 8; q.tiles_x = (a.cls[0].Xc + 15) / 16;     // class (0,0) has the largest grid
        q.ntiles = N * q.tiles_y * q.tiles_x;
        if (q.ntiles < 1) ok = false;
        const int begin = q.ntiles < 256 ? q.ntiles : 256;                          // one persistent block per CU
        q.relu_y = (ep && ep->act_y) ? (const half_t*)ep->act_y : nullptr;
        if (ok) {
            const int r = igemm_tc32_launch(q, begin, S(stream));       // E_UNSUPPORTED: bias / activation epilogue
            if (r == OK && ep_done && q.relu_y) *ep_done |= FMRI_EP_ACT_APPLIED;
            if (r != E_UNSUPPORTED) return r;
        }
    }
    // generic kernel: statistics epilogue (one row per row tile) when no tile straddles two statistics groups
    a.st = se;
    int prows = 0;
    if (se.part && !out_f32) {
        const int bm = igemm_bm(a, maxM, bn_tile, copad, false);
        for (int i = 0; i < a.ncls; ++i) {
            const int64_t mg = se.group_n > 0 ? (int64_t)se.group_n * a.cls[i].Yc * a.cls[i].Xc : a.cls[i].M;
            if (se.group_n > 0 && (mg % bm)) a.st.part = nullptr;
            a.st.tpg[i] = (int)((mg + bm - 1) / bm);
            prows += a.st.tpg[i];
        }
        if (prows > se.rows_cap) a.st.part = nullptr;
    } else {
        a.st.part = nullptr;
    }
    if (!a.st.part) a.bb.x = nullptr;           // no statistics rows: plain output (*ep_done = 0 tells the caller)
    const int r = igemm_launch(a, maxM, bn_tile, copad, out_f32 != 0, S(stream));
    if (r == OK && ep_done && a.st.part) *ep_done = prows;
    return r;
}

// K pieces (8 x 8 pixel tiles per block) of the four parity planes of fmri_wgrad's window kernel for a budget of `splits`
// blocks per (row block, column block) over the planes.  A K-step costs ~128 cycles per shift of the plane (8 MFMAs of one
// wave) + ~550 of barrier, fragment reads and waits (csrc/wgrad_win.hip, measured 1 825 / 1 430 / 1 033 cycles at 9 / 6 / 4
// shifts), so the planes get pieces in inverse proportion: all blocks of a launch finish together.  Returns the number of pieces of the plane with the most (= slabs written).
static int wgrad_plane_pieces(int N, int Yc, int Xc, int k, int pad, int splits, int tps_out[4]) {
    const int ntiles = N * ((Yc + 7) / 8) * ((Xc + 7) / 8);
    int nsh[2] = {0, 0};
    for (int t = 0; t < k; ++t) ++nsh[(t - pad) & 1];
    int total = splits < 4 ? 4 : splits;
    total -= total & 3;
    double cost[4], csum = 0;
    for (int pl = 0; pl < 4; ++pl) { cost[pl] = 128.0 * nsh[pl >> 1] * nsh[pl & 1] + 550.0; csum += cost[pl]; }
    int smax = 1;
    for (int pl = 0; pl < 4; ++pl) {
        int sp = (int)(total * cost[pl] / csum + 0.5);
        if (sp < 1) sp = 1;
        if (sp > ntiles) sp = ntiles > 0 ? ntiles : 1;
        const int tps = (ntiles + sp - 1) / sp;
        tps_out[pl] = tps < 1 ? 1 : tps;
        const int pieces = ntiles > 0 ? (ntiles + tps_out[pl] - 1) / tps_out[pl] : 1;
        if (pieces > smax) smax = pieces;
    }
    return smax;
}

// number of per-split slabs fmri_wgrad(..., atomic = 2) writes for a budget of `splits` blocks per tile group
int fmri_wgrad_slabs(int N, int Yc, int Xc, int k, int pad, int splits) {
    int tps[4];
    return wgrad_plane_pieces(N, Yc, Xc, k, pad, splits, tps);
}

static int wgrad_narrow_blocks(int N, int Yc, int Xc) {
    const int64_t ntiles = (int64_t)N * ((Yc + 7) / 8) * ((Xc + 7) / 8);
    int64_t nb = (ntiles + 31) / 32;                // >= 16 tiles per wave PAIR, at most three 4-wave blocks per CU
    if (nb > 768) nb = 768;
    if (nb < 1) nb = 1;
    return (int)nb;
}
int fmri_wgrad_narrow_blocks(int N, int Yc, int Xc) {
    return (N < 1 || Yc < 1 || Xc < 1) ? 0 : wgrad_narrow_blocks(N, Yc, Xc);
}

int fmri_set_deterministic(int on) {
    const int was = g_deterministic;
    g_deterministic = on ? 1 : 0;
    return was;
}
int fmri_get_deterministic(void) { return g_deterministic; }

int fmri_wgrad(const void* P, const void* Q, float* out, const void* zero16, int N, int Yc, int Xc, int A, int Hq,
               int Wq, int Bc, int k, int stride, int pad, int flip, int apad, int ba_tile, int ldo, int splits,
               int atomic, void* stream) {
    return fmri_wgrad_if(nullptr, P, Q, out, zero16, N, Yc, Xc, A, Hq, Wq, Bc, k, stride, pad, flip, apad, ba_tile, ldo,
                         splits, atomic, stream);
}

int fmri_wgrad_if(const int* gate, const void* P, const void* Q, float* out, const void* zero16, int N, int Yc, int Xc,
                  int A, int Hq, int Wq, int Bc, int k, int stride, int pad, int flip, int apad, int ba_tile, int ldo,
                  int splits, int atomic, void* stream) {
    if (!P || !Q || !out || !zero16) return FMRI_E_BADARG;
    if (flip && stride != 1) return FMRI_E_UNSUPPORTED;
    if (N < 1 || A < 8 || (A & 7) || Bc < 8 || (Bc & 7) || splits < 1) return FMRI_E_BADARG;
    if (ba_tile != 32 && ba_tile != 64 && ba_tile != 128) return FMRI_E_UNSUPPORTED;
    if (apad % ba_tile || apad < A) return FMRI_E_BADARG;
    const int T = k * k;
    if (ldo % 128 || ldo < T * Bc) return FMRI_E_BADARG;
    if (splits > 1 && !atomic) return FMRI_E_BADARG;
    if (atomic < 0 || atomic > 4) return FMRI_E_BADARG;
    const int64_t M = (int64_t)N * Yc * Xc;
    if (M < 1 || M > 0x7fffff00LL) return FMRI_E_BADARG;
    // stride-2 sampling, >= 128 rows, 32-channel column blocks, pre-zeroed fp32 output (atomic accumulation):
    // window-resident kernel (csrc/wgrad_win.hip).  FMRI_WGRAD_WIN=off disables.
    static const char* ww_env = getenv("FMRI_WGRAD_WIN");
    static const bool no_ww = ww_env && !strcmp(ww_env, "off");
    if (!no_ww && stride == 2 && !flip && (atomic == 1 || atomic == 2) && ba_tile == 128 && (Bc & 31) == 0 && Yc * Xc > 1 &&
        (int64_t)N * Hq * Wq * Bc < 0x7fffffffLL && M * A < 0x7fffffffLL) {
        WgradWinArgs w;
        w.gate = gate;
        w.P = (const half_t*)P; w.Q = (const half_t*)Q; w.out = out; w.zero = (const half_t*)zero16;
        w.N = N; w.Yc = Yc; w.Xc = Xc; w.A = A; w.Hq = Hq; w.Wq = Wq; w.Bc = Bc; w.pad = pad; w.TW = k; w.ldo = ldo;
        w.slab_stride = atomic == 2 ? (int64_t)apad * ldo : 0;
        w.a_tiles = apad / 128;
        bool ok = true;
        for (int par = 0; par < 2; ++par) {
            int cnt = 0, emin = 0;
            for (int t = 0; t < k; ++t)
                if (((t - pad) & 1) == par) { if (!cnt) emin = t - pad; ++cnt; }
            w.nsy[par] = w.nsx[par] = cnt;
            w.tmin[par] = (emin - par) / 2;      // exact: emin and par have the same parity
            if (cnt < 2 || cnt > 3) ok = false;
        }
        if (ok) {
            w.tiles_y = (Yc + 7) / 8; w.tiles_x = (Xc + 7) / 8;
            w.ntiles = N * w.tiles_y * w.tiles_x;
            // `splits` = blocks per (row block, column block) over the 4 planes; K pieces per plane: wgrad_plane_pieces()
            int tps4[4];
            const int smax = wgrad_plane_pieces(N, Yc, Xc, k, pad, splits, tps4);      // = fmri_wgrad_slabs(): in slab mode the
            for (int pl = 0; pl < 4; ++pl) {                                          // caller sized the output with it, and
                w.plane_tps[pl] = tps4[pl];                                           // every allocated slab is written
                w.plane_pieces[pl] = (w.ntiles + tps4[pl] - 1) / tps4[pl];
            }
            w.splits = smax;
            w.fdTPI = make_fastdiv((uint32_t)(w.tiles_y * w.tiles_x));
            w.fdTX = make_fastdiv((uint32_t)w.tiles_x);
            return wgrad_win_launch(w, apad, S(stream));
        }
    }
    // 5x5 stride-1 layers between 32 and 3(8) channels, pre-zeroed output: wave-private window kernel
    // (csrc/wgrad_narrow.hip).  FMRI_WGRAD_NARROW=off disables.
    static const char* wn_env = getenv("FMRI_WGRAD_NARROW");
    static const bool no_wn = wn_env && !strcmp(wn_env, "off");
    if (atomic == 3 && (no_wn || !(stride == 1 && k == 5 && pad == 2 && A == 32 && Bc == 8 && apad == 32 && Yc == Hq &&
                           Xc == Wq)))
        return FMRI_E_UNSUPPORTED;
    if (atomic == 3 && stride == 1 && k == 5 && pad == 2 && A == 32 && Bc == 8 && apad == 32 && Yc == Hq &&
        Xc == Wq && (int64_t)N * Yc * Xc * 32 < 0x7fffffffLL) {
        WgradNarrowArgs w;
        w.gate = gate;
        w.P = (const half_t*)P; w.Q = (const half_t*)Q; w.out = out; w.zero = (const half_t*)zero16;
        w.N = N; w.H = Yc; w.W = Xc; w.ldo = ldo; w.flip = flip;
        w.tiles_y = (Yc + 7) / 8; w.tiles_x = (Xc + 7) / 8;
        w.ntiles = N * w.tiles_y * w.tiles_x;
        const int nb = wgrad_narrow_blocks(N, Yc, Xc);
        w.nslabs = splits < 1 ? 1 : splits;         // the caller allocated `splits` zeroed slabs of apad x ldo
        w.pad0 = 0;
        w.slab_stride = (int64_t)apad * ldo;
        return wgrad_narrow_launch(w, nb, S(stream));
    }
    if (atomic == 2) return FMRI_E_UNSUPPORTED;      // plane-piece slabs exist only in the window-resident kernel
    WgradArgs a;
    a.gate = gate;
    a.P = (const half_t*)P; a.Q = (const half_t*)Q; a.out = out; a.zero = (const half_t*)zero16;
    a.N = N; a.Yc = Yc; a.Xc = Xc; a.A = A; a.Hq = Hq; a.Wq = Wq; a.Bc = Bc;
    a.s = stride; a.T = T; a.TW = k;
    a.dy0 = flip ? pad : -pad; a.dx0 = a.dy0; a.dstep = flip ? -1 : 1;
    a.M = (int)M; a.ldo = ldo;
    const int steps = (int)((M + 63) / 64);
    if (splits > steps) splits = steps;
    a.steps_per_split = (steps + splits - 1) / splits;
    a.splits = (steps + a.steps_per_split - 1) / a.steps_per_split;
    // atomic == 4: per-split slabs of the generic kernel -- the caller allocated `splits` (as passed in) slabs of
    // apad x ldo; the kernel writes every element of the first a.splits (<= splits) of them with plain stores, the
    // rest stay as the caller left them (zero-filled)
    a.atomic = atomic == 4 ? 2 : atomic;
    a.slab_stride = (int64_t)apad * ldo;
    a.ncol_chunks = T * Bc / 8;
    a.fdX = make_fastdiv((uint32_t)Xc);
    a.fdYX = make_fastdiv((uint32_t)(Yc * Xc));
    a.fdTW = make_fastdiv((uint32_t)k);
    a.fdBc8 = make_fastdiv((uint32_t)(Bc / 8));
    return wgrad_launch(a, apad, ba_tile, S(stream));
}

int fmri_nchw_to_nhwc(const float* src, void* dst, int N, int C, int HW, int Cp, void* stream) {
    if (!src || !dst || (Cp & 7) || C > Cp) return FMRI_E_BADARG;
    return nchw_to_nhwc_launch(src, (half_t*)dst, N, C, HW, Cp, S(stream));
}
int fmri_nhwc_to_nchw(const void* src, float* dst, int N, int C, int HW, int Cp, float scale, void* stream) {
    if (!src || !dst || C > Cp) return FMRI_E_BADARG;
    return nhwc_to_nchw_launch((const half_t*)src, dst, N, C, HW, Cp, scale, S(stream));
}
int fmri_rows_f32_to_f16(const float* src, void* dst, int M, int C, int Cp, float scale, void* stream) {
    if (!src || !dst || C > Cp) return FMRI_E_BADARG;
    return rows_f32_to_f16_launch(src, (half_t*)dst, M, C, Cp, scale, S(stream));
}
int fmri_rows_f16_to_f32(const void* src, float* dst, int M, int C, int Cp, float scale, void* stream) {
    if (!src || !dst || C > Cp) return FMRI_E_BADARG;
    return rows_f16_to_f32_launch((const half_t*)src, dst, M, C, Cp, scale, S(stream));
}
int fmri_reduce_slabs(const float* slabs, int nslabs, int64_t slab_stride, int M, int C, int ld, const float* bias,
                      int act, float* out32, int ld32, void* out16, int ld16, void* stream) {
    if (!slabs || nslabs < 1 || C > ld) return FMRI_E_BADARG;
    return reduce_slabs_launch(slabs, nslabs, slab_stride, M, C, ld, bias, act, out32, ld32, (half_t*)out16, ld16,
                               S(stream));
}
int fmri_permute_chw(const float* src, float* dst, int C, int HW, int to_engine, float scale, int accumulate,
                     void* stream) {
    if (!src || !dst) return FMRI_E_BADARG;
    return permute_chw_launch(src, dst, C, HW, to_engine, scale, accumulate, S(stream));
}

int64_t fmri_bn_ws_floats(int M, int C) { return (M < 1 || C < 8) ? 0 : bn_ws_floats(M, C); }

int fmri_bn_stats(const void* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, void* stream) {
    if (!x || !sums2C || (C & 7) || M < 1) return FMRI_E_BADARG;
    return bn_stats_launch((const half_t*)x, M, C, sums2C, ws, ws_floats, S(stream));
}
int fmri_bn_finalize(const float* sums2C, int C, float count, const float* gamma, const float* beta, float eps,
                     float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, int64_t* num_batches_tracked, void* stream) {
    if (!sums2C || !gamma || !beta || !mean || !rstd || !scale || !shift) return FMRI_E_BADARG;
    return bn_finalize_launch(sums2C, C, count, gamma, beta, eps, momentum, updates, running_mean, running_var, mean,
                              rstd, scale, shift, (long long*)num_batches_tracked, nullptr, S(stream));
}
int fmri_bn_finalize_s(const float* sums2C, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                       float* scale, float* shift, int64_t* num_batches_tracked, const float* in_scale, void* stream) {
    if (!sums2C || !gamma || !beta || !mean || !rstd || !scale || !shift) return FMRI_E_BADARG;
    return bn_finalize_launch(sums2C, C, count, gamma, beta, eps, momentum, updates, running_mean, running_var, mean,
                              rstd, scale, shift, (long long*)num_batches_tracked, in_scale, S(stream));
}
int fmri_bn_stats_finalize(const void* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float count,
                           const float* gamma, const float* beta, float eps, float momentum, int updates,
                           float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                           float* shift, int64_t* num_batches_tracked, void* stream) {
    if (!x || !sums2C || (C & 7) || M < 1 || !gamma || !beta || !mean || !rstd || !scale || !shift)
        return FMRI_E_BADARG;
    return bn_stats_finalize_launch((const half_t*)x, M, C, sums2C, ws, ws_floats, count, gamma, beta, eps, momentum,
                                    updates, running_mean, running_var, mean, rstd, scale, shift,
                                    (long long*)num_batches_tracked, S(stream));
}
int fmri_bn_cols_fwd(const void* x, void* y, int M, int C, float count, const float* gamma, const float* beta, float eps,
                     float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, float* sums2C, int64_t* num_batches_tracked, int relu, void* stream) {
    if (!x || !y || !sums2C || (C & 7) || C < 8 || M < 1 || !gamma || !beta || !mean || !rstd || !scale || !shift)
        return FMRI_E_BADARG;
    return bn_cols_fwd_launch((const half_t*)x, (half_t*)y, M, C, count, gamma, beta, eps, momentum, updates, running_mean,
                              running_var, mean, rstd, scale, shift, sums2C, (long long*)num_batches_tracked, relu,
                              nullptr, S(stream));
}
int fmri_bn_cols_fwd_s(const void* x, void* y, int M, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                       float* scale, float* shift, float* sums2C, int64_t* num_batches_tracked, int relu,
                       const float* in_scale, void* stream) {
    if (!x || !y || !sums2C || (C & 7) || C < 8 || M < 1 || !gamma || !beta || !mean || !rstd || !scale || !shift)
        return FMRI_E_BADARG;
    return bn_cols_fwd_launch((const half_t*)x, (half_t*)y, M, C, count, gamma, beta, eps, momentum, updates, running_mean,
                              running_var, mean, rstd, scale, shift, sums2C, (long long*)num_batches_tracked, relu,
                              in_scale, S(stream));
}
int fmri_bn_cols_bwd(const void* x, const void* dy, void* dx, int M, int C, int nstreams, float count, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int relu, float* sums, float* dbeta,
                     float* dgamma, float gscale, int param_stream, void* stream) {
    if (!x || !dy || !dx || !sums || (C & 7) || C < 8 || M < 1 || !mean || !rstd || !gamma || !beta || count <= 0.f ||
        param_stream < 0 || param_stream >= nstreams)
        return FMRI_E_BADARG;
    return bn_cols_bwd_launch((const half_t*)x, (const half_t*)dy, (half_t*)dx, M, C, nstreams, count, mean, rstd, gamma,
                              beta, relu, sums, dbeta, dgamma, gscale, param_stream, S(stream));
}
int fmri_bn_fold_finalize(const float* stat_part, int rows, int C, float* scratch, float* sums2C, float count,
                          const float* gamma, const float* beta, float eps, float momentum, int updates,
                          float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                          int64_t* num_batches_tracked, void* stream) {
    if (!stat_part || rows < 1 || C < 1 || !scratch || !sums2C || !gamma || !beta || !mean || !rstd || !scale || !shift)
        return FMRI_E_BADARG;
    return bn_fold_finalize_launch(stat_part, rows, C, scratch, sums2C, count, gamma, beta, eps, momentum, updates,
                                   running_mean, running_var, mean, rstd, scale, shift, (long long*)num_batches_tracked,
                                   S(stream));
}
int fmri_bn_fold(const float* stat_part, int rows, int C, float* scratch, float* sums2C, void* stream) {
    if (!stat_part || rows < 1 || C < 1 || !scratch || !sums2C) return FMRI_E_BADARG;
    return bn_fold_launch(stat_part, rows, 2 * C, scratch, sums2C, S(stream));
}
int fmri_bn_fold_scratch_floats(int C) { return C < 1 ? 0 : FOLD_STAGE_ROWS * 2 * C; }
int fmri_bn_bwd_fold(const float* stat_part, int rows, int rows_cap, int C, int groups, float* scratch, float* sums,
                     float* dbeta, float* dgamma, float gscale, int param_group, void* stream) {
    if (!stat_part || rows < 1 || rows > rows_cap || C < 1 || groups < 1 || groups > 4 || !scratch || !sums)
        return FMRI_E_BADARG;
    return bn_bwd_fold_launch(stat_part, rows, rows_cap, C, groups, scratch, sums, dbeta, dgamma, gscale, param_group,
                              S(stream));
}
int fmri_bn_apply(const void* x, void* y, int M, int C, const float* scale, const float* shift, int relu,
                  void* stream) {
    if (!x || !y || (C & 7)) return FMRI_E_BADARG;
    return bn_apply_launch((const half_t*)x, (half_t*)y, M, C, scale, shift, relu, S(stream));
}
int fmri_bn_bwd_reduce(const void* x, const void* dy, int M, int C, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int relu, float* sums2C, float* ws,
                       int64_t ws_floats, float* dbeta, float* dgamma, float gscale, void* stream) {
    if (!x || !dy || !sums2C || (C & 7)) return FMRI_E_BADARG;
    return bn_bwd_reduce_launch((const half_t*)x, (const half_t*)dy, M, C, mean, rstd, gamma, beta, relu, sums2C, ws,
                                ws_floats, dbeta, dgamma, gscale, S(stream));
}
int fmri_bn_bwd_reduce2(const void* x, const void* dy2, int M, int C, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int relu, float* sums4C, float* ws, int64_t ws_floats,
                        float* dbeta, float* dgamma, float gscale, int param_stream, void* stream) {
    if (!x || !dy2 || !sums4C || (C & 7) || M < 1 || (param_stream & ~1)) return FMRI_E_BADARG;
    return bn_bwd_reduce2_launch((const half_t*)x, (const half_t*)dy2, M, C, mean, rstd, gamma, beta, relu, sums4C, ws,
                                 ws_floats, dbeta, dgamma, gscale, param_stream, S(stream));
}
int fmri_bn_bwd_apply2(const void* x, const void* dy2, void* dx2, int M, int C, float count, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, int relu, const float* sums4C,
                       void* stream) {
    if (!x || !dy2 || !dx2 || !sums4C || (C & 7) || M < 1) return FMRI_E_BADARG;
    return bn_bwd_apply2_launch((const half_t*)x, (const half_t*)dy2, (half_t*)dx2, M, C, count, mean, rstd, gamma, beta,
                                relu, sums4C, S(stream));
}
int fmri_bn_bwd_apply(const void* x, const void* dy, void* dx, int M, int C, float count, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, const float* sums2C,
                      void* stream) {
    if (!x || !dy || !dx || (C & 7)) return FMRI_E_BADARG;
    return bn_bwd_apply_launch((const half_t*)x, (const half_t*)dy, (half_t*)dx, M, C, count, mean, rstd, gamma, beta,
                               relu, sums2C, S(stream));
}
int fmri_act_bwd(const void* y, const void* dy, void* dpre, int M, int C, int act, float* colsum2C, float* ws,
                 int64_t ws_floats, float* dbias, int dbias_n, float gscale, void* stream) {
    if (!y || !dy || !dpre || (C & 7) || (dbias && (!colsum2C || dbias_n < 1 || dbias_n > C))) return FMRI_E_BADARG;
    return act_bwd_launch((const half_t*)y, (const half_t*)dy, (half_t*)dpre, M, C, act, colsum2C, ws, ws_floats, dbias,
                          dbias_n, gscale, S(stream));
}
int fmri_colsum_rows(const void* x16, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float* dbias,
                     int dbias_n, float gscale, void* stream) {
    if (!x16 || !sums2C || !ws || M < 1 || C < 8 || (C & 7) || (dbias && (dbias_n < 1 || dbias_n > C))) return FMRI_E_BADARG;
    return colsum_rows_launch((const half_t*)x16, M, C, sums2C, ws, ws_floats, dbias, dbias_n, gscale, S(stream));
}
int fmri_colsum_acc(const void* src, int is_f16, int M, int C, int64_t ld_row, int64_t ld_col, float scale, float* dst,
                    void* stream) {
    if (!src || !dst || M < 1 || C < 1) return FMRI_E_BADARG;
    return colsum_acc_launch(src, is_f16, M, C, ld_row, ld_col, scale, dst, S(stream));
}

int fmri_latent_fwd(const float* head, const float* eps, int B, int Z, int zp, void* z16, float* kl_rows,
                    float* kl_total, int sample, void* stream) {
    if (!head || !z16 || (sample && !eps) || zp < Z) return FMRI_E_BADARG;
    return latent_fwd_launch(head, eps, B, Z, zp, (half_t*)z16, kl_rows, kl_total, sample, S(stream));
}
int fmri_latent_fwd_ranged(const float* head, const float* eps, int B, int Z, int zp, void* z16, float* kl_rows,
                           float* kl_total, int sample, float* z32, float* zmax, float* zscale, float cap, int phase,
                           void* stream) {
    if (!z32 || !zmax || phase < 1 || phase > 3 || zp < Z || B < 1 || Z < 1) return FMRI_E_BADARG;
    if ((phase & 1) && (!head || (sample && !eps))) return FMRI_E_BADARG;
    if ((phase & 2) && (!z16 || !zscale || !(cap > 0.f))) return FMRI_E_BADARG;
    return latent_ranged_launch(head, eps, B, Z, zp, (half_t*)z16, kl_rows, kl_total, sample, z32, zmax, zscale, cap, phase,
                                S(stream));
}
float fmri_latent_range_scale(float zmax, float cap) { return latent_range_scale_host(zmax, cap); }
int fmri_rows_absmax(const float* x, int64_t n, float* zmax, void* stream) {
    if (!x || !zmax || n < 1) return FMRI_E_BADARG;
    return rows_absmax_launch(x, n, zmax, S(stream));
}
int fmri_latent_bwd(const float* head, const float* eps, const float* dz, int ldz, float dz_unscale, float kl_w,
                    const float* kl_dev, int B, int Z, float out_scale, void* dhead16, float* dhead32, int sample,
                    void* stream) {
    if (!head || (sample && !eps)) return FMRI_E_BADARG;
    return latent_bwd_launch(head, eps, dz, ldz, dz_unscale, kl_w, kl_dev, B, Z, out_scale, (half_t*)dhead16, dhead32,
                             sample, S(stream));
}
int fmri_feat_mse(const void* feat, int B, int F, float* mse_rows, float* mse_total, void* stream) {
    if (!feat || (F & 7)) return FMRI_E_BADARG;
    return feat_mse_launch((const half_t*)feat, B, F, mse_rows, mse_total, S(stream));
}
int fmri_feat_mse_bwd(const void* feat, int B, int F, void* dfeat, float gscale, const float* norm, void* stream) {
    if (!feat || !dfeat || (F & 7)) return FMRI_E_BADARG;
    return feat_mse_bwd_launch((const half_t*)feat, B, F, (half_t*)dfeat, gscale, norm, S(stream));
}
int fmri_pixel_sq(const void* x, const void* xt, int64_t npix, int C, int Cp, float* total, void* dxt, float gscale,
                  void* stream) {
    if (!x || !xt) return FMRI_E_BADARG;
    return pixel_sq_launch((const half_t*)x, (const half_t*)xt, npix, C, Cp, total, (half_t*)dxt, gscale, S(stream));
}
int fmri_gan_head(const float* logit, int ldl, int B, float* prob, float* scal, void* stream) {
    if (!logit || !scal) return FMRI_E_BADARG;
    return gan_head_launch(logit, ldl, B, prob, scal, 7, S(stream));
}
int fmri_gan_head_bwd(const float* logit, int ldl, int B, void* dlogit, int ldg, float gscale, const float* norm,
                      void* stream) {
    if (!logit || !dlogit) return FMRI_E_BADARG;
    return gan_head_bwd_launch(logit, ldl, B, (half_t*)dlogit, ldg, gscale, norm, 7, S(stream));
}
int fmri_gan_head_parts(const float* logit, int ldl, int B, float* prob, float* scal, int parts, void* stream) {
    if (!logit || !scal || parts < 0 || parts > 7) return FMRI_E_BADARG;
    return gan_head_launch(logit, ldl, B, prob, scal, parts, S(stream));
}
int fmri_gan_head_bwd_parts(const float* logit, int ldl, int B, void* dlogit, int ldg, float gscale, const float* norm,
                            int parts, void* stream) {
    if (!logit || !dlogit || parts < 0 || parts > 7) return FMRI_E_BADARG;
    return gan_head_bwd_launch(logit, ldl, B, (half_t*)dlogit, ldg, gscale, norm, parts, S(stream));
}
int fmri_wae_logloss(const float* logit, int ldl, int n, int one_minus, float w, float* total, float* prob,
                     void* dlogit, int ldg, float gscale, void* stream) {
    if (!logit) return FMRI_E_BADARG;
    return wae_logloss_launch(logit, ldl, n, one_minus, w, total, prob, (half_t*)dlogit, ldg, gscale, S(stream));
}
int fmri_mlp_fwd(const void* z16, int M, int Zp, int H, const void* const* w5, const int* kp5, const float* const* bias5,
                 void* const* hs4, float* logit, void* stream) {
    if (!z16 || !w5 || !kp5 || !bias5 || !hs4 || !logit || M < 1) return FMRI_E_BADARG;
    MlpFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.z = (const half_t*)z16; a.M = M; a.Zp = Zp; a.H = H; a.logit = logit;
    for (int i = 0; i < 5; ++i) {
        if (!w5[i] || kp5[i] < (i == 0 ? Zp : H)) return FMRI_E_BADARG;
        a.w[i] = (const half_t*)w5[i]; a.kp[i] = kp5[i]; a.bias[i] = bias5[i];
    }
    for (int i = 0; i < 4; ++i) {
        if (!hs4[i]) return FMRI_E_BADARG;
        a.hs[i] = (half_t*)hs4[i];
    }
    return mlp_fwd_launch(a, S(stream));
}
int fmri_mlp_bwd(const void* dlogit16, int ldl, int M, int Zp, int Z, int H, const void* const* hs4, const void* w4row,
                 const void* const* wd4, const int* kpd4, void* const* delta4, float* const* dbias5, float* dz32,
                 float inv_scale, void* stream) {
    if (!dlogit16 || !hs4 || !w4row || !wd4 || !kpd4 || !delta4 || M < 1 || ldl < 1 || Z > Zp) return FMRI_E_BADARG;
    MlpBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dlogit = (const half_t*)dlogit16; a.ldl = ldl; a.M = M; a.Zp = Zp; a.H = H; a.Z = Z;
    a.w4 = (const half_t*)w4row; a.dz = dz32; a.inv_scale = inv_scale;
    for (int i = 0; i < 4; ++i) {
        if (!hs4[i] || !delta4[i]) return FMRI_E_BADARG;
        if ((i > 0 || dz32) && (!wd4[i] || kpd4[i] < H)) return FMRI_E_BADARG;
        a.hs[i] = (const half_t*)hs4[i]; a.delta[i] = (half_t*)delta4[i];
        a.wd[i] = (const half_t*)wd4[i]; a.kpd[i] = kpd4[i];
    }
    for (int i = 0; i < 5; ++i) a.dbias[i] = dbias5 ? dbias5[i] : nullptr;
    return mlp_bwd_launch(a, S(stream));
}
int fmri_compose_gate(float* scal, int* flags, float batch, float nfeat, float lambda_mse, float equilibrium,
                      float margin, int gate_on, int force_dis, int force_dec, void* stream) {
    if (!scal || !flags) return FMRI_E_BADARG;
    return compose_gate_launch(scal, flags, batch, nfeat, 0.f, lambda_mse, equilibrium, margin, 1.f, nullptr, 0, gate_on,
                               force_dis, force_dec, S(stream));
}
int fmri_compose_gate_dev(float* scal, int* flags, float batch, float nfeat, float npix, const float* hp4_dev, int mode,
                          int gate_on, int force_dis, int force_dec, void* stream) {
    if (!scal || !flags || !hp4_dev || mode < 0 || mode > 3) return FMRI_E_BADARG;
    return compose_gate_launch(scal, flags, batch, nfeat, npix, 0.f, 0.f, 0.f, 0.f, hp4_dev, mode, gate_on, force_dis,
                               force_dec, S(stream));
}
int fmri_counter_inc(int* counter_dev, void* stream) {
    if (!counter_dev) return FMRI_E_BADARG;
    return counter_inc_launch(counter_dev, S(stream));
}
int fmri_axpby_f16(const void* x, const void* y, void* out, int64_t n, float a, float b, const float* a_dev,
                   void* stream) {
    if (!x || !out || (n & 7)) return FMRI_E_BADARG;
    return axpby_f16_launch((const half_t*)x, (const half_t*)y, (half_t*)out, n, a, b, a_dev, nullptr, S(stream));
}
int fmri_axpby2_f16(const void* x, const void* y, void* out, int64_t n, float a, float b, const float* a_dev,
                    const float* b_dev, void* stream) {
    if (!x || !out || (n & 7)) return FMRI_E_BADARG;
    return axpby_f16_launch((const half_t*)x, (const half_t*)y, (half_t*)out, n, a, b, a_dev, b_dev, S(stream));
}
int fmri_sumsq(const float* x, int64_t n, float* acc, void* stream) {
    if (!x || !acc) return FMRI_E_BADARG;
    return sumsq_launch(x, n, acc, S(stream));
}
int fmri_renorm(const float* x, void* out16, int64_t n, float scale, const float* sumsq, float count,
                const float* factor_in, float* factor_out, void* stream) {
    if (!x || !out16 || !sumsq || count <= 0.f) return FMRI_E_BADARG;
    return renorm_launch(x, (half_t*)out16, n, scale, sumsq, count, factor_in, factor_out, S(stream));
}
int fmri_sumsq_f64(const float* x, int64_t n, double* acc, int zero_first, void* stream) {
    if (!x || !acc || ((uintptr_t)acc & 7)) return FMRI_E_BADARG;
    return sumsq64_launch(x, n, acc, zero_first, S(stream));
}
int fmri_renorm_f64(const float* x, void* out16, int64_t n, float scale, const double* sumsq, float count,
                    const float* factor_in, float* factor_out, void* stream) {
    if (!x || !out16 || !sumsq || ((uintptr_t)sumsq & 7) || count <= 0.f) return FMRI_E_BADARG;
    return renorm64_launch(x, (half_t*)out16, n, scale, sumsq, count, factor_in, factor_out, S(stream));
}
int fmri_rmsprop(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float gscale,
                 const float* gdev, float clamp, const int* flag, void* stream) {
    if (!p || !g || !sq) return FMRI_E_BADARG;
    return rmsprop_launch(p, g, sq, n, lr, alpha, eps, gscale, gdev, clamp, flag, nullptr, S(stream));
}
int fmri_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
              float bc1, float bc2_sqrt, float gscale, const float* gdev, float clamp, const int* flag,
              void* stream) {
    if (!p || !g || !m || !v) return FMRI_E_BADARG;
    return adam_launch(p, g, m, v, n, lr, b1, b2, eps, bc1, bc2_sqrt, gscale, gdev, clamp, flag, nullptr, nullptr,
                       S(stream));
}
int fmri_rmsprop_dev(float* p, const float* g, float* sq, int64_t n, const float* lr_dev, float alpha, float eps,
                     float gscale, const float* gdev, float clamp, const int* flag, void* stream) {
    if (!p || !g || !sq || !lr_dev) return FMRI_E_BADARG;
    return rmsprop_launch(p, g, sq, n, 0.f, alpha, eps, gscale, gdev, clamp, flag, lr_dev, S(stream));
}
int fmri_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float b1, float b2,
                  float eps, const int* t_dev, float gscale, const float* gdev, float clamp, const int* flag,
                  void* stream) {
    if (!p || !g || !m || !v || !lr_dev || !t_dev) return FMRI_E_BADARG;
    return adam_launch(p, g, m, v, n, 0.f, b1, b2, eps, 1.f, 1.f, gscale, gdev, clamp, flag, lr_dev, t_dev, S(stream));
}

}  // extern "C"
