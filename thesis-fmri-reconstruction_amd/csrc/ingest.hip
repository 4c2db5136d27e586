// Device-side tail of the reference's image pipeline: a batch of already cropped/resized uint8 HWC
// images -> the engine's fp16 NHWC(8) input (and optionally the fp32 NCHW tensor the module API takes), fusing
//   RandomHorizontalFlip (train_vgan_stage1.py:166)  /  RandomShift = scipy.ndimage.shift(order=0, mode='nearest')
//   by whole pixels (data_preprocessing/data_loader.py:186-217)  ->  ToTensor (u8 / 255)  ->  GreyToColor
//   (data_loader.py:374-401: 1 -> 3 channels)  ->  Normalize(mean, std) (train_vgan_stage1.py:169).
// The random draws (flip flag, shift) are made by the caller and passed per image, so the kernel is deterministic.
// One 16-byte store per pixel; HBM bound and tiny (256 images of 64x64: 3 MB in, 8 MB out).
#include "kernels.h"

namespace fmri {

__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ src, int N, int H, int W, int C,
                                                        const int* __restrict__ flip, const int* __restrict__ shift,
                                                        float m0, float m1, float m2, float r0, float r1, float r2,
                                                        half_t* __restrict__ dst16, float* __restrict__ dst32) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int n = (int)(i / ((int64_t)W * H));
        // output pixel (y, x) of the shifted image = input pixel (y - sy, x - sx), clamped to the edge ('nearest');
        // the flip is applied before the shift (transform order of the scripts)
        int sy = 0, sx = 0;
        if (shift) { sy = shift[2 * n]; sx = shift[2 * n + 1]; }
        int yy = y - sy, xx = x - sx;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        if (flip && flip[n]) xx = W - 1 - xx;
        const uint8_t* p = src + ((int64_t)(n * H + yy) * W + xx) * C;
        const float v0 = p[0] * (1.f / 255.f);
        const float v1 = C == 3 ? p[1] * (1.f / 255.f) : v0;
        const float v2 = C == 3 ? p[2] * (1.f / 255.f) : v0;
        const float o0 = (v0 - m0) * r0, o1 = (v1 - m1) * r1, o2 = (v2 - m2) * r2;
        if (dst16) {
            h8 o = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f,
                    (half_t)0.f};
            *(h8*)(dst16 + i * 8) = o;
        }
        if (dst32) {
            const int64_t hw = (int64_t)H * W, b = (int64_t)n * 3 * hw + (int64_t)y * W + x;
            dst32[b] = o0; dst32[b + hw] = o1; dst32[b + 2 * hw] = o2;
        }
    }
}

int ingest_u8_launch(const uint8_t* src, int N, int H, int W, int C, const int* flip, const int* shift,
                     const float* mean3, const float* std3, half_t* dst16, float* dst32, hipStream_t st) {
    const int64_t total = (int64_t)N * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3(blocks), dim3(256), 0, st, src, N, H, W, C, flip, shift, mean3[0],
                       mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2], dst16, dst32);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}


// ----------------------------------------------------------------------------------------------------------------
// Device-side HEAD of the pipeline (round 3): CenterCrop((crop, crop)) + Resize((S, S)) of a ragged batch of decoded
// uint8 images (train_vgan_stage1.py:162-165), bit-exact with what torchvision 0.5.0 / Pillow 8.0.1 compute:
//   * crop box = torchvision's center_crop (round-half-even of (h - crop) / 2), zeros outside the image (PIL crop);
//   * Pillow's ImagingResample, 8 bits per channel: separable antialiased triangle filter, 22-bit fixed-point
//     coefficients (computed on the host in double precision, resize_coeffs below = precompute_coeffs +
//     normalize_coeffs_8bpc), HORIZONTAL pass first, its result rounded and clipped to uint8, then the vertical pass; a
//     pass whose input and output size agree is the identity.
// One block per (image, output row): the <= ksize_v horizontally resampled input rows the output row needs are built in
// LDS (uint8, as Pillow's intermediate image), then combined.  Grey images leave as three equal channels (GreyToColor,
// data_loader.py:374-401, commutes with everything that follows).  Byte work, HBM-bound: 256 COCO-sized images
// (~110 MB in, 3 MB out) take tens of microseconds; the crop of wider images is never read.
// ----------------------------------------------------------------------------------------------------------------
namespace {
constexpr int RS_PREC = 32 - 8 - 2;
__device__ __forceinline__ uint8_t rs_clip8(int v) {
    v >>= RS_PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
}  // namespace

// hb / vb: [S][2] (first input index, count) of the horizontal / vertical pass, hk / vk: [S][hks / vks] coefficients;
// hks == 0 / vks == 0: that pass is the identity (crop == S).  dims[n] = (H, W, C).
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const uint8_t* __restrict__ pool,
                                                             const int64_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ dims, int crop, int S,
                                                             const int32_t* __restrict__ hb, const int32_t* __restrict__ hk,
                                                             int hks, const int32_t* __restrict__ vb,
                                                             const int32_t* __restrict__ vk, int vks,
                                                             uint8_t* __restrict__ out) {
    extern __shared__ uint8_t rows[];                 // [vcount][S][3] horizontally resampled rows
    const int n = blockIdx.y, oy = blockIdx.x;
    const int H = dims[3 * n], W = dims[3 * n + 1], C = dims[3 * n + 2];
    const uint8_t* img = pool + offsets[n];
    // torchvision 0.5.0 center_crop: int(round((h - th) / 2.)), Python 3 round = half to even
    const int dy = H - crop, dx = W - crop;
    auto rhe = [](int d) {
        if ((d & 1) == 0) return d / 2;
        const int k = (d - 1) / 2;                              // d odd: d - 1 even, the division is exact: k = floor(d / 2)
        return (k & 1) ? k + 1 : k;                             // d / 2 = k + 0.5 -> the even one of k, k + 1
    };
    const int top = rhe(dy), left = rhe(dx);
    const int vlo = vks ? vb[2 * oy] : oy, vcnt = vks ? vb[2 * oy + 1] : 1;
    // ---- horizontal pass of crop rows vlo .. vlo + vcnt - 1 into LDS
    for (int i = threadIdx.x; i < vcnt * S * 3; i += 256) {
        const int c = i % 3;
        const int ox = (i / 3) % S;
        const int r = i / (3 * S);
        const int iy = top + vlo + r;                           // image row of crop row vlo + r
        const int cs = C == 3 ? c : 0;
        int v = 0;
        if ((unsigned)iy < (unsigned)H) {
            const uint8_t* row = img + (int64_t)iy * W * C;
            if (hks) {
                const int lo = hb[2 * ox], cnt = hb[2 * ox + 1];
                int acc = 1 << (RS_PREC - 1);
                for (int t = 0; t < cnt; ++t) {
                    const int ix = left + lo + t;
                    const int p = (unsigned)ix < (unsigned)W ? row[ix * C + cs] : 0;
                    acc += p * hk[ox * hks + t];
                }
                v = rs_clip8(acc);
            } else {
                const int ix = left + ox;
                v = (unsigned)ix < (unsigned)W ? row[ix * C + cs] : 0;
            }
        } else if (hks) {
            v = rs_clip8(1 << (RS_PREC - 1));                   // a row of zeros resamples to clip8(rounding term) = 0
        }
        rows[i] = (uint8_t)v;
    }
    __syncthreads();
    // ---- vertical pass
    for (int i = threadIdx.x; i < S * 3; i += 256) {
        int v;
        if (vks) {
            int acc = 1 << (RS_PREC - 1);
            for (int t = 0; t < vcnt; ++t) acc += (int)rows[t * S * 3 + i] * vk[oy * vks + t];
            v = rs_clip8(acc);
        } else {
            v = rows[i];
        }
        out[((int64_t)n * S + oy) * S * 3 + i] = (uint8_t)v;
    }
}

// Pillow's precompute_coeffs (bilinear filter, support 1.0, box = whole input) + normalize_coeffs_8bpc.  Host function.
// Returns ksize (0: identity, in_size == out_size), or -1 if ksize exceeds ksize_cap.
int resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize_cap) {
    if (in_size == out_size) return 0;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize > ksize_cap) return -1;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double k[64];
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = ww != 0.0 ? k[x] / ww : k[x];
            coef[xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << RS_PREC)) : (int)(0.5 + v * (1 << RS_PREC));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

int crop_resize_u8_launch(const uint8_t* pool, const int64_t* offsets, const int32_t* dims, int N, int crop, int S,
                          const int32_t* hb, const int32_t* hk, int hks, const int32_t* vb, const int32_t* vk, int vks,
                          int vcount_max, uint8_t* out, hipStream_t st) {
    const int lds = vcount_max * S * 3;
    if (lds > 64 * 1024) return E_UNSUPPORTED;
    hipLaunchKernelGGL(crop_resize_u8_kernel, dim3(S, N), dim3(256), lds, st, pool, offsets, dims, crop, S, hb, hk, hks, vb,
                       vk, vks, out);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
