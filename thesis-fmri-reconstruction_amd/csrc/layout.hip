// Layout / packing kernels: reference fp32 layouts <-> engine fp16 NHWC / packed-GEMM layouts.
//
//  * pack_weight : S[a*sa + ta*sta + b*sb + t(tb)*stb] (fp32) -> D[(ta,a)][(tb,b)] (fp16, zero padded)
//                  with a tap selection t(tb) = (py + step*ty)*KW + (px + step*tx).  One kernel covers
//                  Conv2d / ConvTranspose2d / Linear weights in forward and data-gradient orientation,
//                  including the (C,H,W)->(H,W,C) permutation at the conv->fc flatten
//                  (reference models/vae_gan.py:89,127,181) and the 4 parity classes of the
//                  stride-2 transposed convolution.
//  * unpack_grad : the inverse map for fp32 weight gradients (packed -> reference layout, scaled).
//  * image / vector casts between NCHW fp32 and channel-padded NHWC fp16.
#include "kernels.h"

namespace fmri {

__global__ void pack_weight_kernel(const PackArgs p) {
    const int64_t total = (int64_t)p.rows_pad * p.kpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / p.kpad);
        const int k = (int)(i - (int64_t)row * p.kpad);
        float v = 0.f;
        const int ta = row / p.A, a = row - ta * p.A;
        const int tb = k / p.Bp, b = k - tb * p.Bp;
        if (ta < p.TA && tb < p.TH * p.TW && b < p.B) {
            const int ty = tb / p.TW, tx = tb - ty * p.TW;
            const int t = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
            v = p.src[a * p.sa + ta * p.sta + b * p.sb + t * p.stb];
        }
        p.dst[i] = (half_t)v;
    }
}

// Fast path 1 ("tap-inner"): source taps are contiguous (stb == 1).  One thread per (row, b): it reads its
// taps from one contiguous run of the source and, for every tap, the threads of a wave write consecutive b.
// Padding regions of dst are pre-zeroed by the caller and never touched.
__global__ void pack_tapinner_kernel(const PackArgs p) {
    const int64_t total = (int64_t)p.TA * p.A * p.B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % p.B);
        const int row = (int)(i / p.B);
        const int ta = row / p.A, a = row - ta * p.A;
        const float* s = p.src + a * p.sa + ta * p.sta + b * p.sb;
        half_t* d = p.dst + (int64_t)row * p.kpad + b;
        for (int ty = 0; ty < p.TH; ++ty)
            for (int tx = 0; tx < p.TW; ++tx) {
                const int t = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
                d[(ty * p.TW + tx) * p.Bp] = (half_t)s[t];
            }
    }
}

// Fast path 1b: LDS-tiled version of the tap-inner paths.  A block handles one row and 32 consecutive b:
// the fp32 side is touched as runs of `run` contiguous taps per (row, b) (one 32*run-float run when
// sb == run), the packed side as 32-element (64/128-byte) segments per tap.  run <= 64.
__global__ __launch_bounds__(256) void pack_tile_kernel(const PackArgs p, int run) {
    __shared__ float t[32 * 65];
    const int nbt = (p.B + 31) / 32;
    const int row = blockIdx.x / nbt, b0 = (blockIdx.x - row * nbt) * 32;
    const int ta = row / p.A, a = row - ta * p.A;
    const int stride = run | 1;
    const float* s = p.src + a * p.sa + ta * p.sta;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int bl = e / run, j = e - bl * run;
        if (b0 + bl < p.B) t[bl * stride + j] = s[(b0 + bl) * p.sb + j];
    }
    __syncthreads();
    half_t* d = p.dst + (int64_t)row * p.kpad + b0;
    const int ntaps = p.TH * p.TW;
    for (int e = threadIdx.x; e < ntaps * 32; e += 256) {
        const int tb = e >> 5, bl = e & 31;
        const int ty = tb / p.TW, tx = tb - ty * p.TW;
        const int j = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
        if (b0 + bl < p.B) d[tb * p.Bp + bl] = (half_t)t[bl * stride + j];
    }
}

// Batched form of pack_tile_kernel: one launch repacks every (weight, orientation, class) of a sub-network after its
// optimizer step.  `tab` lives in device memory and never changes (masters and packed buffers are persistent);
// entry e owns the blocks [tile_begin[e], tile_begin[e+1]).
__global__ __launch_bounds__(256) void pack_tile_batch_kernel(const PackEntry* __restrict__ tab, int n) {
    __shared__ float t[32 * 65];
    __shared__ int sel;
    // entry of this block: the table is sorted by tile_begin (tile_begin[0] = 0), so the entry index is the number of
    // entries that begin at or before this block, minus one.  One parallel round of loads by the first wave (round 3
    // walked the table serially in thread 0: up to n dependent global loads in front of 3 KB of work per block).
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int e = threadIdx.x; e < n; e += 64) cnt += (int)blockIdx.x >= tab[e].tile_begin ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (threadIdx.x == 0) sel = cnt - 1;
    }
    __syncthreads();
    const PackEntry& en = tab[sel];
    const PackArgs& p = en.p;
    const int run = en.run;
    const int tile = blockIdx.x - en.tile_begin;
    const int nbt = (p.B + 31) / 32;
    const int row = tile / nbt, b0 = (tile - row * nbt) * 32;
    const int ta = row / p.A, a = row - ta * p.A;
    const int stride = run | 1;
    const float* s = p.src + a * p.sa + ta * p.sta;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int bl = e / run, j = e - bl * run;
        if (b0 + bl < p.B) t[bl * stride + j] = s[(b0 + bl) * p.sb + j];
    }
    __syncthreads();
    half_t* d = p.dst + (int64_t)row * p.kpad + b0;
    const int ntaps = p.TH * p.TW;
    for (int e = threadIdx.x; e < ntaps * 32; e += 256) {
        const int tb = e >> 5, bl = e & 31;
        const int ty = tb / p.TW, tx = tb - ty * p.TW;
        const int j = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
        if (b0 + bl < p.B) d[tb * p.Bp + bl] = (half_t)t[bl * stride + j];
    }
}

// unpack counterpart (full tap set only: py = px = 0, step = 1, TH*TW == run)
__global__ __launch_bounds__(256) void unpack_tile_kernel(const UnpackArgs p, int run) {
    __shared__ float t[32 * 65];
    const int nbt = (p.B + 31) / 32;
    const int row = blockIdx.x / nbt, b0 = (blockIdx.x - row * nbt) * 32;
    const int ta = row / p.A, a = row - ta * p.A;
    const int stride = run | 1;
    const float* s = p.src + (int64_t)row * p.ld + b0;
    for (int e = threadIdx.x; e < run * 32; e += 256) {
        const int tb = e >> 5, bl = e & 31;
        if (b0 + bl < p.B) {
            // slab sum with four loads in flight (a one-at-a-time loop is a chain of memory latencies)
            const float* q = s + tb * p.Bp + bl;
            float v0 = q[0], v1 = 0.f, v2 = 0.f, v3 = 0.f;
            int z = 1;
            for (; z + 3 < p.nslabs; z += 4) {
                v0 += q[z * p.slab_stride];
                v1 += q[(z + 1) * p.slab_stride];
                v2 += q[(z + 2) * p.slab_stride];
                v3 += q[(z + 3) * p.slab_stride];
            }
            for (; z < p.nslabs; ++z) v0 += q[z * p.slab_stride];
            t[bl * stride + tb] = (v0 + v1) + (v2 + v3);
        }
    }
    __syncthreads();
    float* d = p.dst + a * p.sa + ta * p.sta;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int bl = e / run, j = e - bl * run;
        if (b0 + bl < p.B) {
            const float v = t[bl * stride + j] * p.scale;
            float* q = d + (b0 + bl) * p.sb + j;
            if (p.accumulate) *q += v; else *q = v;
        }
    }
}

// Fast path 2: LDS-tiled transpose between a source-contiguous index X (stride 1 in src, stride dX in dst)
// and b (stride sb in src, stride 1 in dst); an outer index o (O values) is iterated by blockIdx.z.
struct PackT { const float* src; half_t* dst; int X, B, O; int64_t sb, so_src, dX, so_dst; };
__global__ __launch_bounds__(256) void pack_transpose_kernel(const PackT p) {
    __shared__ float tile[32][33];
    const int x0 = blockIdx.x * 32, b0 = blockIdx.y * 32, o = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float* s = p.src + o * p.so_src;
    half_t* d = p.dst + o * p.so_dst;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = b0 + ty + 8 * k, x = x0 + tx;
        tile[ty + 8 * k][tx] = (b < p.B && x < p.X) ? s[b * p.sb + x] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + ty + 8 * k, b = b0 + tx;
        if (x < p.X && b < p.B) d[x * p.dX + b] = (half_t)tile[tx][ty + 8 * k];
    }
}

// unpack fast path (stb == 1): one thread per (row, b) writes its taps to one contiguous run of dst.
__global__ void unpack_tapinner_kernel(const UnpackArgs p) {
    const int64_t total = (int64_t)p.TA * p.A * p.B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % p.B);
        const int row = (int)(i / p.B);
        const int ta = row / p.A, a = row - ta * p.A;
        const float* s = p.src + (int64_t)row * p.ld + b;
        float* d = p.dst + a * p.sa + ta * p.sta + b * p.sb;
        for (int ty = 0; ty < p.TH; ++ty)
            for (int tx = 0; tx < p.TW; ++tx) {
                const int t = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
                float v = s[(ty * p.TW + tx) * p.Bp];
                for (int z = 1; z < p.nslabs; ++z) v += s[z * p.slab_stride + (ty * p.TW + tx) * p.Bp];
                v *= p.scale;
                if (p.accumulate) d[t] += v; else d[t] = v;
            }
    }
}

__global__ void unpack_grad_kernel(const UnpackArgs p) {
    const int ntb = p.TH * p.TW;
    const int64_t total = (int64_t)p.TA * p.A * ntb * p.B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // enumerate destination-friendly: b fastest is not contiguous in dst in general; keep simple
        int64_t r = i;
        const int b = (int)(r % p.B); r /= p.B;
        const int tb = (int)(r % ntb); r /= ntb;
        const int a = (int)(r % p.A); r /= p.A;
        const int ta = (int)r;
        const int ty = tb / p.TW, tx = tb - ty * p.TW;
        const int t = (p.py + p.step * ty) * p.KW + (p.px + p.step * tx);
        const float* sp = p.src + (int64_t)(ta * p.A + a) * p.ld + tb * p.Bp + b;
        float v = *sp;
        for (int z = 1; z < p.nslabs; ++z) v += sp[z * p.slab_stride];
        v *= p.scale;
        float* d = p.dst + a * p.sa + ta * p.sta + b * p.sb + t * p.stb;
        if (p.accumulate) *d += v; else *d = v;
    }
}

// NCHW fp32 -> NHWC fp16 with channel padding (zero fill).  One thread per (pixel, 8-channel chunk).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int N, int C, int HW,
                                    int Cp) {
    const int64_t total = (int64_t)N * HW * (Cp / 8);
    const int cch = Cp / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cch);
        const int64_t pix = i / cch;
        const int n = (int)(pix / HW);
        const int hw = (int)(pix - (int64_t)n * HW);
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            v[j] = c < C ? (half_t)src[((int64_t)n * C + c) * HW + hw] : (half_t)0.f;
        }
        *(h8*)(dst + pix * Cp + ch * 8) = v;
    }
}

// NHWC fp16 (channel stride Cp) -> NCHW fp32 (C channels), scaled.
__global__ void nhwc_to_nchw_kernel(const half_t* __restrict__ src, float* __restrict__ dst, int N, int C, int HW,
                                    int Cp, float scale) {
    const int64_t total = (int64_t)N * C * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const int64_t nc = i / HW;
        const int c = (int)(nc % C);
        const int n = (int)(nc / C);
        dst[i] = (float)src[((int64_t)n * HW + hw) * Cp + c] * scale;
    }
}

// rows fp32 [M][C] -> fp16 [M][Cp] (zero padded), scaled
__global__ void rows_f32_to_f16_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int M, int C, int Cp,
                                       float scale) {
    const int64_t total = (int64_t)M * Cp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const int64_t m = i / Cp;
        dst[i] = c < C ? (half_t)(src[m * C + c] * scale) : (half_t)0.f;
    }
}

// rows fp16 [M][Cp] -> fp32 [M][C], scaled
__global__ void rows_f16_to_f32_kernel(const half_t* __restrict__ src, float* __restrict__ dst, int M, int C, int Cp,
                                       float scale) {
    const int64_t total = (int64_t)M * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t m = i / C;
        dst[i] = (float)src[m * Cp + c] * scale;
    }
}

// sum of split-K slabs (+ optional bias, activation) -> fp32 and/or fp16 rows
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs, int64_t slab_stride, int M, int C,
                                    int ld, const float* __restrict__ bias, int act, float* __restrict__ out32,
                                    int ld32, half_t* __restrict__ out16, int ld16) {
    const int64_t total = (int64_t)M * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const int64_t m = i / ld;
        float v = 0.f;
        if (c < C) {
            float v1 = 0.f, v2 = 0.f, v3 = 0.f;
            int s = 0;
            for (; s + 3 < nslabs; s += 4) {       // four loads in flight
                v += slabs[s * slab_stride + i];
                v1 += slabs[(s + 1) * slab_stride + i];
                v2 += slabs[(s + 2) * slab_stride + i];
                v3 += slabs[(s + 3) * slab_stride + i];
            }
            for (; s < nslabs; ++s) v += slabs[s * slab_stride + i];
            v = (v + v1) + (v2 + v3);
            if (bias) v += bias[c];
            v = act_apply(v, act);
        }
        if (out32 && c < C) out32[m * ld32 + c] = v;
        if (out16 && c < ld16) out16[m * ld16 + c] = (half_t)v;
    }
}

// permute a per-feature vector between reference (C,HW) order and engine (HW,C) order
__global__ void permute_chw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                   int to_engine, float scale, int accumulate) {
    const int total = C * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / HW, hw = i - c * HW;   // i indexes reference order
        const int e = hw * C + c;                // engine order
        if (to_engine) {
            dst[e] = src[i] * scale;
        } else {
            if (accumulate) dst[i] += src[e] * scale; else dst[i] = src[e] * scale;
        }
    }
}

// dst[c] += scale * sum_m src[m*ld_row + c*ld_col], c < C (bias gradients: column sums of a cotangent; also the spare
// column of the narrow weight-gradient kernel's slabs).  One block per 8 columns, 32 row lanes, fixed-order sums.
template <typename T>
__global__ __launch_bounds__(256) void colsum_acc_kernel(const T* __restrict__ src, int M, int C, int64_t ld_row,
                                                         int64_t ld_col, float scale, float* __restrict__ dst) {
    __shared__ float red[32][9];
    const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    float s = 0.f;
    if (c < C)
        for (int m = ry; m < M; m += 32) s += (float)src[(int64_t)m * ld_row + (int64_t)c * ld_col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][cx];
        dst[c] += scale * t;
    }
}

static inline int nblocks(int64_t total, int threads = 256, int cap = 4096) {
    int64_t b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// The destination's padding (rows >= TA*A, columns >= taps*Bp, channels >= B) must be pre-zeroed by the
// caller once; the fast paths only write the valid region.
int pack_weight_launch(const PackArgs& p, hipStream_t st) {
    const int ntaps = p.TH * p.TW;
    const int run = (p.py + p.step * (p.TH - 1)) * p.KW + (p.px + p.step * (p.TW - 1)) + 1;   // source taps touched
    const int64_t tiles = (int64_t)p.TA * p.A * ((p.B + 31) / 32);
    if (ntaps > 1 && p.stb == 1 && run <= 64 && tiles < (1ll << 31)) {
        hipLaunchKernelGGL(pack_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, st, p, run);
    } else if (ntaps > 1 && p.stb == 1) {
        hipLaunchKernelGGL(pack_tapinner_kernel, dim3(nblocks((int64_t)p.TA * p.A * p.B, 256, 8192)), dim3(256), 0, st,
                           p);
    } else if (p.TA == 1 && p.sa == 1 && (ntaps == 1 || (p.step == 1 && p.py == 0 && p.px == 0 && p.TW == p.KW))) {
        // X = a (rows), outer = tap: dst[a][tb*Bp + b] = src[a + b*sb + tb*stb]
        PackT t{p.src, p.dst, p.A, p.B, ntaps, p.sb, p.stb, (int64_t)p.kpad, (int64_t)p.Bp};
        hipLaunchKernelGGL(pack_transpose_kernel, dim3((p.A + 31) / 32, (p.B + 31) / 32, ntaps), dim3(256), 0, st, t);
    } else if (ntaps == 1 && p.sta == 1 && p.TA > 1 && p.A <= 65535) {
        // X = ta, outer = a: dst[ta*A + a][b] = src[a*sa + ta + b*sb]
        PackT t{p.src, p.dst, p.TA, p.B, p.A, p.sb, p.sa, (int64_t)p.A * p.kpad, (int64_t)p.kpad};
        hipLaunchKernelGGL(pack_transpose_kernel, dim3((p.TA + 31) / 32, (p.B + 31) / 32, p.A), dim3(256), 0, st, t);
    } else {
        hipLaunchKernelGGL(pack_weight_kernel, dim3(nblocks((int64_t)p.rows_pad * p.kpad)), dim3(256), 0, st, p);
    }
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
// number of pack_tile blocks of `p` (and its `run`), or 0 when the weight needs one of the other pack paths
int pack_tile_count(const PackArgs& p, int* run_out) {
    const int ntaps = p.TH * p.TW;
    const int run = (p.py + p.step * (p.TH - 1)) * p.KW + (p.px + p.step * (p.TW - 1)) + 1;
    const int64_t tiles = (int64_t)p.TA * p.A * ((p.B + 31) / 32);
    if (run_out) *run_out = run;
    return (ntaps > 1 && p.stb == 1 && run <= 64 && tiles < (1 << 24)) ? (int)tiles : 0;
}
int pack_batch_launch(const PackEntry* tab, int n, int total_tiles, hipStream_t st) {
    if (total_tiles < 1 || n < 1) return OK;
    hipLaunchKernelGGL(pack_tile_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, tab, n);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int unpack_grad_launch(const UnpackArgs& p, hipStream_t st) {
    const int64_t total = (int64_t)p.TA * p.A * p.TH * p.TW * p.B;
    const int run = p.TH * p.TW;
    const int64_t tiles = (int64_t)p.TA * p.A * ((p.B + 31) / 32);
    if (run > 1 && run <= 64 && p.stb == 1 && p.py == 0 && p.px == 0 && p.step == 1 && p.TW == p.KW &&
        tiles < (1ll << 31)) {
        hipLaunchKernelGGL(unpack_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, st, p, run);
    } else if (p.TH * p.TW > 1 && p.stb == 1) {
        hipLaunchKernelGGL(unpack_tapinner_kernel, dim3(nblocks((int64_t)p.TA * p.A * p.B, 256, 8192)), dim3(256), 0, st,
                           p);
    } else {
        hipLaunchKernelGGL(unpack_grad_kernel, dim3(nblocks(total)), dim3(256), 0, st, p);
    }
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int nchw_to_nhwc_launch(const float* s, half_t* d, int N, int C, int HW, int Cp, hipStream_t st) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks((int64_t)N * HW * (Cp / 8))), dim3(256), 0, st, s, d, N, C,
                       HW, Cp);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int nhwc_to_nchw_launch(const half_t* s, float* d, int N, int C, int HW, int Cp, float scale, hipStream_t st) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks((int64_t)N * C * HW)), dim3(256), 0, st, s, d, N, C, HW, Cp,
                       scale);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int rows_f32_to_f16_launch(const float* s, half_t* d, int M, int C, int Cp, float scale, hipStream_t st) {
    hipLaunchKernelGGL(rows_f32_to_f16_kernel, dim3(nblocks((int64_t)M * Cp)), dim3(256), 0, st, s, d, M, C, Cp, scale);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int rows_f16_to_f32_launch(const half_t* s, float* d, int M, int C, int Cp, float scale, hipStream_t st) {
    hipLaunchKernelGGL(rows_f16_to_f32_kernel, dim3(nblocks((int64_t)M * C)), dim3(256), 0, st, s, d, M, C, Cp, scale);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int reduce_slabs_launch(const float* slabs, int nslabs, int64_t slab_stride, int M, int C, int ld, const float* bias,
                        int act, float* out32, int ld32, half_t* out16, int ld16, hipStream_t st) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nblocks((int64_t)M * ld)), dim3(256), 0, st, slabs, nslabs,
                       slab_stride, M, C, ld, bias, act, out32, ld32, out16, ld16);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int colsum_acc_launch(const void* src, int is_f16, int M, int C, int64_t ld_row, int64_t ld_col, float scale, float* dst,
                      hipStream_t st) {
    if (is_f16)
        hipLaunchKernelGGL((colsum_acc_kernel<half_t>), dim3((C + 7) / 8), dim3(256), 0, st, (const half_t*)src, M, C,
                           ld_row, ld_col, scale, dst);
    else
        hipLaunchKernelGGL((colsum_acc_kernel<float>), dim3((C + 7) / 8), dim3(256), 0, st, (const float*)src, M, C,
                           ld_row, ld_col, scale, dst);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int permute_chw_launch(const float* s, float* d, int C, int HW, int to_engine, float scale, int accumulate,
                       hipStream_t st) {
    hipLaunchKernelGGL(permute_chw_kernel, dim3(nblocks((int64_t)C * HW)), dim3(256), 0, st, s, d, C, HW, to_engine,
                       scale, accumulate);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}


// fp16 matrix transpose dst[c][r] = src[r][c] (r < R, c < C) in 64 x 64 tiles through LDS, 16-byte loads and stores.
// The two GEMM orientations of a dense layer's weight are transposes of each other ([n][k] for the forward pass, [k][n]
// for the data gradient, the (C,H,W) -> (H,W,C) permutation of the flatten already inside the column order): the second
// copy is made from the first (2 + 2 bytes per element) instead of from the fp32 master (4 + 2).  Rows r >= R of src up
// to Rbuf and columns up to lds exist and are zero (the pack kernels' padding contract), so whole 16-byte items are
// read; dst's padding is never written.
__global__ __launch_bounds__(256) void transpose_f16_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int R,
                                                            int C, int Rbuf, int lds_, int ldd) {
    __shared__ half_t t[64][72];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;          // 64 rows x 8 items
        const int r = e >> 3, ch = e & 7;
        h8 v = {};
        if (r0 + r < Rbuf && c0 + ch * 8 + 8 <= lds_) v = *(const h8*)(src + (int64_t)(r0 + r) * lds_ + c0 + ch * 8);
        *(h8*)&t[r][ch * 8] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;          // 64 output rows (c) x 8 items of 8 source rows
        const int c = e & 63, rh = e >> 6;
        if (c0 + c < C && r0 + rh * 8 < R) {
            h8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r0 + rh * 8 + j < R) ? t[rh * 8 + j][c] : (half_t)0.f;
            *(h8*)(dst + (int64_t)(c0 + c) * ldd + r0 + rh * 8) = v;
        }
    }
}
// Batched form: every second-orientation fp16 copy of a sub-network in one launch, from a device-resident table.  A row
// of the table is one 2-D transpose: a dense weight, or ONE TAP of one parity class of a stride-2 transposed convolution
// (class block [ci][t' * Cop + co] = forward copy [co][t * Cip + ci]: the [co][ci] slice at column t * Cip of the source,
// written at column t' * Cop of the class block).  `width` = source columns that may be read from src on (its row's end).
__global__ __launch_bounds__(256) void transpose_f16_batch_kernel(const TransposeEntry* __restrict__ tab, int n) {
    __shared__ half_t t[64][72];
    __shared__ int sel;
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int e = threadIdx.x; e < n; e += 64) cnt += (int)blockIdx.x >= tab[e].tile_begin ? 1 : 0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
        if (threadIdx.x == 0) sel = cnt - 1;
    }
    __syncthreads();
    const TransposeEntry& en = tab[sel];
    const int tile = blockIdx.x - en.tile_begin;
    const int ntc = (en.C + 63) / 64;
    const int r0 = (tile / ntc) * 64, c0 = (tile - (tile / ntc) * ntc) * 64;
    const int R = en.R, C = en.C;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int r = e >> 3, ch = e & 7;
        h8 v = {};
        if (r0 + r < en.Rbuf && c0 + ch * 8 + 8 <= en.width) v = *(const h8*)(en.src + (int64_t)(r0 + r) * en.lds + c0 + ch * 8);
        *(h8*)&t[r][ch * 8] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int c = e & 63, rh = e >> 6;
        if (c0 + c < C && r0 + rh * 8 < R) {
            h8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r0 + rh * 8 + j < R) ? t[rh * 8 + j][c] : (half_t)0.f;
            *(h8*)(en.dst + (int64_t)(c0 + c) * en.ldd + r0 + rh * 8) = v;
        }
    }
}
int transpose_batch_launch(const TransposeEntry* tab, int n, int total_tiles, hipStream_t st) {
    if (n < 1 || total_tiles < 1) return OK;
    hipLaunchKernelGGL(transpose_f16_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, tab, n);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

int transpose_f16_launch(const half_t* src, half_t* dst, int R, int C, int Rbuf, int lds_, int ldd, hipStream_t st) {
    hipLaunchKernelGGL(transpose_f16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, src, dst, R, C, Rbuf,
                       lds_, ldd);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------------------------
// fmri_apply_batch (round 4): ONE launch per sub-network between its last weight gradient and its next forward pass.
// Rounds 1-3 ran, per parameter tensor, unpack_grad (slab sum + map to the reference layout, read-modify-write of the
// gradient buffer), then the optimizer over the whole buffer, then the re-pack into both GEMM orientations, with a
// memset of the gradient buffer in front: 48 bytes per parameter and ~35 launches per step.  Here a block takes the
// same tile the unpack kernel took, sums the slabs, transposes through LDS, updates w and its RMSprop state in place
// (the same float operations in the same order as unpack_tile_kernel + rmsprop_kernel: results are bit-identical),
// transposes the NEW weights back and stores the fp16 GEMM copy of the gradient's orientation: 22 bytes per parameter.
// The other orientation is still a pack launch.  1-D parameters (biases, BatchNorm gamma / beta) are kind-2 rows:
// flat segments of the buffer whose gradients the backward pass accumulated in place; mode 2 clears those segments (the
// start of a backward pass: instead of a memset of the whole gradient buffer).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float slab_sum(const float* q, int nslabs, int64_t slab_stride) {
    // (unpack_tile_kernel's order: four loads in flight)
    float v0 = q[0], v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int z = 1;
    for (; z + 3 < nslabs; z += 4) {
        v0 += q[z * slab_stride];
        v1 += q[(z + 1) * slab_stride];
        v2 += q[(z + 2) * slab_stride];
        v3 += q[(z + 3) * slab_stride];
    }
    for (; z < nslabs; ++z) v0 += q[z * slab_stride];
    return (v0 + v1) + (v2 + v3);
}
__device__ __forceinline__ void slab_clear(float* q, int nslabs, int64_t slab_stride) {
    for (int z = 0; z < nslabs; ++z) q[z * slab_stride] = 0.f;
}
// rmsprop_kernel's update of one element (csrc/loss.hip); returns the new parameter
__device__ __forceinline__ float rms_update(float g, float* w, float* sq, float gs, float lr, const ApplyOpt& o) {
    float gg = g * gs;
    if (o.clamp > 0.f) gg = fminf(fmaxf(gg, -o.clamp), o.clamp);
    const float s = o.alpha * *sq + (1.f - o.alpha) * gg * gg;
    *sq = s;
    const float nw = *w - lr * gg / (sqrtf(s) + o.eps);
    *w = nw;
    return nw;
}

__global__ __launch_bounds__(256) void apply_batch_kernel(const ApplyEntry* __restrict__ tab, int n, const ApplyOpt o) {
    __shared__ float t[64 * 65];
    __shared__ int sel;
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int e = threadIdx.x; e < n; e += 64) cnt += (int)blockIdx.x >= tab[e].tile_begin ? 1 : 0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
        if (threadIdx.x == 0) sel = cnt - 1;
    }
    __syncthreads();
    const ApplyEntry& en = tab[sel];
    const int tile = blockIdx.x - en.tile_begin;
    const bool update = o.mode == 1 || o.mode == 3;
    const bool from_grad = o.mode == 3;     // the gradient is read from the reference layout (summed over the ranks there)
    const bool live = !update || !o.flag || *o.flag != 0;          // a gated-off optimizer step changes nothing
    if (!live && (!en.clear || en.kind == 2 || from_grad || o.gated)) return;
    const float lr = (update && o.lr_dev) ? *o.lr_dev : 0.f;
    const float gs = update ? o.gscale / (o.gdev ? *o.gdev : 1.f) : 1.f;
    if (en.kind == 2) {
        if (o.mode == 0) return;                                   // (the gradient is already where it belongs)
        const int64_t i = (int64_t)tile * APPLY_CHUNK + threadIdx.x;
#pragma unroll
        for (int k = 0; k < APPLY_CHUNK / 256; ++k) {
            const int64_t j = i + k * 256;
            if (j < en.n) {
                if (o.mode == 2) en.grad[j] = 0.f;
                else rms_update(en.grad[j], en.w + j, en.sq + j, gs, lr, o);
            }
        }
        return;
    }
    if (o.mode == 2) return;
    float* gsrc = const_cast<float*>(en.gsrc);
    if (en.kind == 1) {
        const int64_t total = (int64_t)en.TA * en.A * en.B;
        const int64_t i0 = (int64_t)tile * APPLY_CHUNK + threadIdx.x;
#pragma unroll
        for (int k = 0; k < APPLY_CHUNK / 256; ++k) {
            const int64_t i = i0 + k * 256;
            if (i >= total) break;
            const int row = (int)(i / en.B), b = (int)(i - (int64_t)row * en.B);
            const int ta = row / en.A, a = row - ta * en.A;
            const int64_t off = a * en.sa + ta * en.sta + b * en.sb;
            float v;
            if (from_grad) {
                v = en.grad[off];
            } else {
                float* q = gsrc + (int64_t)row * en.ld + b;
                v = slab_sum(q, en.nslabs, en.slab_stride) * en.scale;
                if (en.clear) slab_clear(q, en.nslabs, en.slab_stride);
            }
            if (!live) continue;
            if (!update) { en.grad[off] = v; continue; }
            const float nw = rms_update(v, en.w + off, en.sq + off, gs, lr, o);
            if (en.pk) en.pk[(int64_t)row * en.kpad + b] = (half_t)nw;
        }
        return;
    }
    // kind 0: one row x bt consecutive b x all taps
    const int bt = en.bt, sh = bt == 64 ? 6 : 5, run = en.run;
    const int nbt = (en.B + bt - 1) / bt;
    const int row = tile / nbt, b0 = (tile - row * nbt) * bt;
    const int ta = row / en.A, a = row - ta * en.A;
    const int stride = run | 1;
    float* s = gsrc + (int64_t)row * en.ld + b0;
    if (!from_grad) {
        for (int e = threadIdx.x; e < run * bt; e += 256) {
            const int tb = e >> sh, bl = e & (bt - 1);
            if (b0 + bl < en.B) {
                float* q = s + tb * en.Bp + bl;
                t[bl * stride + tb] = slab_sum(q, en.nslabs, en.slab_stride);
                if (en.clear) slab_clear(q, en.nslabs, en.slab_stride);
            }
        }
        // columns behind the taps belong to the weight-gradient kernel too (the narrow kernel keeps sum_pixels P, a bias
        // gradient, in the first of them): the row's first block hands them back zeroed as well
        if (en.clear && b0 == 0)
            for (int c = run * en.Bp + threadIdx.x; c < en.ld; c += 256)
                slab_clear(gsrc + (int64_t)row * en.ld + c, en.nslabs, en.slab_stride);
        if (!live) return;
        __syncthreads();
    }
    const int64_t base = a * en.sa + ta * en.sta;
    for (int e = threadIdx.x; e < bt * run; e += 256) {
        const int bl = e / run, j = e - bl * run;
        if (b0 + bl < en.B) {
            const int64_t off = base + (b0 + bl) * en.sb + j;
            const float v = from_grad ? en.grad[off] : t[bl * stride + j] * en.scale;
            if (!update) en.grad[off] = v;
            else t[bl * stride + j] = rms_update(v, en.w + off, en.sq + off, gs, lr, o);
        }
    }
    if (!update || !en.pk) return;
    __syncthreads();
    half_t* d = en.pk + (int64_t)row * en.kpad + b0;
    for (int e = threadIdx.x; e < run * bt; e += 256) {
        const int tb = e >> sh, bl = e & (bt - 1);
        if (b0 + bl < en.B) d[tb * en.Bp + bl] = (half_t)t[bl * stride + tb];
    }
}

// Completes a row of the table (kind, run, bt) from the tap geometry of its PackSpec and returns the number of blocks it
// occupies; 0 = this tensor's layout map is not one the kernel knows (the caller keeps the separate launches).
int apply_entry_tiles(ApplyEntry& e, int TH, int TW, int KW, int py, int px, int step, int64_t stb) {
    if (e.kind == 2) return (int)((e.n + APPLY_CHUNK - 1) / APPLY_CHUNK);
    const int run = TH * TW;
    const int64_t rows = (int64_t)e.TA * e.A;
    if (run > 1 && run <= 64 && stb == 1 && py == 0 && px == 0 && step == 1 && TW == KW) {
        e.kind = 0; e.run = run; e.bt = e.B >= 64 ? 64 : 32;
        const int64_t tiles = rows * ((e.B + e.bt - 1) / e.bt);
        return tiles < (1 << 24) ? (int)tiles : 0;
    }
    if (run == 1 && e.sb == 1) {
        e.kind = 1; e.run = 1; e.bt = 0;
        const int64_t tiles = (rows * e.B + APPLY_CHUNK - 1) / APPLY_CHUNK;
        return tiles < (1 << 24) ? (int)tiles : 0;
    }
    return 0;
}
int apply_batch_launch(const ApplyEntry* tab, int n, int total_tiles, const ApplyOpt& o, hipStream_t st) {
    if (total_tiles < 1 || n < 1) return OK;
    hipLaunchKernelGGL(apply_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, tab, n, o);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
