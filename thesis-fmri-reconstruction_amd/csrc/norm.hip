// Batch-norm (train mode) kernels over NHWC fp16 rows [M][C]  (BatchNorm2d and BatchNorm1d alike).
//
// Replaces nn.BatchNorm2d / nn.BatchNorm1d(momentum=0.9) + ReLU of the reference blocks
// (models/vae_gan.py:21,28-29,54,58-59,81,108,158) and their autograd backward:
//   stats    : per-channel sum / sum-of-squares                                      -- HBM bound
//   finalize : mean, rstd, scale/shift, running-stat update (momentum 0.9, unbiased running var,
//              `updates` consecutive updates for the discriminator's REC+GAN double pass)
//   apply    : y = relu(x*scale + shift)                                               -- HBM bound
//   bwd_reduce / bwd_apply : dgamma, dbeta and dx through ReLU + BN (batch statistics)
// The sums are kept outside the kernels so that a data-parallel run can all-reduce them (SyncBN).
//
// All streaming kernels use one thread mapping: a 256-thread block is CX chunk-columns (8 channels =
// 16 bytes each) x RY row lanes; a thread keeps its channel chunk for the whole kernel, so the
// per-channel parameters live in registers and a wave always touches whole contiguous rows.  Row loops
// are unrolled x4 to keep 4 independent 16-byte loads in flight per lane.  Reductions write per-block
// partials to a workspace and a second tiny kernel folds them (no float atomics: 2048 blocks adding
// into the same 2*C words serialise at the memory side).
#include "kernels.h"

namespace fmri {

struct RowGeom {
    int cx_log2;
    int gx, gy;
};

static RowGeom row_geometry(int M, int C, int max_gy) {
    RowGeom g;
    const int nch = C / 8;
    g.cx_log2 = 0;
    while ((1 << g.cx_log2) < nch && g.cx_log2 < 8) ++g.cx_log2;
    const int CX = 1 << g.cx_log2, RY = 256 >> g.cx_log2;
    g.gx = (nch + CX - 1) / CX;
    int gy = (M + RY * 16 - 1) / (RY * 16);          // >= 16 rows per thread
    int cap = 768 / g.gx;                             // ~3 blocks per CU
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    g.gy = gy;
    return g;
}

// MODE 0: sum x, sum x^2.  MODE 1: sum g, sum g*xhat (g = dy masked by the ReLU of the forward).
// MODE 2: activation backward: dpre = dy*act'(y) written to `dout`, column sums of dpre.
// fp16 store of a BatchNorm-backward result that SATURATES at +-65504 instead of overflowing to inf.  dx = gamma * rstd *
// (...) is the one place of the backward pass where a healthy cotangent is multiplied by an unbounded factor: a feature
// whose batch variance is tiny has rstd up to 1 / sqrt(eps) = 316 -- and 1 / (s sqrt(eps)) behind a range-scaled latent
// batch (bn_finalize_channel) -- so a few of the 8 M results of a step that follows a latent excursion cross fp16's
// range (measured: 1-6 values per such step).  An inf there turns the whole step into NaN (inf - inf in the
// GEMMs that consume it); the saturated value is a clipped gradient for the handful of weights it touches.
// A NaN stays a NaN (v_med3_f32 would return one of the bounds for it): only finite overflow is clipped.
__device__ __forceinline__ half_t sat16(float v) {
    return (half_t)(v == v ? __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f) : v);
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                        half_t* __restrict__ dout, int M, int C, int cx_log2,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu_or_act,
                                                        float* __restrict__ part /* [gridDim.y][2][C] */) {
    __shared__ float red[256 * 17];
    const int CX = 1 << cx_log2;
    const int RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1);
    const int ry = threadIdx.x >> cx_log2;
    const int chunk = blockIdx.x * CX + cx;
    const int nch = C >> 3;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (chunk < nch) {
        float mu[8], rs[8], ga[8], be[8];
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mu[j] = mean[chunk * 8 + j]; rs[j] = rstd[chunk * 8 + j];
                ga[j] = gamma[chunk * 8 + j]; be[j] = beta[chunk * 8 + j];
            }
        }
        const int stride = gridDim.y * RY;
        const int64_t coff = (int64_t)chunk * 8;
        auto body = [&](const h8& xv, const h8& gv, int m) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)xv[j]; s0[j] += f; s1[j] += f * f; }
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    float g = (float)gv[j];
                    if (relu_or_act && !(xh * ga[j] + be[j] > 0.f)) g = 0.f;
                    s0[j] += g; s1[j] += g * xh;
                }
            } else {
                h8 ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float yy = (float)xv[j];
                    float g = (float)gv[j];
                    if (relu_or_act == ACT_RELU) g = yy > 0.f ? g : 0.f;
                    else if (relu_or_act == ACT_TANH) g = g * (1.f - yy * yy);
                    ov[j] = (half_t)g;
                    s0[j] += g;
                }
                *(h8*)(dout + (int64_t)m * C + coff) = ov;
            }
        };
        int m = blockIdx.y * RY + ry;
        for (; m + 3 * stride < M; m += 4 * stride) {
            h8 xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *(const h8*)(x + (int64_t)(m + u * stride) * C + coff);
                if (MODE != 0) gv[u] = *(const h8*)(dy + (int64_t)(m + u * stride) * C + coff);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(xv[u], gv[u], m + u * stride);
        }
        for (; m < M; m += stride) {
            h8 xv = *(const h8*)(x + (int64_t)m * C + coff), gv;
            if (MODE != 0) gv = *(const h8*)(dy + (int64_t)m * C + coff);
            body(xv, gv, m);
        }
    }
    if (part == nullptr) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 17 + j] = s0[j]; red[threadIdx.x * 17 + 8 + j] = s1[j]; }
    __syncthreads();
    // 16 values per chunk column, summed over the RY row lanes; spread over the block's threads
    for (int t = threadIdx.x; t < CX * 16; t += 256) {
        const int c = t >> 4, j = t & 15;
        const int ch = blockIdx.x * CX + c;
        if (ch >= nch) continue;
        float s = 0.f;
        for (int r = 0; r < RY; ++r) s += red[((r << cx_log2) + c) * 17 + j];
        part[((int64_t)blockIdx.y * 2 + (j >> 3)) * C + ch * 8 + (j & 7)] = s;
    }
}

// sums[i] = sum_p part[p][i], i < n (n = 2*C).  A block folds 32 columns with 8 row groups (4 loads in
// flight per thread): the partial matrix is small but a single serial chain per column is latency bound.
// Optionally accumulates the folded sums into parameter gradients (BN backward: g0 += gscale * sums[0..C) = d beta,
// g1 += gscale * sums[C..2C) = d gamma) -- one writer per element, no atomics.
__global__ __launch_bounds__(1024) void fold_partials_kernel(const float* __restrict__ part, int nparts, int n,
                                                             float* __restrict__ sums, float* __restrict__ g0,
                                                             float* __restrict__ g1, float gscale, int gC, int gOff) {
    // 32 columns x 32 row lanes per block: the kernel is a chain of memory latencies (few blocks, tiny data), so the
    // partial rows are spread over as many lanes as a block has and each lane keeps four loads in flight
    __shared__ float red[32][33];
    const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = gy;
        for (; p + 96 < nparts; p += 128) {
            s0 += part[(int64_t)p * n + i];
            s1 += part[(int64_t)(p + 32) * n + i];
            s2 += part[(int64_t)(p + 64) * n + i];
            s3 += part[(int64_t)(p + 96) * n + i];
        }
        for (; p < nparts; p += 32) s0 += part[(int64_t)p * n + i];
    }
    red[gy][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][cx];
        sums[i] = s;
        // columns [gOff, gOff + C) -> g0, [gOff + C, gOff + 2C) -> g1 (one cotangent stream's sums)
        const int C = gC > 0 ? gC : (n >> 1);
        if (g0 && i >= gOff && i < gOff + C) g0[i - gOff] += gscale * s;
        if (g1 && i >= gOff + C && i < gOff + 2 * C) g1[i - gOff - C] += gscale * s;
    }
}

// stage 1 of a long fold (statistics rows written by a contraction's epilogue, StatEpi): block (x, y) sums rows
// [y*per, (y+1)*per) of columns 32x .. 32x+31 into out[y][n].  Same lane layout as fold_partials_kernel.
__global__ __launch_bounds__(1024) void fold_rows_kernel(const float* __restrict__ part, int nparts, int n, int per,
                                                         float* __restrict__ out, int64_t part_gstride,
                                                         int64_t out_gstride) {
    part += blockIdx.z * part_gstride;      // statistics group
    out += blockIdx.z * out_gstride;
    __shared__ float red[32][33];
    const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cx;
    const int lo = blockIdx.y * per;
    const int hi = lo + per < nparts ? lo + per : nparts;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = lo + gy;
        for (; p + 96 < hi; p += 128) {
            s0 += part[(int64_t)p * n + i];
            s1 += part[(int64_t)(p + 32) * n + i];
            s2 += part[(int64_t)(p + 64) * n + i];
            s3 += part[(int64_t)(p + 96) * n + i];
        }
        for (; p < hi; p += 32) s0 += part[(int64_t)p * n + i];
    }
    red[gy][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][cx];
        out[(int64_t)blockIdx.y * n + i] = s;
    }
}

// BatchNorm-backward statistics rows of a contraction's epilogue (BnBwdEpi), G cotangent groups at once: block (x, g)
// folds columns 32x .. 32x+31 of group g's rows into sums[g][n] (n = 2C: sum g | sum g*xhat); the group `pgroup` also
// accumulates the parameter gradients d beta += gscale * sum g, d gamma += gscale * sum g*xhat (one writer per element).
__global__ __launch_bounds__(1024) void fold_groups_kernel(const float* __restrict__ part, int nparts, int n,
                                                           int64_t part_gstride, float* __restrict__ sums,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                           float gscale, int pgroup) {
    __shared__ float red[32][33];
    part += blockIdx.y * part_gstride;
    const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = gy;
        for (; p + 96 < nparts; p += 128) {
            s0 += part[(int64_t)p * n + i];
            s1 += part[(int64_t)(p + 32) * n + i];
            s2 += part[(int64_t)(p + 64) * n + i];
            s3 += part[(int64_t)(p + 96) * n + i];
        }
        for (; p < nparts; p += 32) s0 += part[(int64_t)p * n + i];
    }
    red[gy][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][cx];
        sums[(int64_t)blockIdx.y * n + i] = s;
        const int C = n >> 1;
        if ((int)blockIdx.y == pgroup) {
            if (dbeta && i < C) dbeta[i] += gscale * s;
            if (dgamma && i >= C) dgamma[i - C] += gscale * s;
        }
    }
}

// shared by bn_finalize_kernel and fold_finalize_kernel: channel c from its batch sums.
// ``in_s``: the rows are s * x for a power of two s (a latent batch stored range-scaled, fmri_latent_fwd_ranged):
// BN_eps(x) == BN_{eps s^2}(s x), so the normalisation runs on the stored values with eps * s^2 -- mean / rstd / scale /
// shift are those of the STORED rows (what the apply and backward kernels read) -- and the running statistics receive
// the true-scale mean / s and var / s^2.
__device__ __forceinline__ void bn_finalize_channel(int c, float sx, float sxx, float count, const float* gamma,
                                                    const float* beta, float eps, float momentum, int updates,
                                                    float* running_mean, float* running_var, float* mean_out,
                                                    float* rstd_out, float* scale_out, float* shift_out,
                                                    float in_s = 1.f) {
    const float mean = sx / count;
    float var = sxx / count - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + eps * in_s * in_s);
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    const float sc = gamma[c] * rstd;
    scale_out[c] = sc;
    shift_out[c] = beta[c] - mean * sc;
    if (running_mean && updates > 0) {
        const float inv_s = 1.f / in_s;
        const float unb = (count > 1.f ? var * count / (count - 1.f) : var) * inv_s * inv_s;
        const float mean_t = mean * inv_s;
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * mean_t;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// fold of the statistics partials [nparts][2][C] + finalize in one launch (forward BatchNorm without a statistics
// exchange between ranks): block = 32 channels x 32 row lanes, both sums of a channel are folded by the same lanes.
__global__ __launch_bounds__(1024) void fold_finalize_kernel(const float* __restrict__ part, int nparts, int C,
                                                             float* __restrict__ sums, float count,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float momentum,
                                                             int updates, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             float* __restrict__ scale_out,
                                                             float* __restrict__ shift_out, long long* __restrict__ nbt) {
    __shared__ float red[2][32][33];
    const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int n = 2 * C;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (c < C) {
        int p = gy;
        for (; p + 32 < nparts; p += 64) {
            a0 += part[(int64_t)p * n + c];
            b0 += part[(int64_t)p * n + C + c];
            a1 += part[(int64_t)(p + 32) * n + c];
            b1 += part[(int64_t)(p + 32) * n + C + c];
        }
        for (; p < nparts; p += 32) {
            a0 += part[(int64_t)p * n + c];
            b0 += part[(int64_t)p * n + C + c];
        }
    }
    red[0][gy][cx] = a0 + a1;
    red[1][gy][cx] = b0 + b1;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && updates > 0) *nbt += updates;      // num_batches_tracked
    if (gy == 0 && c < C) {
        float sx = 0.f, sxx = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) { sx += red[0][r][cx]; sxx += red[1][r][cx]; }
        sums[c] = sx;
        sums[C + c] = sxx;
        bn_finalize_channel(c, sx, sxx, count, gamma, beta, eps, momentum, updates, running_mean, running_var, mean_out,
                            rstd_out, scale_out, shift_out);
    }
}

// one thread per channel
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, int updates,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out,
                                   long long* __restrict__ nbt, const float* __restrict__ in_scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt && updates > 0) *nbt += updates;      // num_batches_tracked
    if (c >= C) return;
    bn_finalize_channel(c, sums[c], sums[C + c], count, gamma, beta, eps, momentum, updates, running_mean, running_var,
                        mean_out, rstd_out, scale_out, shift_out, in_scale ? *in_scale : 1.f);
}

// MODE 0: y = act(x*scale + shift).   MODE 1: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M), g = dy*mask
template <int MODE>
__global__ __launch_bounds__(256) void bn_stream_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                        half_t* __restrict__ out, int M, int C, int cx_log2,
                                                        const float* __restrict__ p0, const float* __restrict__ p1,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu, float inv_count,
                                                        const float* __restrict__ sums) {
    const int CX = 1 << cx_log2;
    const int RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1);
    const int ry = threadIdx.x >> cx_log2;
    const int chunk = blockIdx.x * CX + cx;
    if (chunk >= (C >> 3)) return;
    // MODE 0: a = scale, b = shift.  MODE 1: xhat = (x - b)*a with a = rstd, b = mean;
    //         dx = k*(g - c0 - xhat*c1), k = gamma*rstd, c0 = sum_g/M, c1 = sum_gx/M
    float a[8], b[8], k[8], c0[8], c1[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j;
        if (MODE == 0) {
            a[j] = p0[c]; b[j] = p1[c];
        } else {
            const float mu = p0[c], rs = p1[c];
            a[j] = rs; b[j] = mu; ga[j] = gamma[c]; be[j] = beta[c];
            k[j] = ga[j] * rs; c0[j] = sums[c] * inv_count; c1[j] = sums[C + c] * inv_count;
        }
    }
    const int stride = gridDim.y * RY;
    const int64_t coff = (int64_t)chunk * 8;
    auto body = [&](const h8& xv, const h8& gv, int m) {
        h8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                float f = (float)xv[j] * a[j] + b[j];
                if (relu) f = f > 0.f ? f : 0.f;
                ov[j] = (half_t)f;
            } else {
                const float xh = ((float)xv[j] - b[j]) * a[j];
                float g = (float)gv[j];
                if (relu && !(xh * ga[j] + be[j] > 0.f)) g = 0.f;
                ov[j] = sat16(k[j] * (g - c0[j] - xh * c1[j]));
            }
        }
        *(h8*)(out + (int64_t)m * C + coff) = ov;
    };
    int m = blockIdx.y * RY + ry;
    for (; m + 3 * stride < M; m += 4 * stride) {
        h8 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = *(const h8*)(x + (int64_t)(m + u * stride) * C + coff);
            if (MODE == 1) gv[u] = *(const h8*)(dy + (int64_t)(m + u * stride) * C + coff);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) body(xv[u], gv[u], m + u * stride);
    }
    for (; m < M; m += stride) {
        h8 xv = *(const h8*)(x + (int64_t)m * C + coff), gv;
        if (MODE == 1) gv = *(const h8*)(dy + (int64_t)m * C + coff);
        body(xv, gv, m);
    }
}

// ---- BatchNorm backward of TWO cotangent streams through one saved forward (the discriminator's logit stream A and
// feature stream B, stacked as dy = [A rows | B rows], M rows each).  The forward tensor x and everything derived from
// it (xhat, the ReLU mask) are read and computed once for both: 3 + 5 tensor passes instead of 2 x (2 + 3).
// partials / sums layout: [A: sum g | A: sum g*xhat | B: sum g | B: sum g*xhat], C floats each.
__global__ __launch_bounds__(256) void bn_reduce2_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                         int M, int C, int cx_log2, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int relu,
                                                         float* __restrict__ part /* [gridDim.y][4][C] */) {
    __shared__ float red[256 * 17];
    const int CX = 1 << cx_log2;
    const int RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1);
    const int ry = threadIdx.x >> cx_log2;
    const int chunk = blockIdx.x * CX + cx;
    const int nch = C >> 3;
    float sa0[8], sa1[8], sb0[8], sb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa0[j] = 0.f; sa1[j] = 0.f; sb0[j] = 0.f; sb1[j] = 0.f; }
    if (chunk < nch) {
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = mean[chunk * 8 + j]; rs[j] = rstd[chunk * 8 + j];
            ga[j] = gamma[chunk * 8 + j]; be[j] = beta[chunk * 8 + j];
        }
        const int stride = gridDim.y * RY;
        const int64_t coff = (int64_t)chunk * 8;
        const half_t* dyb = dy + (int64_t)M * C;
        auto body = [&](const h8& xv, const h8& ga_v, const h8& gb_v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = ((float)xv[j] - mu[j]) * rs[j];
                const bool on = !relu || (xh * ga[j] + be[j] > 0.f);
                const float g_a = on ? (float)ga_v[j] : 0.f, g_b = on ? (float)gb_v[j] : 0.f;
                sa0[j] += g_a; sa1[j] += g_a * xh;
                sb0[j] += g_b; sb1[j] += g_b * xh;
            }
        };
        int m = blockIdx.y * RY + ry;
        for (; m + 3 * stride < M; m += 4 * stride) {
            h8 xv[4], av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t o = (int64_t)(m + u * stride) * C + coff;
                xv[u] = *(const h8*)(x + o);
                av[u] = *(const h8*)(dy + o);
                bv[u] = *(const h8*)(dyb + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(xv[u], av[u], bv[u]);
        }
        for (; m < M; m += stride) {
            const int64_t o = (int64_t)m * C + coff;
            body(*(const h8*)(x + o), *(const h8*)(dy + o), *(const h8*)(dyb + o));
        }
    }
    // block reduction of the 16 per-thread values of a stream over the RY row lanes, stream A then stream B
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[threadIdx.x * 17 + j] = s ? sb0[j] : sa0[j];
            red[threadIdx.x * 17 + 8 + j] = s ? sb1[j] : sa1[j];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < CX * 16; t += 256) {
            const int c = t >> 4, j = t & 15;
            const int ch = blockIdx.x * CX + c;
            if (ch >= nch) continue;
            float v = 0.f;
            for (int r = 0; r < RY; ++r) v += red[((r << cx_log2) + c) * 17 + j];
            part[((int64_t)blockIdx.y * 4 + 2 * s + (j >> 3)) * C + ch * 8 + (j & 7)] = v;
        }
        __syncthreads();
    }
}

// dx_s = gamma*rstd*(g_s - sum_g_s/M - xhat*sum_gx_s/M) for both streams, x read once
__global__ __launch_bounds__(256) void bn_stream2_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                         half_t* __restrict__ out, int M, int C, int cx_log2,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int relu, float inv_count,
                                                         const float* __restrict__ sums /* [4][C] */) {
    const int CX = 1 << cx_log2;
    const int RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1);
    const int ry = threadIdx.x >> cx_log2;
    const int chunk = blockIdx.x * CX + cx;
    if (chunk >= (C >> 3)) return;
    float a[8], b[8], k[8], ga[8], be[8], a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j;
        a[j] = rstd[c]; b[j] = mean[c]; ga[j] = gamma[c]; be[j] = beta[c];
        k[j] = ga[j] * a[j];
        a0[j] = sums[c] * inv_count; a1[j] = sums[C + c] * inv_count;
        b0[j] = sums[2 * C + c] * inv_count; b1[j] = sums[3 * C + c] * inv_count;
    }
    const int stride = gridDim.y * RY;
    const int64_t coff = (int64_t)chunk * 8;
    const int64_t sb = (int64_t)M * C;
    auto body = [&](const h8& xv, const h8& av, const h8& bv, int m) {
        h8 oa, ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)xv[j] - b[j]) * a[j];
            const bool on = !relu || (xh * ga[j] + be[j] > 0.f);
            const float g_a = on ? (float)av[j] : 0.f, g_b = on ? (float)bv[j] : 0.f;
            oa[j] = sat16(k[j] * (g_a - a0[j] - xh * a1[j]));
            ob[j] = sat16(k[j] * (g_b - b0[j] - xh * b1[j]));
        }
        const int64_t o = (int64_t)m * C + coff;
        *(h8*)(out + o) = oa;
        *(h8*)(out + sb + o) = ob;
    };
    int m = blockIdx.y * RY + ry;
    for (; m + stride < M; m += 2 * stride) {
        h8 xv[2], av[2], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t o = (int64_t)(m + u * stride) * C + coff;
            xv[u] = *(const h8*)(x + o);
            av[u] = *(const h8*)(dy + o);
            bv[u] = *(const h8*)(dy + sb + o);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) body(xv[u], av[u], bv[u], m + u * stride);
    }
    for (; m < M; m += stride) {
        const int64_t o = (int64_t)m * C + coff;
        body(*(const h8*)(x + o), *(const h8*)(dy + o), *(const h8*)(dy + sb + o), m);
    }
}

// ---- BatchNorm over FEW rows (the dense layers: BatchNorm1d behind fc.0 / fc1.0, M = batch rows) in ONE launch per
// direction.  The streaming kernels above pay three launches per call (partial sums, fold, apply) -- 15-20 us for half a
// megabyte of data, six times per step and direction.  Here a block owns 32 channels (4 chunk columns x 64 row lanes) for
// all rows: pass 1 sums its columns, the block folds them in LDS and finalizes (mean, rstd, scale, shift, running
// statistics / the backward constants and the parameter gradients), pass 2 re-reads the rows (L2) and writes the result.
// Fixed summation order: bit-reproducible.
constexpr int COLS_CX = 4, COLS_RY = 64;

// dst[16] <- sums of the 16 per-thread values of chunk column cx over the row lanes (called by threads < COLS_CX * 16)
__device__ __forceinline__ float cols_fold(const float* red, int cx, int j) {
    float s = 0.f;
    for (int r = 0; r < COLS_RY; ++r) s += red[((r * COLS_CX) + cx) * 17 + j];
    return s;
}

__global__ __launch_bounds__(256) void bn_cols_fwd_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int M,
                                                          int C, float count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          int updates, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ scale_out,
                                                          float* __restrict__ shift_out, float* __restrict__ sums,
                                                          long long* __restrict__ nbt, int relu,
                                                          const float* __restrict__ in_scale) {
    __shared__ float red[256 * 17];
    __shared__ float par[COLS_CX * 8][2];
    const int cx = threadIdx.x & (COLS_CX - 1), ry = threadIdx.x >> 2;
    const int chunk = blockIdx.x * COLS_CX + cx;
    const int nch = C >> 3;
    const int64_t coff = (int64_t)chunk * 8;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (chunk < nch)
        for (int m = ry; m < M; m += COLS_RY) {
            const h8 v = *(const h8*)(x + (int64_t)m * C + coff);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s0[j] += f; s1[j] += f * f; }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 17 + j] = s0[j]; red[threadIdx.x * 17 + 8 + j] = s1[j]; }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && updates > 0) *nbt += updates;      // num_batches_tracked
    if (threadIdx.x < COLS_CX * 8) {
        const int c_ = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int c = (blockIdx.x * COLS_CX + c_) * 8 + j;
        if (c < C) {
            const float sx = cols_fold(red, c_, j), sxx = cols_fold(red, c_, 8 + j);
            sums[c] = sx;
            sums[C + c] = sxx;
            bn_finalize_channel(c, sx, sxx, count, gamma, beta, eps, momentum, updates, running_mean, running_var, mean_out,
                                rstd_out, scale_out, shift_out, in_scale ? *in_scale : 1.f);
            par[threadIdx.x][0] = scale_out[c];
            par[threadIdx.x][1] = shift_out[c];
        }
    }
    __syncthreads();
    if (chunk >= nch) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = par[cx * 8 + j][0]; b[j] = par[cx * 8 + j][1]; }
    for (int m = ry; m < M; m += COLS_RY) {
        const h8 v = *(const h8*)(x + (int64_t)m * C + coff);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * a[j] + b[j];
            if (relu) f = f > 0.f ? f : 0.f;
            o[j] = (half_t)f;
        }
        *(h8*)(y + (int64_t)m * C + coff) = o;
    }
}

// backward through (ReLU o BN) of NS cotangent streams stacked along the rows (dy = [stream 0 rows | stream 1 rows]);
// sums [NS][2][C] = (sum g | sum g*xhat) per stream; dbeta / dgamma (may be null) += gscale * sums of stream `pstream`
template <int NS>
__global__ __launch_bounds__(256) void bn_cols_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                          half_t* __restrict__ dx, int M, int C, float inv_count,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu,
                                                          float* __restrict__ sums, float* __restrict__ dbeta,
                                                          float* __restrict__ dgamma, float gscale, int pstream) {
    __shared__ float red[256 * 17];
    __shared__ float par[NS][COLS_CX * 8][2];
    const int cx = threadIdx.x & (COLS_CX - 1), ry = threadIdx.x >> 2;
    const int chunk = blockIdx.x * COLS_CX + cx;
    const int nch = C >> 3;
    const int64_t coff = (int64_t)chunk * 8;
    const int64_t sstride = (int64_t)M * C;
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk < nch ? chunk * 8 + j : 0;
        mu[j] = mean[c]; rs[j] = rstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    }
    for (int st = 0; st < NS; ++st) {
        float s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
        if (chunk < nch)
            for (int m = ry; m < M; m += COLS_RY) {
                const int64_t o = (int64_t)m * C + coff;
                const h8 xv = *(const h8*)(x + o);
                const h8 gv = *(const h8*)(dy + st * sstride + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    float g = (float)gv[j];
                    if (relu && !(xh * ga[j] + be[j] > 0.f)) g = 0.f;
                    s0[j] += g; s1[j] += g * xh;
                }
            }
        __syncthreads();                     // (the previous stream's fold is done with `red`)
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[threadIdx.x * 17 + j] = s0[j]; red[threadIdx.x * 17 + 8 + j] = s1[j]; }
        __syncthreads();
        if (threadIdx.x < COLS_CX * 8) {
            const int c_ = threadIdx.x >> 3, j = threadIdx.x & 7;
            const int c = (blockIdx.x * COLS_CX + c_) * 8 + j;
            if (c < C) {
                const float sg = cols_fold(red, c_, j), sgx = cols_fold(red, c_, 8 + j);
                sums[(st * 2) * C + c] = sg;
                sums[(st * 2 + 1) * C + c] = sgx;
                par[st][threadIdx.x][0] = sg * inv_count;
                par[st][threadIdx.x][1] = sgx * inv_count;
                if (st == pstream) {
                    if (dbeta) dbeta[c] += gscale * sg;
                    if (dgamma) dgamma[c] += gscale * sgx;
                }
            }
        }
    }
    __syncthreads();
    if (chunk >= nch) return;
    for (int st = 0; st < NS; ++st) {
        float c0[8], c1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { c0[j] = par[st][cx * 8 + j][0]; c1[j] = par[st][cx * 8 + j][1]; }
        for (int m = ry; m < M; m += COLS_RY) {
            const int64_t o = (int64_t)m * C + coff;
            const h8 xv = *(const h8*)(x + o);
            const h8 gv = *(const h8*)(dy + st * sstride + o);
            h8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = ((float)xv[j] - mu[j]) * rs[j];
                float g = (float)gv[j];
                if (relu && !(xh * ga[j] + be[j] > 0.f)) g = 0.f;
                ov[j] = sat16(ga[j] * rs[j] * (g - c0[j] - xh * c1[j]));
            }
            *(h8*)(dx + st * sstride + o) = ov;
        }
    }
}

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? OK : E_LAUNCH)

int64_t bn_ws_floats(int M, int C) {
    const RowGeom g = row_geometry(M, C, 1 << 30);
    return (int64_t)g.gy * 2 * C;
}

template <int MODE>
static int reduce_launch(const half_t* x, const half_t* dy, half_t* dout, int M, int C, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int flag, float* sums, float* ws,
                         int64_t ws_floats, hipStream_t st, float* g0 = nullptr, float* g1 = nullptr,
                         float gscale = 0.f, int g_count = 0) {
    int max_gy = 1 << 30;
    if (sums) {
        if (!ws || ws_floats < 2 * (int64_t)C) return E_WORKSPACE;
        max_gy = (int)(ws_floats / (2 * (int64_t)C));
    }
    const RowGeom g = row_geometry(M, C, max_gy);
    hipLaunchKernelGGL((bn_reduce_kernel<MODE>), dim3(g.gx, g.gy), dim3(256), 0, st, x, dy, dout, M, C, g.cx_log2,
                       mean, rstd, gamma, beta, flag, sums ? ws : (float*)nullptr);
    if (sums) {
        const int n = 2 * C;
        hipLaunchKernelGGL(fold_partials_kernel, dim3((n + 31) / 32), dim3(1024), 0, st, ws, g.gy, n, sums, g0, g1,
                           gscale, g_count, 0);
    }
    return LAUNCH_OK();
}

int bn_stats_finalize_launch(const half_t* x, int M, int C, float* sums, float* ws, int64_t ws_floats, float count,
                             const float* gamma, const float* beta, float eps, float momentum, int updates, float* rm,
                             float* rv, float* mean, float* rstd, float* scale, float* shift, long long* nbt,
                             hipStream_t st) {
    if (!ws || ws_floats < 2 * (int64_t)C) return E_WORKSPACE;
    const RowGeom g = row_geometry(M, C, (int)(ws_floats / (2 * (int64_t)C)));
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(g.gx, g.gy), dim3(256), 0, st, x, (const half_t*)nullptr,
                       (half_t*)nullptr, M, C, g.cx_log2, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, ws);
    hipLaunchKernelGGL(fold_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, ws, g.gy, C, sums, count, gamma,
                       beta, eps, momentum, updates, rm, rv, mean, rstd, scale, shift, nbt);
    return LAUNCH_OK();
}
// Statistics rows of a contraction's epilogue (StatEpi): part [rows][2][C] -> sums (+ finalize).  More than 512 rows are
// folded in two stages through `scratch` (FOLD_STAGE_ROWS x 2C floats).
static const float* fold_stage1(const float* part, int& rows, int n, float* scratch, hipStream_t st) {
    if (rows <= 512) return part;        // one 1024-thread block per 32 channels folds 512 rows in 8 four-deep iterations
    const int per = (rows + FOLD_STAGE_ROWS - 1) / FOLD_STAGE_ROWS;
    const int ny = (rows + per - 1) / per;
    hipLaunchKernelGGL(fold_rows_kernel, dim3((n + 31) / 32, ny), dim3(1024), 0, st, part, rows, n, per, scratch,
                       (int64_t)0, (int64_t)0);
    rows = ny;
    return scratch;
}
int bn_fold_finalize_launch(const float* part, int rows, int C, float* scratch, float* sums, float count,
                            const float* gamma, const float* beta, float eps, float momentum, int updates, float* rm,
                            float* rv, float* mean, float* rstd, float* scale, float* shift, long long* nbt,
                            hipStream_t st) {
    const float* src = fold_stage1(part, rows, 2 * C, scratch, st);
    hipLaunchKernelGGL(fold_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, st, src, rows, C, sums, count, gamma,
                       beta, eps, momentum, updates, rm, rv, mean, rstd, scale, shift, nbt);
    return LAUNCH_OK();
}
int bn_fold_launch(const float* part, int rows, int n, float* scratch, float* sums, hipStream_t st) {
    const float* src = fold_stage1(part, rows, n, scratch, st);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((n + 31) / 32), dim3(1024), 0, st, src, rows, n, sums,
                       (float*)nullptr, (float*)nullptr, 0.f, 0, 0);
    return LAUNCH_OK();
}
// part [G][rows_cap][2][C] (the first `rows` rows of each group are valid) -> sums [G][2][C] (+ parameter gradients)
int bn_bwd_fold_launch(const float* part, int rows, int rows_cap, int C, int G, float* scratch, float* sums,
                       float* dbeta, float* dgamma, float gscale, int pgroup, hipStream_t st) {
    const int n = 2 * C;
    int64_t gstride = (int64_t)rows_cap * n;
    const float* src = part;
    if (rows > 512) {
        const int per = (rows + FOLD_STAGE_ROWS - 1) / FOLD_STAGE_ROWS;
        const int ny = (rows + per - 1) / per;
        hipLaunchKernelGGL(fold_rows_kernel, dim3((n + 31) / 32, ny, G), dim3(1024), 0, st, part, rows, n, per, scratch,
                           gstride, (int64_t)FOLD_STAGE_ROWS * n);
        src = scratch;
        rows = ny;
        gstride = (int64_t)FOLD_STAGE_ROWS * n;
    }
    hipLaunchKernelGGL(fold_groups_kernel, dim3((n + 31) / 32, G), dim3(1024), 0, st, src, rows, n, gstride, sums, dbeta,
                       dgamma, gscale, pgroup);
    return LAUNCH_OK();
}
int bn_cols_fwd_launch(const half_t* x, half_t* y, int M, int C, float count, const float* gamma, const float* beta,
                       float eps, float momentum, int updates, float* rm, float* rv, float* mean, float* rstd,
                       float* scale, float* shift, float* sums2C, long long* nbt, int relu, const float* in_scale,
                       hipStream_t st) {
    const int nch = C / 8;
    hipLaunchKernelGGL(bn_cols_fwd_kernel, dim3((nch + COLS_CX - 1) / COLS_CX), dim3(256), 0, st, x, y, M, C, count, gamma,
                       beta, eps, momentum, updates, rm, rv, mean, rstd, scale, shift, sums2C, nbt, relu, in_scale);
    return LAUNCH_OK();
}
int bn_cols_bwd_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, int nstreams, float count,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, float* sums,
                       float* dbeta, float* dgamma, float gscale, int pstream, hipStream_t st) {
    const int nch = C / 8;
    const dim3 grid((nch + COLS_CX - 1) / COLS_CX);
    if (nstreams == 1)
        hipLaunchKernelGGL((bn_cols_bwd_kernel<1>), grid, dim3(256), 0, st, x, dy, dx, M, C, 1.f / count, mean, rstd, gamma,
                           beta, relu, sums, dbeta, dgamma, gscale, pstream);
    else if (nstreams == 2)
        hipLaunchKernelGGL((bn_cols_bwd_kernel<2>), grid, dim3(256), 0, st, x, dy, dx, M, C, 1.f / count, mean, rstd, gamma,
                           beta, relu, sums, dbeta, dgamma, gscale, pstream);
    else
        return E_UNSUPPORTED;
    return LAUNCH_OK();
}
int bn_stats_launch(const half_t* x, int M, int C, float* sums, float* ws, int64_t ws_floats, hipStream_t st) {
    return reduce_launch<0>(x, nullptr, nullptr, M, C, nullptr, nullptr, nullptr, nullptr, 0, sums, ws, ws_floats,
                            st);
}
// column sums of fp16 rows [M][C] (C % 8 == 0) through the statistics reduction: sums2C = [sum x | sum x^2]; dbias (may be
// null): dbias[c] += gscale * sum x[c], c < dbias_n.  The many-row form of colsum_acc_launch (layout.hip).
int colsum_rows_launch(const half_t* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float* dbias,
                       int dbias_n, float gscale, hipStream_t st) {
    return reduce_launch<0>(x, nullptr, nullptr, M, C, nullptr, nullptr, nullptr, nullptr, 0, sums2C, ws, ws_floats, st, dbias,
                            nullptr, gscale, dbias_n);
}
int bn_bwd_reduce_launch(const half_t* x, const half_t* dy, int M, int C, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int relu, float* sums, float* ws, int64_t ws_floats,
                         float* dbeta, float* dgamma, float gscale, hipStream_t st) {
    return reduce_launch<1>(x, dy, nullptr, M, C, mean, rstd, gamma, beta, relu, sums, ws, ws_floats, st, dbeta, dgamma,
                            gscale);
}
// colsum may be null (then no reduction is performed); colsum gets [2][C] (second half unused).  dbias (may be null):
// dbias[c] += gscale * colsum[c], c < dbias_n <= C, in the fold of the partial sums (one writer per element).
int act_bwd_launch(const half_t* y, const half_t* dy, half_t* dpre, int M, int C, int act, float* colsum, float* ws,
                   int64_t ws_floats, float* dbias, int dbias_n, float gscale, hipStream_t st) {
    return reduce_launch<2>(y, dy, dpre, M, C, nullptr, nullptr, nullptr, nullptr, act, colsum, ws, ws_floats, st, dbias,
                            nullptr, gscale, dbias_n);
}
int bn_bwd_reduce2_launch(const half_t* x, const half_t* dy, int M, int C, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int relu, float* sums4C, float* ws, int64_t ws_floats,
                          float* dbeta, float* dgamma, float gscale, int param_stream, hipStream_t st) {
    if (!ws || ws_floats < 4 * (int64_t)C) return E_WORKSPACE;
    const RowGeom g = row_geometry(M, C, (int)(ws_floats / (4 * (int64_t)C)));
    hipLaunchKernelGGL(bn_reduce2_kernel, dim3(g.gx, g.gy), dim3(256), 0, st, x, dy, M, C, g.cx_log2, mean, rstd, gamma,
                       beta, relu, ws);
    const int n = 4 * C;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((n + 31) / 32), dim3(1024), 0, st, ws, g.gy, n, sums4C, dbeta, dgamma,
                       gscale, C, param_stream ? 2 * C : 0);
    return LAUNCH_OK();
}
int bn_finalize_launch(const float* sums, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* rm, float* rv, float* mean, float* rstd, float* scale,
                       float* shift, long long* nbt, const float* in_scale, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, count, gamma, beta, eps,
                       momentum, updates, rm, rv, mean, rstd, scale, shift, nbt, in_scale);
    return LAUNCH_OK();
}
static RowGeom stream_geometry(int M, int C) {
    RowGeom g = row_geometry(M, C, 1 << 30);
    // streaming kernels have no per-block epilogue: allow more blocks for short row loops
    const int RY = 256 >> g.cx_log2;
    int gy = (M + RY * 8 - 1) / (RY * 8);
    int cap = 2048 / g.gx;
    if (cap < 1) cap = 1;
    g.gy = gy > cap ? cap : (gy < 1 ? 1 : gy);
    return g;
}
int bn_apply_launch(const half_t* x, half_t* y, int M, int C, const float* scale, const float* shift, int relu,
                    hipStream_t st) {
    const RowGeom g = stream_geometry(M, C);
    hipLaunchKernelGGL((bn_stream_kernel<0>), dim3(g.gx, g.gy), dim3(256), 0, st, x, (const half_t*)nullptr, y, M, C,
                       g.cx_log2, scale, shift, (const float*)nullptr, (const float*)nullptr, relu, 0.f,
                       (const float*)nullptr);
    return LAUNCH_OK();
}
int bn_bwd_apply2_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, float count, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int relu, const float* sums4C,
                         hipStream_t st) {
    const RowGeom g = stream_geometry(M, C);
    hipLaunchKernelGGL(bn_stream2_kernel, dim3(g.gx, g.gy), dim3(256), 0, st, x, dy, dx, M, C, g.cx_log2, mean, rstd,
                       gamma, beta, relu, 1.f / count, sums4C);
    return LAUNCH_OK();
}
int bn_bwd_apply_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, float count, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int relu, const float* sums,
                        hipStream_t st) {
    const RowGeom g = stream_geometry(M, C);
    hipLaunchKernelGGL((bn_stream_kernel<1>), dim3(g.gx, g.gy), dim3(256), 0, st, x, dy, dx, M, C, g.cx_log2, mean,
                       rstd, gamma, beta, relu, 1.f / count, sums);
    return LAUNCH_OK();
}

}  // namespace fmri
