// Batch-norm (train mode) kernels over NHWC fp16 rows [M][C]  (BatchNorm2d and BatchNorm1d alike).
//
// Replaces nn.BatchNorm2d / nn.BatchNorm1d(momentum=0.9) + ReLU of the reference blocks
// (models/vae_gan.py:21,28-29,54,58-59,81,108,158) and their autograd backward:
//   stats    : per-channel sum / sum-of-squares (fp32 atomics into [2][C])           -- HBM bound
//   finalize : mean, rstd, scale/shift, running-stat update (momentum 0.9, unbiased running var,
//              `updates` consecutive updates for the discriminator's REC+GAN double pass)
//   apply    : y = relu(x*scale + shift)                                               -- HBM bound
//   bwd_reduce / bwd_apply : dgamma, dbeta and dx through ReLU + BN (batch statistics)
// The sums are kept outside the kernels so that a data-parallel run can all-reduce them (SyncBN).
#include "kernels.h"

namespace fmri {

// 2-D thread block: CX chunk-columns (8 channels each) x RY row lanes, CX*RY = 256.
template <int MODE>  // 0: stats (sum x, sum x^2); 1: bwd reduce (sum g, sum g*xhat)
__global__ __launch_bounds__(256) void bn_reduce_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                        int M, int C, int cx_log2, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu,
                                                        float* __restrict__ out /* [2][C] */) {
    __shared__ float red[256 * 16];
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
        for (int m = blockIdx.y * RY + ry; m < M; m += gridDim.y * RY) {
            const h8 xv = *(const h8*)(x + (int64_t)m * C + chunk * 8);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)xv[j]; s0[j] += f; s1[j] += f * f; }
            } else {
                const h8 gv = *(const h8*)(dy + (int64_t)m * C + chunk * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    float g = (float)gv[j];
                    if (relu && !(xh * ga[j] + be[j] > 0.f)) g = 0.f;
                    s0[j] += g; s1[j] += g * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s0[j]; red[threadIdx.x * 16 + 8 + j] = s1[j]; }
    __syncthreads();
    // threads with ry == 0 finish their column: 16 values summed over RY lanes
    if (ry == 0 && chunk < nch) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = 0.f;
            for (int r = 0; r < RY; ++r) s += red[((r << cx_log2) + cx) * 16 + j];
            const int c = chunk * 8 + (j & 7);
            atomicAdd(out + (j >> 3) * C + c, s);
        }
    }
}

// one thread per channel
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, int updates,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = sums[c] / count;
    float var = sums[C + c] / count - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + eps);
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    const float sc = gamma[c] * rstd;
    scale_out[c] = sc;
    shift_out[c] = beta[c] - mean * sc;
    if (running_mean && updates > 0) {
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * mean;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// y = act(x*scale + shift), flat over [M][C]
__global__ void bn_apply_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int64_t nchunks_total, int nch,
                                const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks_total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const h8 xv = *(const h8*)(x + i * 8);
        h8 yv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)xv[j] * scale[ch * 8 + j] + shift[ch * 8 + j];
            if (relu) f = f > 0.f ? f : 0.f;
            yv[j] = (half_t)f;
        }
        *(h8*)(y + i * 8) = yv;
    }
}

// dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M), g = dy*relu_mask
__global__ void bn_bwd_apply_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                    half_t* __restrict__ dx, int64_t nchunks_total, int nch, float inv_count,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                    const float* __restrict__ sums /* [2][C] */, int C) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks_total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const h8 xv = *(const h8*)(x + i * 8);
        const h8 gv = *(const h8*)(dy + i * 8);
        h8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            const float xh = ((float)xv[j] - mean[c]) * rstd[c];
            float g = (float)gv[j];
            if (relu && !(xh * gamma[c] + beta[c] > 0.f)) g = 0.f;
            const float v = gamma[c] * rstd[c] * (g - sums[c] * inv_count - xh * sums[C + c] * inv_count);
            ov[j] = (half_t)v;
        }
        *(h8*)(dx + i * 8) = ov;
    }
}

// activation backward for (bias + act) layers without BN: dpre = dy * act'(y); optional column sums
// (bias gradient) into colsum[C] via atomics.  act: ReLU (mask y>0) or tanh (1-y^2).
__global__ __launch_bounds__(256) void act_bwd_kernel(const half_t* __restrict__ y, const half_t* __restrict__ dy,
                                                      half_t* __restrict__ dpre, int M, int C, int cx_log2, int act,
                                                      float* __restrict__ colsum) {
    __shared__ float red[256 * 8];
    const int CX = 1 << cx_log2;
    const int RY = 256 >> cx_log2;
    const int cx = threadIdx.x & (CX - 1);
    const int ry = threadIdx.x >> cx_log2;
    const int chunk = blockIdx.x * CX + cx;
    const int nch = C >> 3;
    float s0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = 0.f;
    if (chunk < nch) {
        for (int m = blockIdx.y * RY + ry; m < M; m += gridDim.y * RY) {
            const int64_t off = (int64_t)m * C + chunk * 8;
            const h8 yv = *(const h8*)(y + off);
            const h8 gv = *(const h8*)(dy + off);
            h8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float yy = (float)yv[j];
                float g = (float)gv[j];
                if (act == ACT_RELU) g = yy > 0.f ? g : 0.f;
                else if (act == ACT_TANH) g = g * (1.f - yy * yy);
                ov[j] = (half_t)g;
                s0[j] += g;
            }
            *(h8*)(dpre + off) = ov;
        }
    }
    if (colsum) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = s0[j];
        __syncthreads();
        if (ry == 0 && chunk < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.f;
                for (int r = 0; r < RY; ++r) s += red[((r << cx_log2) + cx) * 8 + j];
                atomicAdd(colsum + chunk * 8 + j, s);
            }
        }
    }
}

static void reduce_geometry(int M, int C, int& cx_log2, dim3& grid) {
    const int nch = C / 8;
    cx_log2 = 0;
    while ((1 << cx_log2) < nch && cx_log2 < 8) ++cx_log2;
    const int CX = 1 << cx_log2, RY = 256 >> cx_log2;
    const int gx = (nch + CX - 1) / CX;
    int gy = (M + RY * 8 - 1) / (RY * 8);            // >= 8 rows per thread
    const int cap = 2048 / (gx > 0 ? gx : 1);
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    grid = dim3(gx, gy);
}

int bn_stats_launch(const half_t* x, int M, int C, float* sums, hipStream_t st) {
    int cxl; dim3 grid;
    reduce_geometry(M, C, cxl, grid);
    hipLaunchKernelGGL((bn_reduce_kernel<0>), grid, dim3(256), 0, st, x, (const half_t*)nullptr, M, C, cxl,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
                       sums);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int bn_bwd_reduce_launch(const half_t* x, const half_t* dy, int M, int C, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int relu, float* sums, hipStream_t st) {
    int cxl; dim3 grid;
    reduce_geometry(M, C, cxl, grid);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), grid, dim3(256), 0, st, x, dy, M, C, cxl, mean, rstd, gamma, beta, relu,
                       sums);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int bn_finalize_launch(const float* sums, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* rm, float* rv, float* mean, float* rstd, float* scale,
                       float* shift, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, count, gamma, beta, eps,
                       momentum, updates, rm, rv, mean, rstd, scale, shift);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
static inline int nblk(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
int bn_apply_launch(const half_t* x, half_t* y, int M, int C, const float* scale, const float* shift, int relu,
                    hipStream_t st) {
    const int64_t n = (int64_t)M * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(n)), dim3(256), 0, st, x, y, n, C / 8, scale, shift, relu);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int bn_bwd_apply_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, float count, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int relu, const float* sums,
                        hipStream_t st) {
    const int64_t n = (int64_t)M * (C / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(n)), dim3(256), 0, st, x, dy, dx, n, C / 8, 1.f / count, mean,
                       rstd, gamma, beta, relu, sums, C);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}
int act_bwd_launch(const half_t* y, const half_t* dy, half_t* dpre, int M, int C, int act, float* colsum,
                   hipStream_t st) {
    int cxl; dim3 grid;
    reduce_geometry(M, C, cxl, grid);
    hipLaunchKernelGGL(act_bwd_kernel, grid, dim3(256), 0, st, y, dy, dpre, M, C, cxl, act, colsum);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
