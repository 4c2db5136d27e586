// On-device evaluation metrics of the reference's validation loop (train/train_utils.py): Pearson correlation
// (PearsonCorrelation.forward, :276-292) and mean SSIM (StructuralSimilarity.forward, :343-420) between a batch of
// reconstructed and ground-truth images, fp32 NCHW in, scalars out, no host round trip.
//
//   PCC  = sum(vx*vy) / (sqrt(sum vx^2) * sqrt(sum vy^2)),  vx = x - mean(x), vy = y - mean(y)   (means over the batch)
//   SSIM = mean over (n, c, y, x) of ((2 mu12 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),
//          mu / s = 11x11 Gaussian (sigma 1.5) local means / (co)variances with zero padding, C1 = 1e-4, C2 = 9e-4.
//
// Both are bandwidth-trivial (a few MB): one pass each, fp64 accumulation of the global sums.
#include "kernels.h"

namespace fmri {

__device__ __forceinline__ double block_sum_256d(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// sums: [0] sum x, [1] sum y, [2] sum x^2, [3] sum y^2, [4] sum x*y
__global__ __launch_bounds__(256) void pcc_sums_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       int64_t n, double* __restrict__ sums) {
    __shared__ double sh[4];
    double s[5] = {0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double a = x[i], b = y[i];
        s[0] += a; s[1] += b; s[2] += a * a; s[3] += b * b; s[4] += a * b;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double t = block_sum_256d(s[k], sh);
        if (threadIdx.x == 0) atomicAdd(sums + k, t);
    }
}

__global__ void pcc_final_kernel(const double* __restrict__ sums, double n, float* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    const double mx = sums[0] / n, my = sums[1] / n;
    const double sxx = sums[2] - n * mx * mx, syy = sums[3] - n * my * my, sxy = sums[4] - n * mx * my;
    *out = (float)(sxy / (sqrt(sxx) * sqrt(syy)));
}

// One block = one 16x16 output tile of one (image, channel) plane.  acc: [0] sum of ssim, [1] sum of the contrast term
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H,
                                                   int W, int ws, double* __restrict__ acc) {
    constexpr int TS = 16, MAXW = 11, R = TS + MAXW - 1;      // 26
    __shared__ float ta[R][R + 1], tb[R][R + 1];
    __shared__ float hx[5][R][TS + 1];
    __shared__ float g[MAXW];
    __shared__ double sh[4];
    const int pad = 5;                                         // the reference pads with window_size // 2 = 5 always
    const int tx0 = blockIdx.x * TS, ty0 = blockIdx.y * TS;
    const float* pa = a + (int64_t)blockIdx.z * H * W;
    const float* pb = b + (int64_t)blockIdx.z * H * W;
    if (threadIdx.x < MAXW) {
        // gaussian(ws, 1.5): exp(-(x - ws//2)^2 / (2*1.5^2)) normalised to sum 1 (fp32 like the reference)
        float s = 0.f;
        for (int i = 0; i < ws; ++i) s += expf(-(float)((i - ws / 2) * (i - ws / 2)) / 4.5f);
        const int i = threadIdx.x;
        g[i] = i < ws ? expf(-(float)((i - ws / 2) * (i - ws / 2)) / 4.5f) / s : 0.f;
    }
    const int span = TS + ws - 1;
    for (int e = threadIdx.x; e < span * span; e += 256) {
        const int j = e / span, i = e - j * span;
        const int y = ty0 - pad + j, x = tx0 - pad + i;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        ta[j][i] = ok ? pa[(int64_t)y * W + x] : 0.f;
        tb[j][i] = ok ? pb[(int64_t)y * W + x] : 0.f;
    }
    __syncthreads();
    // horizontal pass: 5 filtered quantities for span rows x 16 columns
    for (int e = threadIdx.x; e < span * TS; e += 256) {
        const int j = e / TS, i = e - j * TS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int k = 0; k < ws; ++k) {
            const float w = g[k], u = ta[j][i + k], v = tb[j][i + k];
            s0 += w * u; s1 += w * v; s2 += w * u * u; s3 += w * v * v; s4 += w * u * v;
        }
        hx[0][j][i] = s0; hx[1][j][i] = s1; hx[2][j][i] = s2; hx[3][j][i] = s3; hx[4][j][i] = s4;
    }
    __syncthreads();
    const int oy = threadIdx.x >> 4, ox = threadIdx.x & 15;
    double ss = 0.0, cs = 0.0;
    if (ty0 + oy < H && tx0 + ox < W) {
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
        for (int k = 0; k < ws; ++k) {
            const float w = g[k];
            m1 += w * hx[0][oy + k][ox]; m2 += w * hx[1][oy + k][ox];
            e11 += w * hx[2][oy + k][ox]; e22 += w * hx[3][oy + k][ox]; e12 += w * hx[4][oy + k][ox];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float s1 = e11 - m11, s2 = e22 - m22, s12 = e12 - m12;
        ss = (double)(((2.f * m12 + C1) * (2.f * s12 + C2)) / ((m11 + m22 + C1) * (s1 + s2 + C2)));
        cs = (double)((2.f * s12 + C2) / (s1 + s2 + C2));
    }
    const double t0 = block_sum_256d(ss, sh);
    const double t1 = block_sum_256d(cs, sh);
    if (threadIdx.x == 0) { atomicAdd(acc, t0); atomicAdd(acc + 1, t1); }
}

__global__ void ssim_final_kernel(const double* __restrict__ acc, double count, float* __restrict__ ssim,
                                  float* __restrict__ contrast) {
    if (threadIdx.x || blockIdx.x) return;
    if (ssim) *ssim = (float)(acc[0] / count);
    if (contrast) *contrast = (float)(acc[1] / count);
}

int pcc_launch(const float* x, const float* y, int64_t n, double* sums5, float* out, hipStream_t st) {
    (void)hipMemsetAsync(sums5, 0, 5 * sizeof(double), st);
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1 || g_deterministic) blocks = 1;        // one block: fixed-order sums
    hipLaunchKernelGGL(pcc_sums_kernel, dim3(blocks), dim3(256), 0, st, x, y, n, sums5);
    hipLaunchKernelGGL(pcc_final_kernel, dim3(1), dim3(64), 0, st, sums5, (double)n, out);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

int ssim_launch(const float* a, const float* b, int planes, int H, int W, double* acc2, float* ssim, float* contrast,
                hipStream_t st) {
    (void)hipMemsetAsync(acc2, 0, 2 * sizeof(double), st);
    if (H < 11 || W < 11) return E_UNSUPPORTED;   // the reference's window shrinks but its padding does not
    const int ws = 11;
    hipLaunchKernelGGL(ssim_kernel, dim3((W + 15) / 16, (H + 15) / 16, planes), dim3(256), 0, st, a, b, H, W, ws, acc2);
    hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(64), 0, st, acc2, (double)planes * H * W, ssim, contrast);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
