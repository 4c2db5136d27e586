// The WAE latent discriminator -- Linear(z, H) ReLU [Linear(H, H) ReLU] x 3, Linear(H, 1) -- as ONE launch forward and
// ONE launch for the backward chain (gfx950).
//
// Replaces (reference models/vae_gan.py:499-529, WaeDiscriminator.main / forward before the sigmoid) and its autograd
// data path: train/train_wae_stage1.py:278-288 (D phase: forward over [real ; fake], backward), :296-303 (penalty:
// forward, gradient w.r.t. z).  With 1.7 MFLOP per row the layer-by-layer engine path is ~50 launches of 3-8 us per
// phase; here a block owns 32 rows for the whole network:
//   * activations of the block's rows live in LDS (two ping-pong tiles of 32 x H fp16), every layer's output is
//     written there and (hidden activations, for the backward pass) to HBM;
//   * wave w computes output features [w H/4, (w+1) H/4): its weight fragments are its own, so they go global -> register
//     (16-byte rows of the packed fp16 matrices the layer-wise path uses: same rounding), prefetched one K-step ahead;
//     MFMA 16x16x32 f16 with the weights as the A operand, fp32 accumulation;
//   * backward chain: delta4 = dlogit * W4 masked by h4 > 0, delta_j = (delta_{j+1} . W_j) masked by h_j > 0 through the
//     data-gradient orientation of the same packed weights, the bias gradients (column sums) by atomics, optionally
//     dz = delta1 . W0 in fp32.  The weight gradients (reductions over ALL rows) stay with fmri_wgrad on the side stream.
#include "kernels.h"

namespace fmri {

namespace {

constexpr int MLP_RB = 32;                      // rows per block

// acc[tn][tm] += W[o0 + 16 tn + ..][k] * X[16 tm + ..][k] over k < K.  W: global fp16 rows of kp elements; X: LDS rows
// of `pitch` halves.  TN 16-row weight tiles per wave, 2 row tiles.
template <int TN>
__device__ __forceinline__ void mlp_gemm(const half_t* __restrict__ W, int kp, const half_t* X, int pitch, int K,
                                         int lane, f4 (&acc)[TN][2]) {
    const int frow = lane & 15, fq = lane >> 4;
    const half_t* wl = W + (int64_t)frow * kp + fq * 8;
    const half_t* xl = X + frow * pitch + fq * 8;
    h8 wc[TN], wn[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) wc[tn] = *(const h8*)(wl + (int64_t)tn * 16 * kp);
    for (int k0 = 0; k0 < K; k0 += 32) {
        const bool more = k0 + 32 < K;
        if (more) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wn[tn] = *(const h8*)(wl + (int64_t)tn * 16 * kp + k0 + 32);
        }
        h8 xf[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) xf[tm] = *(const h8*)(xl + tm * 16 * pitch + k0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[tn], xf[tm], acc[tn][tm], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wc[tn] = wn[tn];
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace

template <int H>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpFwdArgs a) {
    constexpr int PITCH = H + 8;                 // halves: 16-byte aligned rows, 4-bank rotation per row
    constexpr int TN = H / 64;                   // 16-feature tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* const buf0 = (half_t*)smem;
    half_t* const buf1 = (half_t*)smem + MLP_RB * PITCH;
    auto bufp = [&](int i) { return (i & 1) ? buf1 : buf0; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * MLP_RB;
    const int frow = lane & 15, fq = lane >> 4;

    // input rows (zero beyond M)
    for (int u = tid; u < MLP_RB * (a.Zp / 8); u += 256) {
        const int r = u / (a.Zp / 8), c = u - r * (a.Zp / 8);
        h8 v = (h8)(half_t)0.f;
        if (row0 + r < a.M) v = *(const h8*)(a.z + (int64_t)(row0 + r) * a.Zp + c * 8);
        *(h8*)(buf0 + r * PITCH + c * 8) = v;
    }
    __syncthreads();

    const int o0 = wave * (H / 4);
    for (int L = 0; L < 4; ++L) {
        const half_t* in = bufp(L);
        half_t* out = bufp(L + 1);
        const int K = L == 0 ? a.Zp : H;
        f4 acc[TN][2];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        mlp_gemm<TN>(a.w[L] + (int64_t)o0 * a.kp[L], a.kp[L], in, PITCH, K, lane, acc);
        half_t* hg = a.hs[L];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int co = o0 + tn * 16 + fq * 4;
            const f4 b = a.bias[L] ? *(const f4*)(a.bias[L] + co) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int r = tm * 16 + frow;
                h4 hv;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float f = acc[tn][tm][rg] + b[rg];
                    hv[rg] = (half_t)(f > 0.f ? f : 0.f);
                }
                *(h4*)(out + r * PITCH + co) = hv;
                if (row0 + r < a.M) *(h4*)(hg + (int64_t)(row0 + r) * H + co) = hv;
            }
        }
        __syncthreads();
    }
    // output layer: logit[r] = b4 + h4[r] . W4[0]
    const half_t* h4t = buf0;
    const half_t* w4 = a.w[4];
    const float b4 = a.bias[4] ? a.bias[4][0] : 0.f;
    for (int r = wave * (MLP_RB / 4); r < (wave + 1) * (MLP_RB / 4); ++r) {
        float s = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const h8 x = *(const h8*)(h4t + r * PITCH + c);
            const h8 w = *(const h8*)(w4 + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)x[j] * (float)w[j];
        }
        s = wave_sum(s);
        if (lane == 0 && row0 + r < a.M) a.logit[row0 + r] = s + b4;
    }
}

template <int H>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const MlpBwdArgs a) {
    constexpr int PITCH = H + 8;
    constexpr int TN = H / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* const buf0 = (half_t*)smem;
    half_t* const buf1 = (half_t*)smem + MLP_RB * PITCH;
    auto bufp = [&](int i) { return (i & 1) ? buf1 : buf0; };
    __shared__ float dl[MLP_RB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * MLP_RB;
    const int frow = lane & 15, fq = lane >> 4;

    if (tid < MLP_RB) dl[tid] = row0 + tid < a.M ? (float)a.dlogit[(int64_t)(row0 + tid) * a.ldl] : 0.f;
    __syncthreads();
    if (a.dbias[4] && wave == 0) {
        const float s = wave_sum(lane < MLP_RB ? dl[lane] : 0.f);
        if (lane == 0) atomicAdd(a.dbias[4], s * a.inv_scale);
    }
    // delta4 = dlogit * W4 masked by h4 > 0
    for (int u = tid; u < MLP_RB * (H / 8); u += 256) {
        const int r = u / (H / 8), c = (u - r * (H / 8)) * 8;
        h8 v = (h8)(half_t)0.f;
        if (row0 + r < a.M) {
            const h8 w = *(const h8*)(a.w4 + c);
            const h8 hh = *(const h8*)(a.hs[3] + (int64_t)(row0 + r) * H + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)hh[j] > 0.f ? dl[r] * (float)w[j] : 0.f);
            *(h8*)(a.delta[3] + (int64_t)(row0 + r) * H + c) = v;
        }
        *(h8*)(buf0 + r * PITCH + c) = v;
    }
    __syncthreads();

    // column sums of the delta tile in `t` -> bias gradient of layer `layer`
    auto bias_sums = [&](const half_t* t, int layer) {
        if (!a.dbias[layer]) return;
        for (int c = tid; c < H; c += 256) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < MLP_RB; ++r) s += (float)t[r * PITCH + c];
            atomicAdd(a.dbias[layer] + c, s * a.inv_scale);
        }
    };
    bias_sums(buf0, 3);

    const int o0 = wave * (H / 4);
    int cur = 0;
    for (int L = 3; L >= 1; --L) {              // delta_L = (delta_{L+1} . W_L) masked by h_L > 0
        const half_t* in = bufp(cur);
        half_t* out = bufp(cur ^ 1);
        f4 acc[TN][2];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        mlp_gemm<TN>(a.wd[L] + (int64_t)o0 * a.kpd[L], a.kpd[L], in, PITCH, H, lane, acc);
        const half_t* hg = a.hs[L - 1];
        half_t* dg = a.delta[L - 1];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int ci = o0 + tn * 16 + fq * 4;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int r = tm * 16 + frow;
                h4 hv = (h4)(half_t)0.f;
                if (row0 + r < a.M) {
                    const h4 hh = *(const h4*)(hg + (int64_t)(row0 + r) * H + ci);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)((float)hh[rg] > 0.f ? acc[tn][tm][rg] : 0.f);
                    *(h4*)(dg + (int64_t)(row0 + r) * H + ci) = hv;
                }
                *(h4*)(out + r * PITCH + ci) = hv;
            }
        }
        __syncthreads();
        cur ^= 1;
        bias_sums(bufp(cur), L - 1);
    }
    // dz = delta1 . W0 (fp32, true scale): wave w owns z features [w Zp/4, (w+1) Zp/4), Zp <= 256
    if (a.dz) {
        const int zt = a.Zp / 64;                // 16-feature tiles per wave: 1 .. 4
        const int z0 = wave * (a.Zp / 4);
        for (int t = 0; t < zt; ++t) {
            f4 acc[1][2];
            acc[0][0] = acc[0][1] = (f4){0.f, 0.f, 0.f, 0.f};
            mlp_gemm<1>(a.wd[0] + (int64_t)(z0 + t * 16) * a.kpd[0], a.kpd[0], bufp(cur), PITCH, H, lane, acc);
            const int zi = z0 + t * 16 + fq * 4;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int r = tm * 16 + frow;
                if (row0 + r >= a.M) continue;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (zi + rg < a.Z) a.dz[(int64_t)(row0 + r) * a.Z + zi + rg] = acc[0][tm][rg] * a.inv_scale;
            }
        }
    }
}

int mlp_fwd_launch(const MlpFwdArgs& a, hipStream_t st) {
    if (a.H != 512 || a.Zp < 64 || a.Zp > 512 || (a.Zp & 63) || a.M < 1) return E_UNSUPPORTED;
    auto kern = mlp_fwd_kernel<512>;
    constexpr int lds = 2 * MLP_RB * (512 + 8) * 2;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.M + MLP_RB - 1) / MLP_RB), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

int mlp_bwd_launch(const MlpBwdArgs& a, hipStream_t st) {
    if (a.H != 512 || a.Zp < 64 || a.Zp > 256 || (a.Zp & 63) || a.M < 1) return E_UNSUPPORTED;
    auto kern = mlp_bwd_kernel<512>;
    constexpr int lds = 2 * MLP_RB * (512 + 8) * 2;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.M + MLP_RB - 1) / MLP_RB), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
