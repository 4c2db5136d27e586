// Latent / loss / optimizer kernels of the VAE/GAN step (wavefront reductions, fp32).
//
// Restates (reference):
//   reparameterize           models/vae_gan.py:266-269     z = eps*exp(0.5*logvar) + mu
//   VaeGan.loss              models/vae_gan.py:302-320     nle, kl, feature mse, 3 x bce (eps 1e-3 in the log)
//   loss composition + gate  train/train_vgan_stage1.py:368-404
//   RMSprop / Adam updates   train/train_vgan_stage1.py:275-283, train/train_wae_stage1.py:221-224
//
// fp16 cotangents: the two back-propagated streams start from very different, training-dependent
// magnitudes (d bce/d logit vs. d mse/d feature).  Each stream is therefore normalised ON THE DEVICE to
// unit RMS at its starting point (norm factors nA, nB in the scalar block, derived from all-reduced sums so
// every data-parallel rank uses the same factor); everything downstream is linear, so the factor is divided
// out again by the fused optimizer kernels.  No host synchronisation is involved.
#include <cstdlib>

#include "kernels.h"

namespace fmri {

// scalar block layout (floats) -- keep in sync with fmri_hip/steps.py
enum Slot { S_BCE_O = 0, S_BCE_P = 1, S_BCE_S = 2, S_KL = 3, S_MSE = 4, S_NLE = 5, S_LENC = 6, S_LDIS = 7,
            S_LDEC = 8, S_DL2 = 9, S_NA = 10, S_NB = 11, S_RATIO = 12, S_ONE = 13,
            // [14], [15]: encoder-stream re-normalisation (fmri_sumsq / fmri_renorm)
            S_NP = 16,      // 1 / rms(d nle / d x_tilde)                          (modes 'vae', 'dcgan')
            S_C1 = 17,      // lambda * nA / nB : weight of the feature stream in the decoder cotangent ('vae-gan')
            S_C2 = 18,      // 1 - lambda       : weight of the discriminator stream in the decoder cotangent
            S_C3 = 19,      // lambda * nA      : weight of d nle in the decoder cotangent ('dcgan')
            S_GDEC = 20,    // factor the decoder gradients carry ('vae': nP / lambda; else nA)
            S_KLW = 21 };   // weight of the KL term in the encoder loss (beta / batch for 'beta-vae', else 1)

// loss compositions of train/train_vgan_stage1.py:359-388
enum Mode { M_VAEGAN = 0, M_BETAVAE = 1, M_DCGAN = 2, M_VAE = 3 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// ---- latent: heads (mu | logvar, fp32 [B][2z]) + eps -> z (fp16, padded), per-sample KL, total KL
// one wave per sample row (grid-stride over the rows); a block adds the sum of its rows to *kl_total ONCE, its four
// waves' sums in a fixed order -- launched as one block (deterministic mode) the total is a fixed-order sum.
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                                         int B, int Z, int zp, half_t* __restrict__ z16,
                                                         float* __restrict__ kl_rows, float* __restrict__ kl_total,
                                                         int sample /* 0: z = mu (WAE) */) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float klw = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < B; row += gridDim.x * 4) {
        float kl = 0.f;
        for (int j = lane; j < zp; j += 64) {
            float zz = 0.f;
            if (j < Z) {
                const float mu = head[(int64_t)row * 2 * Z + j];
                const float lv = head[(int64_t)row * 2 * Z + Z + j];
                zz = sample ? eps[(int64_t)row * Z + j] * __expf(0.5f * lv) + mu : mu;
                kl += -0.5f * (-__expf(lv) - mu * mu + lv + 1.f);
            }
            z16[(int64_t)row * zp + j] = (half_t)zz;
        }
        kl = wave_sum(kl);
        if (lane == 0 && kl_rows) kl_rows[row] = kl;
        klw += kl;
    }
    if (lane == 0) sh[wave] = klw;
    __syncthreads();
    if (threadIdx.x == 0 && kl_total) atomicAdd(kl_total, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// ---- range-safe latent (fmri_latent_fwd_ranged).  sigma = exp(0.5 logvar) leaves fp16's range at logvar > 22.2, which
// an fp32 run survives (the reference's arithmetic overflows at logvar > 88.7) and a training run does reach: one
// outlier row of a BatchNorm1d batch is enough.  The fp16 rows the decoder's first GEMM reads are therefore
// stored as s * z with s = 2^-k the largest power of two <= 1 that brings max |z| of the batch under `cap` -- s = 1, and
// bit-identical rows, whenever max |z| <= cap.  The BatchNorm1d behind that GEMM is invariant under the scaling once its
// eps is scaled by s^2 (bn_finalize_channel), the weight gradient is exact (cotangent / s times input * s) and the data
// gradient is multiplied by s.
//   phase 1: z (fp32, [B][Z]) -> z32, per-sample / total KL, *zmax = max(*zmax, max |z|) (bit pattern of a float >= 0)
//   phase 2: s from *zmax -> *zscale; z16 = z32 * s (padding columns zero)
__global__ __launch_bounds__(256) void latent_z32_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                                         int B, int Z, float* __restrict__ z32,
                                                         float* __restrict__ kl_rows, float* __restrict__ kl_total,
                                                         int sample, float* __restrict__ zmax) {
    __shared__ float sh[4], shm[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float klw = 0.f, mx = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < B; row += gridDim.x * 4) {
        float kl = 0.f;
        for (int j = lane; j < Z; j += 64) {
            const float mu = head[(int64_t)row * 2 * Z + j];
            const float lv = head[(int64_t)row * 2 * Z + Z + j];
            const float zz = sample ? eps[(int64_t)row * Z + j] * __expf(0.5f * lv) + mu : mu;
            kl += -0.5f * (-__expf(lv) - mu * mu + lv + 1.f);
            z32[(int64_t)row * Z + j] = zz;
            mx = fmaxf(mx, fabsf(zz));           // (a NaN is dropped here and reaches z16 through phase 2)
        }
        kl = wave_sum(kl);
        if (lane == 0 && kl_rows) kl_rows[row] = kl;
        klw += kl;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) { sh[wave] = klw; shm[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (kl_total) atomicAdd(kl_total, (sh[0] + sh[1]) + (sh[2] + sh[3]));
        atomicMax((unsigned int*)zmax, __float_as_uint(fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]))));
    }
}

__global__ __launch_bounds__(256) void rows_absmax_kernel(const float* __restrict__ x, int64_t n,
                                                          float* __restrict__ zmax) {
    __shared__ float shm[4];
    float mx = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mx = fmaxf(mx, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax((unsigned int*)zmax, __float_as_uint(fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]))));
}

__host__ __device__ __forceinline__ float range_scale(float m, float cap) {
    // largest power of two s <= 1 with m * s <= cap; 1 for a non-finite maximum (the rows then carry inf / NaN on, as
    // the reference's arithmetic does) -- at most 2^-60, so that eps * s^2 stays a normal fp32 number
    if (!(m > cap) || m > 3.0e38f) return 1.f;
    int e;
    const float f = frexpf(m / cap, &e);         // m / cap = f * 2^e, f in [0.5, 1)  ->  m * 2^-e <= cap
    if (f == 0.5f) --e;                          // an exact power of two: m * 2^-(e-1) == cap
    return ldexpf(1.f, -(e > 60 ? 60 : e));
}

__global__ __launch_bounds__(256) void latent_pack_kernel(const float* __restrict__ z32, int B, int Z, int zp,
                                                          half_t* __restrict__ z16, const float* __restrict__ zmax,
                                                          float* __restrict__ zscale, float cap) {
    const float s = range_scale(*zmax, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) *zscale = s;
    const int64_t total = (int64_t)B * zp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % zp);
        const int64_t row = i / zp;
        z16[i] = (half_t)(j < Z ? z32[row * Z + j] * s : 0.f);
    }
}

// backward: dhead[row] = ([ g + w*mu | g*eps*0.5*exp(0.5 lv) + w*0.5*(exp(lv)-1) ]) * out_scale
//   g = dz * dz_unscale (dz may be null), w = kl_w * (*kl_dev) (kl_dev may be null -> 1): when dz carries a
//   device normalisation factor n the KL term is multiplied by the same n so the sum stays consistent.
__global__ void latent_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                  const float* __restrict__ dz, int ldz, float dz_unscale, float kl_w,
                                  const float* __restrict__ kl_dev, int B, int Z, float out_scale,
                                  half_t* __restrict__ dhead16, float* __restrict__ dhead32, int sample) {
    const int64_t total = (int64_t)B * Z;
    const float w = kl_w * (kl_dev ? *kl_dev : 1.f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Z);
        const int64_t row = i / Z;
        const float mu = head[row * 2 * Z + j];
        const float lv = head[row * 2 * Z + Z + j];
        const float g = dz ? dz[row * ldz + j] * dz_unscale : 0.f;
        const float dmu = g + w * mu;
        float dlv = w * 0.5f * (__expf(lv) - 1.f);
        if (sample) dlv += g * eps[row * Z + j] * 0.5f * __expf(0.5f * lv);
        if (dhead16) {
            dhead16[row * 2 * Z + j] = (half_t)(dmu * out_scale);
            dhead16[row * 2 * Z + Z + j] = (half_t)(dlv * out_scale);
        }
        if (dhead32) {
            dhead32[row * 2 * Z + j] = dmu;
            dhead32[row * 2 * Z + Z + j] = dlv;
        }
    }
}

// ---- feature-matching term: mse_b = sum_f 0.5 (f_o - f_p)^2 over the raw conv-3 features.  A block takes samples
// blockIdx.x, blockIdx.x + gridDim.x, ... and adds the sum of its samples to *mse_total once (one block = deterministic)
__global__ __launch_bounds__(256) void feat_mse_kernel(const half_t* __restrict__ feat, int B, int F,
                                                       float* __restrict__ mse_rows, float* __restrict__ mse_total) {
    __shared__ float sh[4];
    float tot = 0.f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const half_t* fo = feat + (int64_t)b * F;
        const half_t* fp = feat + (int64_t)(B + b) * F;
        float s = 0.f;
        for (int i = threadIdx.x * 8; i < F; i += 256 * 8) {
            const h8 o = *(const h8*)(fo + i);
            const h8 p = *(const h8*)(fp + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float df = (float)o[j] - (float)p[j];
                s += 0.5f * df * df;
            }
        }
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0 && mse_rows) mse_rows[b] = s;
        tot += s;
    }
    if (threadIdx.x == 0 && mse_total) atomicAdd(mse_total, tot);
}

// cotangent of sum(mse) w.r.t. the 3B feature rows: orig +d, pred -d, sampled 0; times gscale * (*norm)
__global__ __launch_bounds__(256) void feat_mse_bwd_kernel(const half_t* __restrict__ feat, int B, int F,
                                                           half_t* __restrict__ dfeat, float gscale,
                                                           const float* __restrict__ norm) {
    const int b = blockIdx.x;
    const float sc = gscale * (norm ? *norm : 1.f);
    const half_t* fo = feat + (int64_t)b * F;
    const half_t* fp = feat + (int64_t)(B + b) * F;
    for (int i = threadIdx.x * 8; i < F; i += 256 * 8) {
        const h8 o = *(const h8*)(fo + i);
        const h8 p = *(const h8*)(fp + i);
        h8 d, nd, z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float df = ((float)o[j] - (float)p[j]) * sc;
            d[j] = (half_t)df;
            nd[j] = (half_t)(-df);
            z[j] = (half_t)0.f;
        }
        *(h8*)(dfeat + (int64_t)b * F + i) = d;
        *(h8*)(dfeat + (int64_t)(B + b) * F + i) = nd;
        *(h8*)(dfeat + (int64_t)(2 * B + b) * F + i) = z;
    }
}

// ---- pixel term: nle_total = sum 0.5 (x - x_tilde)^2 over real channels of NHWC-padded images;
// optional cotangent d nle / d x_tilde = -(x - x_tilde) * gscale.  Cp == 8 (every image tensor of the engine): one
// 16-byte load per pixel and operand.
__global__ __launch_bounds__(256) void pixel_sq_kernel(const half_t* __restrict__ x, const half_t* __restrict__ xt,
                                                       int64_t npix, int C, int Cp, float* __restrict__ total,
                                                       half_t* __restrict__ dxt, float gscale) {
    __shared__ float sh[4];
    float s = 0.f;
    if (Cp == 8) {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix;
             i += (int64_t)gridDim.x * blockDim.x) {
            const h8 a = *(const h8*)(x + i * 8);
            const h8 b = *(const h8*)(xt + i * 8);
            h8 d;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float df = 0.f;
                if (c < C) {
                    df = (float)a[c] - (float)b[c];
                    s += 0.5f * df * df;
                }
                d[c] = (half_t)(-df * gscale);
            }
            if (dxt) *(h8*)(dxt + i * 8) = d;
        }
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix;
             i += (int64_t)gridDim.x * blockDim.x) {
            for (int c = 0; c < Cp; ++c) {
                float df = 0.f;
                if (c < C) {
                    df = (float)x[i * Cp + c] - (float)xt[i * Cp + c];
                    s += 0.5f * df * df;
                }
                if (dxt) dxt[i * Cp + c] = (half_t)(-df * gscale);
            }
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0 && total) atomicAdd(total, s);
}

__device__ __forceinline__ void gan_terms(float l, int part, float& p, float& bce, float& dl) {
    p = 1.f / (1.f + expf(-l));
    if (part == 0) {
        bce = -logf(p + 1e-3f);
        dl = -p * (1.f - p) / (p + 1e-3f);
    } else {
        bce = -logf(1.f - p + 1e-3f);
        dl = p * (1.f - p) / (1.f - p + 1e-3f);
    }
}

// ---- discriminator class head: logits (fp32 [3B], bias already added) -> sigmoid, the three BCE sums
// (eps 1e-3 inside the log, models/vae_gan.py:316-318) and sum of squared d(bce)/d(logit) (for the stream norm).
// scal: [S_BCE_O..S_BCE_S] += bce sums, [S_DL2] += sum dl^2
// parts: bit p set = part p (0 orig, 1 pred, 2 sampled) belongs to the discriminator loss that is back-propagated
// (dcgan / vae: orig + sampled, train_vgan_stage1.py:375,382); the three bce sums are always all reported.
__global__ __launch_bounds__(256) void gan_head_kernel(const float* __restrict__ logit, int ldl, int B,
                                                       float* __restrict__ prob, float* __restrict__ scal,
                                                       int parts) {
    __shared__ float sh[4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * B; i += gridDim.x * blockDim.x) {
        float p, bce, dl;
        const int part = i / B;
        gan_terms(logit[(int64_t)i * ldl], part, p, bce, dl);
        s[part] += bce;
        if ((parts >> part) & 1) s[3] += dl * dl;
        if (prob) prob[i] = p;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = block_sum_256(s[k], sh);
        if (threadIdx.x == 0) atomicAdd(scal + (k < 3 ? k : (int)S_DL2), r);
    }
}

// d(sum bce)/d logit * gscale * (*norm) -> fp16 rows of stride ldg (column 0; others zero)
__global__ void gan_head_bwd_kernel(const float* __restrict__ logit, int ldl, int B, half_t* __restrict__ dlogit,
                                    int ldg, float gscale, const float* __restrict__ norm, int parts) {
    const float sc = gscale * (norm ? *norm : 1.f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * B; i += gridDim.x * blockDim.x) {
        float p, bce, dl;
        gan_terms(logit[(int64_t)i * ldl], i / B, p, bce, dl);
        if (!((parts >> (i / B)) & 1)) dl = 0.f;
        for (int c = 0; c < ldg; ++c) dlogit[(int64_t)i * ldg + c] = (half_t)(c == 0 ? dl * sc : 0.f);
    }
}

// ---- WAE latent-discriminator terms: t = -w * sum log(d + 1e-3)  or  -w * sum log(1 - d + 1e-3)
// on logits (sigmoid applied here); writes loss total and cotangent w.r.t. logits.
__global__ __launch_bounds__(256) void wae_logloss_kernel(const float* __restrict__ logit, int ldl, int n,
                                                          int one_minus, float w, float* __restrict__ total,
                                                          float* __restrict__ prob, half_t* __restrict__ dlogit,
                                                          int ldg, float gscale) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float l = logit[(int64_t)i * ldl];
        const float p = 1.f / (1.f + expf(-l));
        float dl;
        if (one_minus) { s += -w * logf(1.f - p + 1e-3f); dl = w * p * (1.f - p) / (1.f - p + 1e-3f); }
        else { s += -w * logf(p + 1e-3f); dl = -w * p * (1.f - p) / (p + 1e-3f); }
        if (prob) prob[i] = p;
        if (dlogit)
            for (int c = 0; c < ldg; ++c) dlogit[(int64_t)i * ldg + c] = (half_t)(c == 0 ? dl * gscale : 0.f);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0 && total) atomicAdd(total, s);
}

// ---- loss composition + equilibrium gate + stream normalisation factors, one thread.
//  in : bce_o, bce_p, bce_s, kl, mse, nle, dl2 (all already summed over the global batch)
//  out: loss_encoder, loss_discriminator, loss_decoder; nA = 1/rms(dlogit), nB = 1/rms(dfeat), ratio = nA/nB, the
//       mixing weights S_C1..S_C3, S_GDEC, S_KLW; flags[0] = train_dis, flags[1] = train_dec
//  hp (device, may be null -> the host values): [lambda_mse, equilibrium, margin, beta] -- the per-epoch decays of
//  train_vgan_stage1.py:448-458 update that vector, so a replayed HIP graph sees them.
//  mode (train_vgan_stage1.py:359-388): vae-gan, beta-vae (KL weight beta / batch), dcgan (pixel nle instead of the
//  feature mse, discriminator loss without the reconstruction term), vae (dcgan's losses, decoder loss lambda * nle,
//  discriminator not trained unless the gate re-arms both).  dl2 must match the mode's discriminator loss.
__global__ void compose_gate_kernel(float* __restrict__ scal, int* __restrict__ flags, float batch, float nfeat,
                                    float npix, float lambda_mse, float equilibrium, float margin, float beta,
                                    const float* __restrict__ hp, int mode, int gate_on, int force_dis,
                                    int force_dec) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (hp) { lambda_mse = hp[0]; equilibrium = hp[1]; margin = hp[2]; beta = hp[3]; }
    const float bo = scal[S_BCE_O], bp = scal[S_BCE_P], bs = scal[S_BCE_S];
    const bool pix = mode == M_DCGAN || mode == M_VAE;
    const float rec = pix ? scal[S_NLE] : scal[S_MSE];
    const float l_dis = pix ? bo + bs : bo + bp + bs;
    const float klw = mode == M_BETAVAE ? beta / batch : 1.f;
    scal[S_KLW] = klw;
    scal[S_LENC] = klw * scal[S_KL] + rec;
    scal[S_LDIS] = l_dis;
    scal[S_LDEC] = mode == M_VAE ? lambda_mse * rec : lambda_mse * rec - (1.f - lambda_mse) * l_dis;
    const float rms_a = sqrtf(scal[S_DL2] / (3.f * batch));
    const float rms_b = sqrtf(2.f * scal[S_MSE] / (batch * nfeat));
    const float rms_p = sqrtf(2.f * scal[S_NLE] / (batch * fmaxf(npix, 1.f)));
    const float na = 1.f / fmaxf(rms_a, 1e-20f);
    const float nb = 1.f / fmaxf(rms_b, 1e-20f);
    const float np = 1.f / fmaxf(rms_p, 1e-20f);
    scal[S_NA] = na;
    scal[S_NB] = nb;
    scal[S_NP] = np;
    scal[S_RATIO] = na / nb;
    scal[S_ONE] = 1.f;
    scal[14] = 0.f;         // S_ESQ of the float form of the encoder-stream re-normalisation (fmri_sumsq adds into it;
                            // the steps use fmri_sumsq_f64, which clears its own double accumulator)
    scal[S_C1] = lambda_mse * na / nb;
    scal[S_C2] = 1.f - lambda_mse;
    scal[S_C3] = lambda_mse * na;
    scal[S_GDEC] = mode == M_VAE ? np / fmaxf(lambda_mse, 1e-30f) : na;
    int train_dis = mode == M_VAE ? 0 : 1, train_dec = 1;
    if (gate_on) {
        const float mo = bo / batch, mp = bp / batch;
        if (mo < equilibrium - margin || mp < equilibrium - margin) train_dis = 0;
        if (mo > equilibrium + margin || mp > equilibrium + margin) train_dec = 0;
        if (!train_dis && !train_dec) { train_dis = 1; train_dec = 1; }
    }
    if (force_dis >= 0) train_dis = force_dis;
    if (force_dec >= 0) train_dec = force_dec;
    flags[0] = train_dis;
    flags[1] = train_dec;
}

// ---- out = a*(*pa)*x + b*y over fp16 (cotangent mixing, fp32 math); y / pa may be null
__global__ void axpby_f16_kernel(const half_t* __restrict__ x, const half_t* __restrict__ y, half_t* __restrict__ out,
                                 int64_t n8, float a, float b, const float* __restrict__ pa,
                                 const float* __restrict__ pb) {
    const float aa = a * (pa ? *pa : 1.f);
    b *= pb ? *pb : 1.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const h8 xv = *(const h8*)(x + i * 8);
        h8 o;
        if (y) {
            const h8 yv = *(const h8*)(y + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(aa * (float)xv[j] + b * (float)yv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(aa * (float)xv[j]);
        }
        *(h8*)(out + i * 8) = o;
    }
}

// ---- device-side re-normalisation of a cotangent before it enters an fp16 backward pass:
//   sumsq_kernel : *acc += sum x^2           (acc can then be all-reduced across data-parallel ranks)
//   renorm_kernel: f = 1/rms, out16 = x * f * scale;  *factor_out = (*factor_in) * f  (factor_in may be null)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ acc) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += x[i] * x[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}

// The double-precision form (fmri_sumsq_f64 / fmri_renorm_f64) is the one the steps use: the encoder cotangent holds
// 0.5 * (exp(logvar) - 1) (the KL term), whose SQUARE leaves fp32 at logvar > 44 -- the float sum then reads inf, the
// factor 0, and the optimizer divides 0 by 0 -- while the reference's fp32 arithmetic is finite up to logvar 88.
__global__ __launch_bounds__(256) void sumsq64_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ acc) {
    __shared__ double shd[4];
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)x[i] * (double)x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (shd[0] + shd[1]) + (shd[2] + shd[3]));
}

template <typename T>
__global__ void renorm_kernel(const float* __restrict__ x, half_t* __restrict__ out, int64_t n, float scale,
                              const T* __restrict__ sumsq, float count, const float* __restrict__ factor_in,
                              float* __restrict__ factor_out) {
    const float f = 1.f / fmaxf((float)sqrt((double)*sumsq / (double)count), 1e-20f);
    const float sc = f * scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (half_t)(x[i] * sc);
    if (blockIdx.x == 0 && threadIdx.x == 0 && factor_out) *factor_out = (factor_in ? *factor_in : 1.f) * f;
}

// ---- fused optimizers over a flat fp32 parameter buffer; `flag` (device int, may be null) gates the
// whole update so the equilibrium gate never needs a host sync.  g_true = g * gscale / (*gdev), then clamped.
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, int64_t n,
                               float lr, float alpha, float eps, float gscale, const float* __restrict__ gdev,
                               float clamp, const int* __restrict__ flag, const float* __restrict__ lr_dev) {
    if (flag && *flag == 0) return;
    if (lr_dev) lr = *lr_dev;      // device-resident learning rate: schedules reach a replayed HIP graph
    const float gs = gscale / (gdev ? *gdev : 1.f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[i] * gs;
        if (clamp > 0.f) gg = fminf(fmaxf(gg, -clamp), clamp);
        const float s = alpha * sq[i] + (1.f - alpha) * gg * gg;
        sq[i] = s;
        p[i] -= lr * gg / (sqrtf(s) + eps);
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale, const float* __restrict__ gdev, float clamp,
                            const int* __restrict__ flag, const float* __restrict__ lr_dev,
                            const int* __restrict__ t_dev) {
    if (flag && *flag == 0) return;
    if (lr_dev) lr = *lr_dev;
    if (t_dev) {                   // device-resident step count (already incremented for this step): bias corrections
        const float t = (float)*t_dev;
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    const float gs = gscale / (gdev ? *gdev : 1.f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[i] * gs;
        if (clamp > 0.f) gg = fminf(fmaxf(gg, -clamp), clamp);
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

// Process-wide reduction mode (fmri_set_deterministic; FMRI_DETERMINISTIC=1 sets the initial value): when on, every
// kernel of this file that ends in an atomic add onto a scalar is launched as ONE block, so that each total is a
// fixed-order sum and two runs of the same step produce the same bits.  Slower (the feature / pixel sums of a
// 256-image batch take a few hundred microseconds on one CU): a verification mode, not the default.
static int env_deterministic() {
    const char* e = getenv("FMRI_DETERMINISTIC");
    return (e && e[0] == '1') ? 1 : 0;
}
int g_deterministic = env_deterministic();

// grid of a reduction kernel whose blocks meet in an atomic add
static inline int rblk(int blocks) { return g_deterministic ? 1 : blocks; }

static inline int nblk(int64_t total, int cap = 4096) {
    int64_t b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? OK : E_LAUNCH)

int latent_fwd_launch(const float* head, const float* eps, int B, int Z, int zp, half_t* z16, float* kl_rows,
                      float* kl_total, int sample, hipStream_t st) {
    const int grid = kl_total ? rblk((B + 3) / 4) : (B + 3) / 4;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(grid), dim3(256), 0, st, head, eps, B, Z, zp, z16, kl_rows, kl_total,
                       sample);
    return LAUNCH_OK();
}
int latent_ranged_launch(const float* head, const float* eps, int B, int Z, int zp, half_t* z16, float* kl_rows,
                         float* kl_total, int sample, float* z32, float* zmax, float* zscale, float cap, int phase,
                         hipStream_t st) {
    if (phase & 1) {
        const int grid = kl_total ? rblk((B + 3) / 4) : (B + 3) / 4;
        hipLaunchKernelGGL(latent_z32_kernel, dim3(grid), dim3(256), 0, st, head, eps, B, Z, z32, kl_rows, kl_total, sample,
                           zmax);
    }
    if (phase & 2)
        hipLaunchKernelGGL(latent_pack_kernel, dim3(nblk((int64_t)B * zp, 256)), dim3(256), 0, st, (const float*)z32, B, Z,
                           zp, z16, (const float*)zmax, zscale, cap);
    return LAUNCH_OK();
}
float latent_range_scale_host(float zmax, float cap) { return range_scale(zmax, cap); }
int rows_absmax_launch(const float* x, int64_t n, float* zmax, hipStream_t st) {
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(nblk(n, 256)), dim3(256), 0, st, x, n, zmax);
    return LAUNCH_OK();
}
int latent_bwd_launch(const float* head, const float* eps, const float* dz, int ldz, float dz_unscale, float kl_w,
                      const float* kl_dev, int B, int Z, float out_scale, half_t* dhead16, float* dhead32, int sample,
                      hipStream_t st) {
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblk((int64_t)B * Z)), dim3(256), 0, st, head, eps, dz, ldz,
                       dz_unscale, kl_w, kl_dev, B, Z, out_scale, dhead16, dhead32, sample);
    return LAUNCH_OK();
}
int feat_mse_launch(const half_t* feat, int B, int F, float* mse_rows, float* mse_total, hipStream_t st) {
    hipLaunchKernelGGL(feat_mse_kernel, dim3(mse_total ? rblk(B) : B), dim3(256), 0, st, feat, B, F, mse_rows,
                       mse_total);
    return LAUNCH_OK();
}
int feat_mse_bwd_launch(const half_t* feat, int B, int F, half_t* dfeat, float gscale, const float* norm,
                        hipStream_t st) {
    hipLaunchKernelGGL(feat_mse_bwd_kernel, dim3(B), dim3(256), 0, st, feat, B, F, dfeat, gscale, norm);
    return LAUNCH_OK();
}
int pixel_sq_launch(const half_t* x, const half_t* xt, int64_t npix, int C, int Cp, float* total, half_t* dxt,
                    float gscale, hipStream_t st) {
    const int grid = total ? rblk(nblk(npix, 1024)) : nblk(npix, 1024);
    hipLaunchKernelGGL(pixel_sq_kernel, dim3(grid), dim3(256), 0, st, x, xt, npix, C, Cp, total, dxt, gscale);
    return LAUNCH_OK();
}
int gan_head_launch(const float* logit, int ldl, int B, float* prob, float* scal, int parts, hipStream_t st) {
    hipLaunchKernelGGL(gan_head_kernel, dim3(rblk(nblk(3 * (int64_t)B, 64))), dim3(256), 0, st, logit, ldl, B, prob,
                       scal, parts);
    return LAUNCH_OK();
}
int gan_head_bwd_launch(const float* logit, int ldl, int B, half_t* dlogit, int ldg, float gscale, const float* norm,
                        int parts, hipStream_t st) {
    hipLaunchKernelGGL(gan_head_bwd_kernel, dim3(nblk(3 * (int64_t)B, 64)), dim3(256), 0, st, logit, ldl, B, dlogit,
                       ldg, gscale, norm, parts);
    return LAUNCH_OK();
}
int wae_logloss_launch(const float* logit, int ldl, int n, int one_minus, float w, float* total, float* prob,
                       half_t* dlogit, int ldg, float gscale, hipStream_t st) {
    const int grid = total ? rblk(nblk(n, 64)) : nblk(n, 64);
    hipLaunchKernelGGL(wae_logloss_kernel, dim3(grid), dim3(256), 0, st, logit, ldl, n, one_minus, w, total, prob,
                       dlogit, ldg, gscale);
    return LAUNCH_OK();
}
int compose_gate_launch(float* scal, int* flags, float batch, float nfeat, float npix, float lambda_mse,
                        float equilibrium, float margin, float beta, const float* hp_dev, int mode, int gate_on,
                        int force_dis, int force_dec, hipStream_t st) {
    hipLaunchKernelGGL(compose_gate_kernel, dim3(1), dim3(64), 0, st, scal, flags, batch, nfeat, npix, lambda_mse,
                       equilibrium, margin, beta, hp_dev, mode, gate_on, force_dis, force_dec);
    return LAUNCH_OK();
}
__global__ void counter_inc_kernel(int* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t += 1;
}
int counter_inc_launch(int* t, hipStream_t st) {
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(64), 0, st, t);
    return LAUNCH_OK();
}
int axpby_f16_launch(const half_t* x, const half_t* y, half_t* out, int64_t n, float a, float b, const float* pa,
                     const float* pb, hipStream_t st) {
    hipLaunchKernelGGL(axpby_f16_kernel, dim3(nblk(n / 8)), dim3(256), 0, st, x, y, out, n / 8, a, b, pa, pb);
    return LAUNCH_OK();
}
int sumsq_launch(const float* x, int64_t n, float* acc, hipStream_t st) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(rblk(nblk(n, 256))), dim3(256), 0, st, x, n, acc);
    return LAUNCH_OK();
}
int renorm_launch(const float* x, half_t* out, int64_t n, float scale, const float* sumsq, float count,
                  const float* factor_in, float* factor_out, hipStream_t st) {
    hipLaunchKernelGGL(renorm_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, st, x, out, n, scale, sumsq, count,
                       factor_in, factor_out);
    return LAUNCH_OK();
}
int sumsq64_launch(const float* x, int64_t n, double* acc, int zero_first, hipStream_t st) {
    if (zero_first && hipMemsetAsync(acc, 0, sizeof(double), st) != hipSuccess) return E_LAUNCH;
    hipLaunchKernelGGL(sumsq64_kernel, dim3(rblk(nblk(n, 256))), dim3(256), 0, st, x, n, acc);
    return LAUNCH_OK();
}
int renorm64_launch(const float* x, half_t* out, int64_t n, float scale, const double* sumsq, float count,
                    const float* factor_in, float* factor_out, hipStream_t st) {
    hipLaunchKernelGGL(renorm_kernel<double>, dim3(nblk(n, 256)), dim3(256), 0, st, x, out, n, scale, sumsq, count,
                       factor_in, factor_out);
    return LAUNCH_OK();
}
int rmsprop_launch(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float gscale,
                   const float* gdev, float clamp, const int* flag, const float* lr_dev, hipStream_t st) {
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nblk(n)), dim3(256), 0, st, p, g, sq, n, lr, alpha, eps, gscale, gdev,
                       clamp, flag, lr_dev);
    return LAUNCH_OK();
}
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                float bc1, float bc2_sqrt, float gscale, const float* gdev, float clamp, const int* flag,
                const float* lr_dev, const int* t_dev, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2_sqrt,
                       gscale, gdev, clamp, flag, lr_dev, t_dev);
    return LAUNCH_OK();
}

}  // namespace fmri
