// Latent / loss / optimizer kernels of the VAE/GAN step (wavefront reductions, fp32).
//
// Restates (reference):
//   reparameterize           models/vae_gan.py:266-269     z = eps*exp(0.5*logvar) + mu
//   VaeGan.loss              models/vae_gan.py:302-320     nle, kl, feature mse, 3 x bce (eps 1e-3 in the log)
//   loss composition + gate  train/train_vgan_stage1.py:368-404
//   RMSprop / Adam updates   train/train_vgan_stage1.py:275-283, train/train_wae_stage1.py:221-224
#include "kernels.h"

namespace fmri {

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
// one wave per sample row.
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                                         int B, int Z, int zp, half_t* __restrict__ z16,
                                                         float* __restrict__ kl_rows, float* __restrict__ kl_total,
                                                         int sample /* 0: z = mu (WAE) */) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
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
    if (lane == 0) {
        if (kl_rows) kl_rows[row] = kl;
        if (kl_total) atomicAdd(kl_total, kl);
    }
}

// backward: dhead[row] = [ dz + kl_w*mu | dz*eps*0.5*exp(0.5 lv) + kl_w*0.5*(exp(lv)-1) ] * out_scale
// dz arrives as fp32 [B][ldz] scaled by 1/dz_scale_inv (dz_true = dz * dz_unscale).
__global__ void latent_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                  const float* __restrict__ dz, int ldz, float dz_unscale, float kl_w, int B, int Z,
                                  float out_scale, half_t* __restrict__ dhead16, float* __restrict__ dhead32,
                                  int sample) {
    const int64_t total = (int64_t)B * Z;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Z);
        const int64_t row = i / Z;
        const float mu = head[row * 2 * Z + j];
        const float lv = head[row * 2 * Z + Z + j];
        const float g = dz ? dz[row * ldz + j] * dz_unscale : 0.f;
        const float dmu = g + kl_w * mu;
        float dlv = kl_w * 0.5f * (__expf(lv) - 1.f);
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

// ---- feature-matching term: mse_b = sum_f 0.5 (f_o - f_p)^2 over the raw conv3 features and its
// cotangent w.r.t. the 3B feature rows (orig: +d, pred: -d, sampled: 0), scaled for fp16 storage.
// one block per sample.
__global__ __launch_bounds__(256) void feat_mse_kernel(const half_t* __restrict__ feat, int B, int F,
                                                       float* __restrict__ mse_rows, float* __restrict__ mse_total,
                                                       half_t* __restrict__ dfeat, float gscale) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const half_t* fo = feat + (int64_t)b * F;
    const half_t* fp = feat + (int64_t)(B + b) * F;
    float s = 0.f;
    for (int i = threadIdx.x * 8; i < F; i += 256 * 8) {
        const h8 o = *(const h8*)(fo + i);
        const h8 p = *(const h8*)(fp + i);
        h8 d, nd, z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float df = (float)o[j] - (float)p[j];
            s += 0.5f * df * df;
            d[j] = (half_t)(df * gscale);
            nd[j] = (half_t)(-df * gscale);
            z[j] = (half_t)0.f;
        }
        if (dfeat) {
            *(h8*)(dfeat + (int64_t)b * F + i) = d;
            *(h8*)(dfeat + (int64_t)(B + b) * F + i) = nd;
            *(h8*)(dfeat + (int64_t)(2 * B + b) * F + i) = z;
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) {
        if (mse_rows) mse_rows[b] = s;
        if (mse_total) atomicAdd(mse_total, s);
    }
}

// ---- pixel term: nle_total = sum 0.5 (x - x_tilde)^2 over real channels of NHWC-padded images;
// optional cotangent d nle / d x_tilde = -(x - x_tilde) * gscale.
__global__ __launch_bounds__(256) void pixel_sq_kernel(const half_t* __restrict__ x, const half_t* __restrict__ xt,
                                                       int64_t npix, int C, int Cp, float* __restrict__ total,
                                                       half_t* __restrict__ dxt, float gscale) {
    __shared__ float sh[4];
    float s = 0.f;
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
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0 && total) atomicAdd(total, s);
}

// ---- discriminator class head: logits (fp32 [3B], bias already added) -> sigmoid, the three BCE sums
// (eps 1e-3 inside the log, models/vae_gan.py:316-318) and d(sum bce)/d logit (scaled, fp16 row stride ldg)
// scal layout (floats): [0]=bce_orig [1]=bce_pred [2]=bce_samp
__global__ __launch_bounds__(256) void gan_head_kernel(const float* __restrict__ logit, int ldl, int B,
                                                       float* __restrict__ prob, float* __restrict__ scal,
                                                       half_t* __restrict__ dlogit, int ldg, float gscale,
                                                       int pred_is_sampled /* DCGAN: unused */) {
    __shared__ float sh[4];
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * B; i += gridDim.x * blockDim.x) {
        const float l = logit[(int64_t)i * ldl];
        const float p = 1.f / (1.f + expf(-l));
        const int part = i / B;
        float bce, dl;
        if (part == 0) {
            bce = -logf(p + 1e-3f);
            dl = -p * (1.f - p) / (p + 1e-3f);
        } else {
            bce = -logf(1.f - p + 1e-3f);
            dl = p * (1.f - p) / (1.f - p + 1e-3f);
        }
        s[part] += bce;
        if (prob) prob[i] = p;
        if (dlogit) {
            for (int c = 0; c < ldg; ++c) dlogit[(int64_t)i * ldg + c] = (half_t)(c == 0 ? dl * gscale : 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r = block_sum_256(s[k], sh);
        if (threadIdx.x == 0) atomicAdd(scal + k, r);
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

// ---- loss composition + equilibrium gate, one thread.  scal (floats):
//  in : [0]=bce_o [1]=bce_p [2]=bce_s [3]=kl [4]=mse [5]=nle
//  out: [6]=loss_encoder [7]=loss_discriminator [8]=loss_decoder ; flags[0]=train_dis flags[1]=train_dec
__global__ void compose_gate_kernel(float* __restrict__ scal, int* __restrict__ flags, float batch, float lambda_mse,
                                    float equilibrium, float margin, int gate_on, int force_dis, int force_dec) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float bo = scal[0], bp = scal[1], bs = scal[2];
    const float l_dis = bo + bp + bs;
    scal[6] = scal[3] + scal[4];
    scal[7] = l_dis;
    scal[8] = lambda_mse * scal[4] - (1.f - lambda_mse) * l_dis;
    int train_dis = 1, train_dec = 1;
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

// ---- out = a*x + b*y over fp16 (cotangent mixing, fp32 math); y may be null
__global__ void axpby_f16_kernel(const half_t* __restrict__ x, const half_t* __restrict__ y, half_t* __restrict__ out,
                                 int64_t n8, float a, float b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const h8 xv = *(const h8*)(x + i * 8);
        h8 o;
        if (y) {
            const h8 yv = *(const h8*)(y + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(a * (float)xv[j] + b * (float)yv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(a * (float)xv[j]);
        }
        *(h8*)(out + i * 8) = o;
    }
}

// ---- fused optimizers over a flat fp32 parameter buffer; `flag` (device int, may be null) gates the
// whole update so the equilibrium gate never needs a host sync.  g is multiplied by gscale, then clamped.
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, int64_t n,
                               float lr, float alpha, float eps, float gscale, float clamp,
                               const int* __restrict__ flag) {
    if (flag && *flag == 0) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[i] * gscale;
        if (clamp > 0.f) gg = fminf(fmaxf(gg, -clamp), clamp);
        const float s = alpha * sq[i] + (1.f - alpha) * gg * gg;
        sq[i] = s;
        p[i] -= lr * gg / (sqrtf(s) + eps);
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale, float clamp, const int* __restrict__ flag) {
    if (flag && *flag == 0) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[i] * gscale;
        if (clamp > 0.f) gg = fminf(fmaxf(gg, -clamp), clamp);
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

static inline int nblk(int64_t total, int cap = 4096) {
    int64_t b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? OK : E_LAUNCH)

int latent_fwd_launch(const float* head, const float* eps, int B, int Z, int zp, half_t* z16, float* kl_rows,
                      float* kl_total, int sample, hipStream_t st) {
    hipLaunchKernelGGL(latent_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, head, eps, B, Z, zp, z16, kl_rows,
                       kl_total, sample);
    return LAUNCH_OK();
}
int latent_bwd_launch(const float* head, const float* eps, const float* dz, int ldz, float dz_unscale, float kl_w,
                      int B, int Z, float out_scale, half_t* dhead16, float* dhead32, int sample, hipStream_t st) {
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblk((int64_t)B * Z)), dim3(256), 0, st, head, eps, dz, ldz,
                       dz_unscale, kl_w, B, Z, out_scale, dhead16, dhead32, sample);
    return LAUNCH_OK();
}
int feat_mse_launch(const half_t* feat, int B, int F, float* mse_rows, float* mse_total, half_t* dfeat, float gscale,
                    hipStream_t st) {
    hipLaunchKernelGGL(feat_mse_kernel, dim3(B), dim3(256), 0, st, feat, B, F, mse_rows, mse_total, dfeat, gscale);
    return LAUNCH_OK();
}
int pixel_sq_launch(const half_t* x, const half_t* xt, int64_t npix, int C, int Cp, float* total, half_t* dxt,
                    float gscale, hipStream_t st) {
    hipLaunchKernelGGL(pixel_sq_kernel, dim3(nblk(npix, 1024)), dim3(256), 0, st, x, xt, npix, C, Cp, total, dxt,
                       gscale);
    return LAUNCH_OK();
}
int gan_head_launch(const float* logit, int ldl, int B, float* prob, float* scal, half_t* dlogit, int ldg,
                    float gscale, hipStream_t st) {
    hipLaunchKernelGGL(gan_head_kernel, dim3(nblk(3 * (int64_t)B, 64)), dim3(256), 0, st, logit, ldl, B, prob, scal,
                       dlogit, ldg, gscale, 0);
    return LAUNCH_OK();
}
int wae_logloss_launch(const float* logit, int ldl, int n, int one_minus, float w, float* total, float* prob,
                       half_t* dlogit, int ldg, float gscale, hipStream_t st) {
    hipLaunchKernelGGL(wae_logloss_kernel, dim3(nblk(n, 64)), dim3(256), 0, st, logit, ldl, n, one_minus, w, total,
                       prob, dlogit, ldg, gscale);
    return LAUNCH_OK();
}
int compose_gate_launch(float* scal, int* flags, float batch, float lambda_mse, float equilibrium, float margin,
                        int gate_on, int force_dis, int force_dec, hipStream_t st) {
    hipLaunchKernelGGL(compose_gate_kernel, dim3(1), dim3(64), 0, st, scal, flags, batch, lambda_mse, equilibrium,
                       margin, gate_on, force_dis, force_dec);
    return LAUNCH_OK();
}
int axpby_f16_launch(const half_t* x, const half_t* y, half_t* out, int64_t n, float a, float b, hipStream_t st) {
    hipLaunchKernelGGL(axpby_f16_kernel, dim3(nblk(n / 8)), dim3(256), 0, st, x, y, out, n / 8, a, b);
    return LAUNCH_OK();
}
int rmsprop_launch(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float gscale,
                   float clamp, const int* flag, hipStream_t st) {
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nblk(n)), dim3(256), 0, st, p, g, sq, n, lr, alpha, eps, gscale, clamp,
                       flag);
    return LAUNCH_OK();
}
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                float bc1, float bc2_sqrt, float gscale, float clamp, const int* flag, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2_sqrt,
                       gscale, clamp, flag);
    return LAUNCH_OK();
}

}  // namespace fmri
