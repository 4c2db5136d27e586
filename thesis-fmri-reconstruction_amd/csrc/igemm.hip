// Tap-list implicit GEMM on MFMA for gfx950 -- the one contraction kernel behind every
// conv / transposed-conv / dense forward and data-gradient of the VAE/GAN step.
//
//   out[n, y*os+oy0, x*os+ox0, co] = act( bias[co] +
//        sum_{tap<T} sum_{ci<Ci} in[n, y*s+dy(tap), x*s+dx(tap), ci] * w[co][tap*Ci + ci] )
//
// Replaces (reference, models/vae_gan.py): nn.Conv2d k5 s2 p2 (:18-20), nn.ConvTranspose2d k5 s2 p2
// (:46-53, as 4 output-parity classes -> no multiplications by inserted zeros), nn.Linear (:79,:107,
// :156, T = 1) and their autograd data-gradients (conv dgrad == transposed conv, deconv dgrad == conv).
//
// Design (MI355X):
//   * GEMM view M = N*Yc*Xc output pixels, N = Co, K = T*Ci.  Block tile 128 x BN x 64, 4 waves.
//   * Both operand tiles are staged global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4):
//     the im2col gather is done by the DMA's per-lane source address, padding taps read a zero page.
//   * LDS rows are 128 B (64 halfs); 16-B chunks are XOR-swizzled with (row>>1)&7 on the *source*
//     side (the DMA writes linearly) and on the ds_read_b128 side -> conflict-free fragment reads.
//   * 2-stage ring, one barrier per K-step, next tile's DMA in flight under the current tile's MFMAs.
//   * v_mfma_f32_16x16x32_f16, weights as the A operand so every lane owns 4 consecutive output
//     channels of one pixel -> 8-byte NHWC stores.
//   * fp32 accumulate; epilogue fuses bias + ReLU/tanh, or writes fp32 split-K slabs.
#include "kernels.h"
#include <cstdlib>
#include <cstring>

namespace fmri {

// BM x BN x 64 tile, WM x WN waves (4 waves: BM = 128, two blocks per CU; 8 waves: BM = BN = 256, one block per CU --
// half the operand bytes per FLOP and per-wave 128 x 64 sub-tiles for the >= 256-channel layers).
template <int BM, int BN, int WM, int WN, bool OUT_F32, bool UNI>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(const IgemmArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;                 // tile rows covered by one DMA pass of the block
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int AROWS = BM / RPP;
    constexpr int BROWS = BN / RPP;
    static_assert(AROWS == 4, "4 gather rows per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int icls = blockIdx.z % a.ncls;
    const int split = blockIdx.z / a.ncls;
    const IgemmClass& c = a.cls[icls];
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * BM;
    if (m0 >= c.M) return;
    const int co0 = by * BN;

    const int trow = tid >> 3;
    const int cphys = tid & 7;
    const int clog = cphys ^ ((trow >> 1) & 7);

    // geometry of the 4 gather rows owned by this thread (fixed for the whole K loop)
    int iy0[4], ix0[4], pixbase[4];
    const int YX = c.Yc * c.Xc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + trow + RPP * i;
        const bool v = m < c.M;
        const uint32_t mm = v ? (uint32_t)m : 0u;
        const uint32_t n = fd_div(mm, c.fdYX);
        const uint32_t rem = mm - n * (uint32_t)YX;
        const uint32_t y = fd_div(rem, c.fdX);
        const uint32_t x = rem - y * (uint32_t)c.Xc;
        iy0[i] = v ? (int)y * a.s : -(1 << 20);
        ix0[i] = (int)x * a.s;
        pixbase[i] = (int)n * a.Hi * a.Wi;
    }

    const half_t* wrow = a.w + c.w_off + (int64_t)(co0 + trow) * c.Kpad + clog * 8;

    // loop-invariant scalars of the K loop, pinned in SGPRs (see FMRI_KEEP)
    int kT = c.T, kTW = c.TW, kdy0 = c.dy0, kdx0 = c.dx0, kdstep = c.dstep, kHi = a.Hi, kWi = a.Wi, kCi = a.Ci;
    int kKpad = c.Kpad;
    uint32_t tw_magic = c.fdTW.magic, tw_sh = c.fdTW.sh, cpt_magic = a.fdCpt.magic, cpt_sh = a.fdCpt.sh;
    uint32_t ci_magic = a.fdCi.magic, ci_sh = a.fdCi.sh;
    const half_t* kin = a.in;
    const half_t* kzero = a.zero;
    FMRI_KEEP(kT); FMRI_KEEP(kTW); FMRI_KEEP(kdy0); FMRI_KEEP(kdx0); FMRI_KEEP(kdstep); FMRI_KEEP(kHi); FMRI_KEEP(kWi);
    FMRI_KEEP(kCi); FMRI_KEEP(kKpad); FMRI_KEEP(tw_magic); FMRI_KEEP(tw_sh); FMRI_KEEP(cpt_magic); FMRI_KEEP(cpt_sh);
    FMRI_KEEP(ci_magic); FMRI_KEEP(ci_sh); FMRI_KEEP(kin); FMRI_KEEP(kzero);
    const FastDiv fTW{tw_magic, tw_sh, (uint32_t)kTW, 0}, fCpt{cpt_magic, cpt_sh, 0, 0}, fCi{ci_magic, ci_sh, 0, 0};

    // UNI (Ci % 64 == 0): a K-step lies inside one tap, so the tap decode is wave-uniform (scalar) and a
    // row's source address is rowoff[i] + one scalar offset: no per-lane multiplies inside the K loop.
    // The general path (a K-step may straddle taps, e.g. Ci = 32) uses the same row offsets; its tap decode is per lane
    // but once per step and thread, not per row: no 32-bit multiplies (quarter rate) per row inside the K loop.
    int rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        rowoff[i] = iy0[i] >= 0 ? (pixbase[i] + iy0[i] * a.Wi + ix0[i]) * a.Ci + (UNI ? clog * 8 : 0) : 0;
    const int cpt = a.Ci >> 6;     // K-steps per tap (UNI)

    int ltap = 0, lcstep = 0;                    // UNI: position of the step being staged in the chunk-major order
    auto stage_load = [&](int buf, int kstep) {
        char* dstA = smem + buf * STAGE + wave * (8 * 128);
        int wstep = kstep;                       // weight column block of this step
        if (UNI) {
            // chunk-major K order (all taps of a 64-channel chunk, then the next chunk): the input bytes a block
            // re-reads across taps are then one chunk of its pixel region, not all channels -- the L2-resident set
            // (tap, cstep) of this step are carried by the K loop below (ltap, lcstep)
            const int tap = ltap, cstep = lcstep;
            const int ty = (int)fd_div((uint32_t)tap, fTW);
            const int tx = tap - ty * kTW;
            const int dy = kdy0 + ty * kdstep;
            const int dx = kdx0 + tx * kdstep;
            const bool tv = tap < kT;
            const int tapoff = (dy * kWi + dx) * kCi + cstep * 64;
            wstep = tap * cpt + cstep;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = iy0[i] + dy;
                const int ix = ix0[i] + dx;
                const bool ok = tv && (unsigned)iy < (unsigned)kHi && (unsigned)ix < (unsigned)kWi;
                const half_t* src = ok ? kin + (rowoff[i] + tapoff) : kzero;
                glds16(src, dstA + i * (RPP * 128));
            }
        } else {
            const int k = kstep * 64 + clog * 8;
            const int tap = (int)fd_div((uint32_t)k, fCi);
            const int ci = k - tap * kCi;
            const int ty = (int)fd_div((uint32_t)tap, fTW);
            const int tx = tap - ty * kTW;
            const int dy = kdy0 + ty * kdstep;
            const int dx = kdx0 + tx * kdstep;
            const bool tv = tap < kT;
            const int tapoff = (dy * kWi + dx) * kCi + ci;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = iy0[i] + dy;
                const int ix = ix0[i] + dx;
                const bool ok = tv && (unsigned)iy < (unsigned)kHi && (unsigned)ix < (unsigned)kWi;
                const half_t* src = ok ? kin + (rowoff[i] + tapoff) : kzero;
                glds16(src, dstA + i * (RPP * 128));
            }
        }
        char* dstB = smem + buf * STAGE + A_BYTES + wave * (8 * 128);
        const half_t* wsrc = wrow + (int64_t)wstep * 64;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) glds16(wsrc + (int64_t)i * RPP * kKpad, dstB + i * (RPP * 128));
    };

    f4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;

    auto compute = [&](int buf) {
        const char* As = smem + buf * STAGE;
        const char* Bs = As + A_BYTES;
        if constexpr (TM * TN <= 16) {
            // all fragment reads of the K-step are issued up front: the second half's LDS latency hides under
            // the first half's MFMAs (the compiler then waits with a counted lgkmcnt instead of lgkmcnt(0))
            h8 af[2][TM], bf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row = wm * (BM / WM) + tm * 16 + frow;
                    const int ph = (ks * 4 + fq) ^ ((row >> 1) & 7);
                    af[ks][tm] = *(const h8*)(As + row * 128 + ph * 16);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int row = wn * (BN / WN) + tn * 16 + frow;
                    const int ph = (ks * 4 + fq) ^ ((row >> 1) & 7);
                    bf[ks][tn] = *(const h8*)(Bs + row * 128 + ph * 16);
                }
                if (ks == 0) __builtin_amdgcn_sched_barrier(0);   // first-half reads are issued first
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[tn][tm] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][tn], af[ks][tm], acc[tn][tm], 0, 0, 0);
                if (ks == 0) {
                    // interleave the second-half reads with the first TM+TN MFMAs of the first half, then the rest
#pragma unroll
                    for (int i = 0; i < TM + TN; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // big wave tile: one 32-wide half at a time (the register file holds 128 accumulators); the other wave of
            // the SIMD covers the read latency
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 af[TM], bf[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row = wm * (BM / WM) + tm * 16 + frow;
                    const int ph = (ks * 4 + fq) ^ ((row >> 1) & 7);
                    af[tm] = *(const h8*)(As + row * 128 + ph * 16);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int row = wn * (BN / WN) + tn * 16 + frow;
                    const int ph = (ks * 4 + fq) ^ ((row >> 1) & 7);
                    bf[tn] = *(const h8*)(Bs + row * 128 + ph * 16);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
            }
        }
    };

    const int per = (c.ksteps + a.splits - 1) / a.splits;
    const int kb = split * per;
    const int ke = (kb + per < c.ksteps) ? kb + per : c.ksteps;
    if (UNI) { lcstep = kb / kT; ltap = kb - lcstep * kT; }
    if (kb < ke) stage_load(0, kb);
    for (int it = kb; it < ke; ++it) {
        const int cur = (it - kb) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (UNI && ++ltap == kT) { ltap = 0; ++lcstep; }
        if (it + 1 < ke) stage_load(cur ^ 1, it + 1);
        compute(cur);
    }

    // ---- epilogue: D[i = co][j = pixel]; lane owns pixel (lane&15), channels (lane>>4)*4 .. +3
    const bool stats = !OUT_F32 && a.st.part != nullptr;
    const bool bwd = stats && a.bb.x != nullptr;          // BatchNorm backward statistics + ReLU mask (BnBwdEpi)
    int sgrp = 0;                                          // statistics group of the block's rows
    if (stats && a.st.group_n > 0) sgrp = bx / a.st.tpg[icls];
    f4 ssum[TN], ssq[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) { ssum[i] = (f4){0.f, 0.f, 0.f, 0.f}; ssq[i] = (f4){0.f, 0.f, 0.f, 0.f}; }
    int64_t opix[TM];                                      // output pixel of row tile tm, -1: outside
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + wm * (BM / WM) + tm * 16 + frow;
        opix[tm] = -1;
        if (m >= c.M) continue;
        const uint32_t n = fd_div((uint32_t)m, c.fdYX);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)YX;
        const uint32_t y = fd_div(rem, c.fdX);
        const uint32_t x = rem - y * (uint32_t)c.Xc;
        opix[tm] = ((int64_t)n * a.Ho + (y * a.os + c.oy0)) * a.Wo + (x * a.os + c.ox0);
    }
    const float* gmean = nullptr;
    const float* grstd = nullptr;
    int gimg0 = 0;
    if (bwd) bn_bwd_group(a.bb, sgrp, gmean, grstd, gimg0);
    const int64_t xshift = bwd ? (int64_t)(gimg0 - sgrp * a.st.group_n) * a.Ho * a.Wo : 0;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int co = co0 + wn * (BN / WN) + tn * 16 + fq * 4;
        if (co >= a.CoStore) continue;
        f4 mu, rs, ga, be;
        if (bwd) {
            mu = *(const f4*)(gmean + co);
            rs = *(const f4*)(grstd + co);
            ga = *(const f4*)(a.bb.gamma + co);
            be = *(const f4*)(a.bb.beta + co);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if (opix[tm] < 0) continue;
            f4 v = acc[tn][tm];
            if constexpr (OUT_F32) {
                float* o = (float*)a.out + (int64_t)split * a.slab_stride + opix[tm] * a.CoStore + co;
                *(f4*)o = v;
            } else {
                h4 hv;
                if (bwd) {
                    const h4 xr = *(const h4*)(a.bb.x + (opix[tm] + xshift) * a.CoStore + co);
                    hv = bn_bwd_mask4(v, xr, mu, rs, ga, be, a.bb.relu, ssum[tn], ssq[tn]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float f = v[r];
                        if (co + r < a.Co) {
                            if (a.bias) f += a.bias[co + r];
                            f = act_apply(f, a.act);
                        } else {
                            f = 0.f;
                        }
                        hv[r] = (half_t)f;
                    }
                    if (stats) {
                        // BatchNorm statistics of the STORED values (see StatEpi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f = (float)hv[r];
                            ssum[tn][r] += f;
                            ssq[tn][r] += f * f;
                        }
                    }
                }
                *(h4*)((half_t*)a.out + opix[tm] * a.CoStore + co) = hv;
            }
        }
    }
    if constexpr (!OUT_F32) {
        if (stats) {        // block-uniform
            // row of this block: classes one after the other, a group's tiles dense (the host guarantees that no tile
            // straddles two groups)
            int prow = bx - sgrp * a.st.tpg[icls];
            for (int i = 0; i < icls; ++i) prow += a.st.tpg[i];
            float* row = a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C;
            float vs = 0.f, vq = 0.f;
            stat_lane_add<TN>(ssum, ssq, lane, vs, vq);
            stat_store<TN, WM, WN>(vs, vq, lane, wm, wn, co0, (float*)smem, row, a.st.C);
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_bn(const IgemmArgs& a, int maxM, int copad, bool out_f32, hipStream_t st) {
    dim3 grid((maxM + BM - 1) / BM, copad / BN, a.ncls * a.splits);
    constexpr int NT = WM * WN * 64;
    const int lds = 2 * (BM * 128 + BN * 128);
    // fp32-slab output is used for split-K / fp32 consumers; UNI = K-steps never straddle taps
    const bool uni = (a.Ci & 63) == 0;
    if (route_probe("fmri::igemm_kernel<%d,%d,%d,%d,%s,%s>", BM, BN, WM, WN, out_f32 ? "true" : "false", uni ? "true" : "false"))
        return OK;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, a);
    };
    if (out_f32) {
        if (uni) go(igemm_kernel<BM, BN, WM, WN, true, true>);
        else go(igemm_kernel<BM, BN, WM, WN, true, false>);
    } else {
        if (uni) go(igemm_kernel<BM, BN, WM, WN, false, true>);
        else go(igemm_kernel<BM, BN, WM, WN, false, false>);
    }
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// host entry used by api.hip.  bn_tile in {32, 64, 128}; a 128-channel tiling of a layer with >= 256 output channels
// and enough 256-pixel tiles to fill the chip is promoted to the 256 x 256 tile (FMRI_BIG=off disables).
int igemm_bm(const IgemmArgs& a, int maxM, int bn_tile, int copad, bool out_f32) {
    static const char* big_env = getenv("FMRI_BIG");
    static const bool no_big = big_env && !strcmp(big_env, "off");
    static const int big_min = big_env && big_env[0] >= '0' && big_env[0] <= '9' ? atoi(big_env) : 192;
    if (!no_big && bn_tile == 128 && !out_f32 && a.splits == 1 && (copad & 255) == 0 &&
        (int64_t)((maxM + 255) / 256) * (copad / 256) * a.ncls >= big_min)
        return 256;
    return 128;
}

int igemm_launch(const IgemmArgs& a, int maxM, int bn_tile, int copad, bool out_f32, hipStream_t st) {
    if (igemm_bm(a, maxM, bn_tile, copad, out_f32) == 256) return launch_bn<256, 256, 2, 4>(a, maxM, copad, out_f32, st);
    switch (bn_tile) {
        case 128: return launch_bn<128, 128, 2, 2>(a, maxM, copad, out_f32, st);
        case 64: return launch_bn<128, 64, 2, 2>(a, maxM, copad, out_f32, st);
        case 32: return launch_bn<128, 32, 4, 1>(a, maxM, copad, out_f32, st);
        default: return E_UNSUPPORTED;
    }
}

}  // namespace fmri
