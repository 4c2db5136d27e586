// Stride-2 transposed convolution from 128 to <= 32 channels on MFMA (gfx950): decoder.conv.2 forward and the data
// gradient of discriminator.conv.1 (models/vae_gan.py:112-116, 149-153; ConvTranspose2d k5 s2 p2 as 4 output-parity
// classes with 3x3 / 3x2 / 2x3 / 2x2 unit-shift taps).
//
// With only 32 output channels the generic kernels move 8x more operand bytes per FLOP than on a 256-channel layer
// and run at 15 % of the MFMA peak, so this layer class cost as much as the widest one.  Here
//   * blocks are persistent and class-bound (blocks per class in proportion to its taps), 512 threads = 4 pixel
//     groups x 2 channel chunks: wave (g, h) contracts the 64 input channels of chunk h for the 32 pixels of group g
//     and keeps the WHOLE weight slice it needs (T taps x 64 ch x 32 co) as MFMA fragments in registers;
//   * the input window of an 8x16-pixel tile ((8+TH-1) x (16+TW-1) pixels x 128 ch) is staged through registers
//     into LDS with a 144-byte pixel pitch: fragment reads are conflict-free without a swizzle and every tap is a
//     compile-time byte offset -- the tap loop is ds_read_b128 + MFMA only;
//   * the two chunk partial sums of a pixel group meet once per tile through LDS; the next tile's window is in
//     flight (global -> registers) during the MFMAs; one barrier per tile.
#include "kernels.h"

namespace fmri {

template <int TH, int TW>
__device__ __forceinline__ void tc32_body(const Tc32Args& a, const Tc32Class& c, char* smem, int first, int stride) {
    constexpr int T = TH * TW;
    constexpr int WH = 8 + TH - 1, WW = 16 + TW - 1;
    constexpr int NPIX = WH * WW;
    constexpr int PITCH = 144;
    constexpr int CHB = NPIX * PITCH;               // bytes of one 64-channel chunk of the window
    constexpr int BUF = 2 * CHB;
    constexpr int UNITS = 2 * NPIX * 8;             // 16-B units of a window
    constexpr int NU = (UNITS + 511) / 512;
    constexpr int XCH_OFF = 2 * BUF;                // exchange area: [parity][group][half] x 2 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 1, h = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- this wave's weight slice as MFMA fragments: lane = (row co = frow + 16 tn, k = 8 fq .. +7 of a 32-wide step)
    h8 wf[T][2][2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                wf[t][ks][tn] = *(const h8*)(a.w + c.w_off + (int64_t)(tn * 16 + frow) * c.Kpad + t * 128 + h * 64 +
                                             ks * 32 + fq * 8);

    // ---- staged window units of this thread (fixed for all tiles)
    int upix[NU], uoff[NU], ugl[NU];
#pragma unroll
    for (int e = 0; e < NU; ++e) {
        const int u = e * 512 + tid;
        const int chunk = u / (NPIX * 8);
        const int r = u - chunk * (NPIX * 8);
        upix[e] = u < UNITS ? (r >> 3) : -1;
        uoff[e] = chunk * CHB + (r >> 3) * PITCH + (r & 7) * 16;
        ugl[e] = chunk * 64 + (r & 7) * 8;
    }
    const int dymin = c.dy0 - (TH - 1), dxmin = c.dx0 - (TW - 1);
    const int tpi = c.tiles_y * c.tiles_x;
    h8 stg[NU];
    auto fetch = [&](int t) {
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / c.tiles_x, txi = r - tyi * c.tiles_x;
        const int y0 = tyi * 8 + dymin, x0 = txi * 16 + dxmin;
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int j = upix[e] / WW, i = upix[e] - j * WW;
            const int iy = y0 + j, ix = x0 + i;
            const bool ok = upix[e] >= 0 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const half_t* src = a.in + ((int64_t)(n * a.Hi + iy) * a.Wi + ix) * 128 + ugl[e];
            stg[e] = ok ? *(const h8*)src : (h8)(half_t)0.f;
        }
    };
    auto stash = [&](int buf) {
        char* dst = smem + buf * BUF;
#pragma unroll
        for (int e = 0; e < NU; ++e)
            if (upix[e] >= 0) *(h8*)(dst + uoff[e]) = stg[e];
    };

    // fragment read base of this lane: chunk h, window row 2g (+tm), column frow; tap (ty, tx) of a transposed-conv
    // class samples input (y + dy0 - ty, x + dx0 - tx) = window ((TH-1-ty) + row, (TW-1-tx) + column)
    const int lane_off = h * CHB + ((2 * g) * WW + frow) * PITCH + fq * 16;

    int t = first;
    if (t >= c.ntiles) return;
    fetch(t);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (; t < c.ntiles; t += stride) {
        const int tnext = t + stride;
        const bool more = tnext < c.ntiles;
        if (more) fetch(tnext);                     // global loads in flight during the MFMAs below

        f4 acc[2][2];                               // [tn][tm]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        const char* Ws = smem + cur * BUF + lane_off;
#pragma unroll
        for (int tp = 0; tp < T; ++tp) {
            const int ty = tp / TW, tx = tp - ty * TW;
            const int d = (TH - 1 - ty) * WW + (TW - 1 - tx);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 af[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) af[tm] = *(const h8*)(Ws + (d + tm * WW) * PITCH + ks * 64);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tp][ks][tn], af[tm], acc[tn][tm], 0, 0, 0);
            }
        }

        // ---- the two chunk partial sums of a group meet through LDS: wave h keeps row tm = h and hands over tm = 1-h
        float* xch = (float*)(smem + XCH_OFF + ((cur * 4 + g) * 2) * 2048);
        {
            float* dst = xch + (1 - h) * 512 + lane * 8;          // slot read by the partner
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) *(f4*)(dst + tn * 4) = acc[tn][1 - h];
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        {
            const float* src = xch + h * 512 + lane * 8;
            const int n = t / tpi;
            const int r = t - n * tpi;
            const int tyi = r / c.tiles_x, txi = r - tyi * c.tiles_x;
            const int y = tyi * 8 + 2 * g + h, x = txi * 16 + frow;
            if (y < c.Yc && x < c.Xc) {
                half_t* orow = a.out + ((int64_t)(n * a.Ho + (2 * y + c.oy0)) * a.Wo + (2 * x + c.ox0)) * a.CoStore;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int co = tn * 16 + fq * 4;
                    if (co >= a.CoStore) continue;
                    const f4 o = *(const f4*)(src + tn * 4);
                    h4 hv;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        float f = acc[tn][h][rg] + o[rg];
                        if (co + rg < a.Co) {
                            if (a.bias) f += a.bias[co + rg];
                            if (a.act != ACT_NONE) f = act_apply(f, a.act);
                        } else {
                            f = 0.f;
                        }
                        hv[rg] = (half_t)f;
                    }
                    *(h4*)(orow + co) = hv;
                }
            }
        }
        cur ^= 1;
    }
}

__global__ __launch_bounds__(512) void igemm_tc32_kernel(const Tc32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ci = 0;
    while (ci < 3 && (int)blockIdx.x >= a.cls[ci + 1].block_begin) ++ci;
    const Tc32Class& c = a.cls[ci];
    const int first = blockIdx.x - c.block_begin;
    if (c.TH == 3 && c.TW == 3) tc32_body<3, 3>(a, c, smem, first, c.nblocks);
    else if (c.TH == 3 && c.TW == 2) tc32_body<3, 2>(a, c, smem, first, c.nblocks);
    else if (c.TH == 2 && c.TW == 3) tc32_body<2, 3>(a, c, smem, first, c.nblocks);
    else tc32_body<2, 2>(a, c, smem, first, c.nblocks);
}

int igemm_tc32_launch(const Tc32Args& a, int nblocks, hipStream_t st) {
    const int lds = 2 * 2 * (10 * 18 * 144) + 2 * 4 * 2 * 2048;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_tc32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(igemm_tc32_kernel, dim3(nblocks), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
