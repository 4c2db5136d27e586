// Stride-2 transposed convolution (k5 p2) from 128 to <= 32 channels on MFMA (gfx950): decoder.conv.2 forward and the
// data gradient of discriminator.conv.1 (models/vae_gan.py:112-116, 149-153).
//
// With only 32 output channels the generic kernels move 8x more operand bytes per FLOP than on a 256-channel layer
// and ran at 15 % of the MFMA peak, so this layer class cost as much as the widest one.  Here
//   * a ConvTranspose2d(k5, s2, p2) is 4 output-parity classes with 3x3 / 3x2 / 2x3 / 2x2 unit-shift taps that all
//     read the SAME (8+2) x (16+2)-pixel input window per 8x16 tile of class-grid positions.  A 512-thread block loads
//     that window once (global -> registers -> LDS, 144-byte pixel pitch: conflict-free fragment reads without a
//     swizzle, every tap a compile-time byte offset) and its 8 waves are (class, 64-channel chunk) specialists: wave
//     (c, h) keeps the whole weight slice of class c and chunk h (T taps x 64 ch x 32 co) as MFMA fragments in
//     REGISTERS and computes all 128 positions of the tile for it -- the tap loop is ds_read_b128 + MFMA only;
//   * the wave order puts a 9-tap and a 4-tap wave, or two 6-tap waves, on every SIMD (13 / 12 tap-units each);
//   * the two chunk partial sums of a class meet through LDS once per two tile rows; blocks are persistent and the next
//     tile's window is in flight during the MFMAs.
#include "kernels.h"

namespace fmri {

namespace {
constexpr int WH = 10, WW = 18, NPIX = WH * WW;    // union window of the four classes
constexpr int PITCH = 144;
constexpr int CHB = NPIX * PITCH;                  // one 64-channel chunk of the window
constexpr int BUF = 2 * CHB;
constexpr int UNITS = 2 * NPIX * 8;                // 16-B units of a window
constexpr int NU = (UNITS + 511) / 512;
constexpr int XCH_OFF = 2 * BUF;                   // exchange: [row-pair parity][wave pair][direction] x 2 KB
constexpr int TC32_LDS = XCH_OFF + 2 * 4 * 2 * 2048;
}  // namespace

// One (class, chunk) specialist.  CY, CX: output parity of the class; taps of a parity-0 dimension sample window
// offsets 2, 1, 0 (3 taps), of a parity-1 dimension 2, 1 (2 taps) relative to the tile origin.
template <int CY, int CX>
__device__ __forceinline__ void tc32_role(const Tc32Args& a, char* smem, int h, int pair, int lane) {
    constexpr int TH = CY ? 2 : 3, TW = CX ? 2 : 3, T = TH * TW;
    const Tc32Class& c = a.cls[CY * 2 + CX];
    const int tid = threadIdx.x;
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

    // ---- window staging: unit u = e*512 + tid -> chunk u / (NPIX*8), pixel (u % (NPIX*8)) >> 3, 16-B slot & 7.
    // Recomputed per tile (a handful of constant divisions) rather than kept in registers next to the weights.
    const int tpi = a.tiles_y * a.tiles_x;
    // One staging phase (all NU units in flight during the whole tile) measured 5 % faster than two half-sized
    // phases although the 9-tap role then spills 7 VGPRs outside its tap loop.
    constexpr int PHASES = 1;
    constexpr int NH = (NU + PHASES - 1) / PHASES;
    h8 stg[NH];
    auto fetch = [&](int t, int ph) {
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / a.tiles_x, txi = r - tyi * a.tiles_x;
        const int y0 = tyi * 8 - 1, x0 = txi * 16 - 1;            // window origin = tile origin - 1
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int u = (ph * NH + e) * 512 + tid;
            const int chunk = u / (NPIX * 8);
            const int rr = u - chunk * (NPIX * 8);
            const int pix = rr >> 3;
            const int j = pix / WW, i = pix - j * WW;
            const int iy = y0 + j, ix = x0 + i;
            const bool ok = u < UNITS && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const half_t* src = a.in + ((int64_t)(n * a.Hi + iy) * a.Wi + ix) * 128 + chunk * 64 + (rr & 7) * 8;
            stg[e] = ok ? *(const h8*)src : (h8)(half_t)0.f;
        }
    };
    auto stash = [&](int buf, int ph) {
        char* dst = smem + buf * BUF;
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int u = (ph * NH + e) * 512 + tid;
            const int chunk = u / (NPIX * 8);
            const int rr = u - chunk * (NPIX * 8);
            if (u < UNITS) *(h8*)(dst + chunk * CHB + (rr >> 3) * PITCH + (rr & 7) * 16) = stg[e];
        }
    };

    const int lane_off = h * CHB + frow * PITCH + fq * 16;

    int t = blockIdx.x;
#pragma unroll
    for (int ph = 0; ph < PHASES; ++ph) {
        fetch(t, ph);
        stash(0, ph);
    }
    __syncthreads();
    int cur = 0;
    for (; t < a.ntiles; t += gridDim.x) {
        const int tnext = t + gridDim.x;
        const bool more = tnext < a.ntiles;
        if (more) fetch(tnext, 0);                  // global loads in flight during the MFMAs below
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / a.tiles_x, txi = r - tyi * a.tiles_x;
        const char* Ws = smem + cur * BUF + lane_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {               // class-grid rows 2 q, 2 q + 1 of the tile
            f4 acc[2][2];                           // [tn][tm]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                const int ty = tp / TW, tx = tp - ty * TW;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 af[2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        af[tm] = *(const h8*)(Ws + ((2 * q + tm + 2 - ty) * WW + (2 - tx)) * PITCH + ks * 64);
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tn][tm] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tp][ks][tn], af[tm], acc[tn][tm], 0, 0, 0);
                }
            }
            // ---- the two chunk partial sums meet through LDS: wave h keeps row tm = h and hands over row 1 - h
            float* xch = (float*)(smem + XCH_OFF + (((q & 1) * 4 + pair) * 2) * 2048);
            {
                float* dst = xch + (1 - h) * 512 + lane * 8;      // slot read by the partner
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) *(f4*)(dst + tn * 4) = acc[tn][1 - h];
            }
            const int x = txi * 16 + frow;
            const int y = tyi * 8 + 2 * q + h;
            const bool inside = y < c.Yc && x < c.Xc;
            const int64_t opix = ((int64_t)(n * a.Ho + (2 * y + CY)) * a.Wo + (2 * x + CX)) * a.CoStore;
            // ReLU backward of the layer below (relu_y = its saved output, geometry of `out`): the mask values are in
            // flight during the exchange
            h4 mk[2];
            if (a.relu_y) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    mk[tn] = (inside && tn * 16 + fq * 4 < a.CoStore) ? *(const h4*)(a.relu_y + opix + tn * 16 + fq * 4)
                                                                       : (h4)(half_t)0.f;
            }
            if (PHASES == 2 && q == 1 && more) { stash(cur ^ 1, 0); fetch(tnext, 1); }
            if (q == 3 && more) stash(cur ^ 1, PHASES - 1);
            __syncthreads();
            const float* src = xch + h * 512 + lane * 8;
            if (inside) {
                half_t* orow = a.out + opix;
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
                        if (a.relu_y && !((float)mk[tn][rg] > 0.f)) f = 0.f;
                        hv[rg] = (half_t)f;
                    }
                    *(h4*)(orow + co) = hv;
                }
            }
        }
        __syncthreads();                            // next tile's window is in place, everyone is done with this one
        cur ^= 1;
    }
}

__global__ __launch_bounds__(512) void igemm_tc32_kernel(const Tc32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, h = wave & 1;
    // wave w runs on SIMD w % 4: pairs 0 / 2 (9 + 4 taps) share SIMDs 0, 1; pairs 1 / 3 (6 + 6 taps) share SIMDs 2, 3
    if (pair == 0) tc32_role<0, 0>(a, smem, h, pair, lane);
    else if (pair == 1) tc32_role<0, 1>(a, smem, h, pair, lane);
    else if (pair == 2) tc32_role<1, 1>(a, smem, h, pair, lane);
    else tc32_role<1, 0>(a, smem, h, pair, lane);
}

int igemm_tc32_launch(const Tc32Args& a, int nblocks, hipStream_t st) {
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)igemm_tc32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TC32_LDS) != hipSuccess) return E_LAUNCH;
    hipLaunchKernelGGL(igemm_tc32_kernel, dim3(nblocks), dim3(512), TC32_LDS, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
