// Narrow-channel 5x5 stride-1 convolution on MFMA (gfx950): Ci in {8, 32}, Co <= 32.
//
//   out[n, y, x, co] = act(bias[co] + sum_{tap, ci} in[n, y + dy(tap), x + dx(tap), ci] * w[co][tap*Ci + ci])
//
// These are the image-side layers of the model -- discriminator.conv.0 (3 -> 32) and decoder.conv.3 (32 -> 3),
// forward and data gradient (models/vae_gan.py:118-121, 145-147).  They carry ~2 % of the FLOPs but the generic
// kernels spend as long on them as on a 256-channel layer: with 3..32 channels a K-step is a handful of MFMAs, so
// per-step barriers, DMA waits and address arithmetic dominate.  This kernel removes all of them from the inner loop:
//   * the whole weight matrix (<= 25 K-steps x 16..32 rows) lives in REGISTERS as MFMA operand fragments, loaded
//     once per block; blocks are persistent and walk 16x16-pixel tiles;
//   * the (16+4)^2-pixel input window of a tile is staged through registers into LDS with a padded pixel pitch
//     (80 B for 32 channels) that makes every fragment read conflict-free WITHOUT an address swizzle, so a tap is a
//     compile-time byte offset: the tap loop is ds_read_b128 (immediate offset) + MFMA, no VALU;
//   * the next tile's window is fetched (global -> registers) while the current tile computes; one barrier per tile.
#include "kernels.h"

namespace fmri {

template <int CI, int TN, bool FLIP>
__global__ __launch_bounds__(256, 2) void igemm_narrow_kernel(const NarrowArgs a) {
    constexpr int K = 5, PAD = 2, T = K * K;
    constexpr int WW = 16 + K - 1;                       // window width / height in pixels
    constexpr int NPIX = WW * WW;                        // 400
    constexpr int PITCH = CI == 32 ? 80 : 16;            // bytes per window pixel in LDS
    constexpr int UPP = CI / 8;                          // 16-B units per pixel
    constexpr int WBYTES = NPIX * PITCH;
    constexpr int NU = (NPIX * UPP + 255) / 256;         // staged 16-B units per thread
    constexpr int KS = CI == 32 ? T : (T + 3) / 4;       // k32 steps
    constexpr int TM = 4;                                // wave = 4 rows of 16 pixels
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- weights: B fragments of every K-step in registers.  lane = (row co = frow + 16 tn, k = 8 fq .. +7)
    h8 wf[KS][TN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            wf[ks][tn] = *(const h8*)(a.w + (int64_t)(tn * 16 + frow) * a.Kpad + ks * 32 + fq * 8);

    // ---- staged window units of this thread: unit u -> pixel u / UPP, 16-B chunk u % UPP
    int upix[NU], uoff[NU];
#pragma unroll
    for (int e = 0; e < NU; ++e) {
        const int u = e * 256 + tid;
        upix[e] = u / UPP;
        uoff[e] = upix[e] * PITCH + (u % UPP) * 16;
    }

    // ---- fragment read base: wave row 4*wave + tm, pixel x = frow; tap (ty, tx) adds (ty*WW + tx) pixels.
    // CI == 32: K-step = tap, lane reads channels 8 fq .. of its pixel.  CI == 8: K-step = 4 taps, lane fq reads tap
    // 4 ks + fq (all 8 channels); taps >= 25 meet zero weights, so they may read any valid pixel.
    const int pbase = (4 * wave) * WW + frow;
    int lane_off[CI == 32 ? 1 : KS];
    if (CI == 32) {
        lane_off[0] = pbase * PITCH + fq * 16;
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int tap = 4 * ks + fq;
            if (tap >= T) tap = 0;
            const int ty = tap / K, tx = tap - ty * K;
            const int d = FLIP ? (K - 1 - ty) * WW + (K - 1 - tx) : ty * WW + tx;
            lane_off[ks] = (pbase + d) * PITCH;
        }
    }

    const int tiles_x = a.tiles_x, tpi = a.tiles_y * a.tiles_x;
    // Round 3: window loads and output stores go through buffer descriptors (out-of-image pixels / outside lanes carry the
    // out-of-range offset 0x80000000: the load returns zeros, the store is dropped), so the tile loop has no exec-mask
    // branches around memory instructions -- with them the compiler drained the whole VMEM queue (`s_waitcnt vmcnt(0)`)
    // in front of the stash, i.e. waited for the tile's just-issued stores every iteration.  The stash now also sits in
    // FRONT of the stores, and the activation is one wave-uniform switch per tile instead of one per fragment.
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, (int)((uint32_t)a.N * (uint32_t)a.H * (uint32_t)a.W * (uint32_t)(CI * 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.out, 0, (int)((uint32_t)a.N * (uint32_t)a.H * (uint32_t)a.W * (uint32_t)(a.CoStore * 2)), 0x00020000);
    h8 stg[NU];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / tiles_x, txi = r - tyi * tiles_x;
        const int y0 = tyi * 16 - PAD, x0 = txi * 16 - PAD;
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int j = upix[e] / WW, i = upix[e] - j * WW;
            const int iy = y0 + j, ix = x0 + i;
            const bool ok = upix[e] < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t off = ok ? (uint32_t)((((n * a.H + iy) * a.W + ix) * CI + ((e * 256 + tid) % UPP) * 8) * 2)
                                    : 0x80000000u;
            stg[e] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (int)off, 0, 0));
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * WBYTES;
#pragma unroll
        for (int e = 0; e < NU; ++e)
            if (upix[e] < NPIX) *(h8*)(dst + uoff[e]) = stg[e];
    };

    float bv[TN][4];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int co = tn * 16 + fq * 4 + rg;
            bv[tn][rg] = (a.bias && co < a.Co) ? a.bias[co] : 0.f;        // zeros without a bias: added unconditionally
        }
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));

    int t = blockIdx.x;
    if (t >= a.ntiles) return;
    fetch(t);
    stash(0);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (; t < a.ntiles; t += gridDim.x) {
        const int tn_ = t + gridDim.x;
        const bool more = tn_ < a.ntiles;
        if (more) fetch(tn_);                      // global loads in flight during the MFMAs below

        f4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        const char* Ws = smem + cur * WBYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 af[TM];
            if (CI == 32) {
                const int ty = ks / K, tx = ks - ty * K;
                const int d = FLIP ? (K - 1 - ty) * WW + (K - 1 - tx) : ty * WW + tx;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = *(const h8*)(Ws + lane_off[0] + (d + tm * WW) * PITCH);
            } else {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = *(const h8*)(Ws + lane_off[ks] + tm * WW * PITCH);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // next tile's window -> the other LDS buffer (nobody reads it: its tile finished before the last barrier); in
        // front of this tile's stores, so that the wait for the window loads does not cover them
        if (more) stash(cur ^ 1);

        // ---- epilogue: D[i = co][j = pixel x]; lane owns channels fq*4 .. +3 (+16 tn) of pixel (4*wave + tm, frow)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) acc[tn][tm][rg] += bv[tn][rg];
        if (a.act == ACT_RELU) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[tn][tm][rg] = fmaxf(acc[tn][tm][rg], 0.f);
        } else if (a.act == ACT_TANH) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float v = acc[tn][tm][rg];
                        const float e = __expf(-2.f * fabsf(v));                  // tanh(|x|) = (1 - e) / (1 + e)
                        acc[tn][tm][rg] = copysignf((1.f - e) / (1.f + e), v);
                    }
        } else if (a.act == ACT_SIGMOID) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[tn][tm][rg] = 1.f / (1.f + __expf(-acc[tn][tm][rg]));
        }
        {
            const int n = t / tpi;
            const int r = t - n * tpi;
            const int tyi = r / tiles_x, txi = r - tyi * tiles_x;
            const int x = txi * 16 + frow;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int y = tyi * 16 + 4 * wave + tm;
                const bool inside = y < a.H && x < a.W;
                const uint32_t opix = (uint32_t)(((n * a.H + y) * a.W + x) * a.CoStore * 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int co = tn * 16 + fq * 4;
                    h4 hv;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)(co + rg < a.Co ? acc[tn][tm][rg] : 0.f);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, hv), rsrc_out,
                                                          (int)((inside && co < a.CoStore) ? opix + co * 2 : 0x80000000u), 0, 0);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

template <int CI, int TN>
static int launch_narrow(const NarrowArgs& a, bool flip, hipStream_t st) {
    constexpr int WB = 20 * 20 * (CI == 32 ? 80 : 16);
    int blocks = a.ntiles < 512 ? a.ntiles : 512;
    if (route_probe("fmri::igemm_narrow_kernel<%d,%d,%s>", CI, TN, flip ? "true" : "false")) return OK;
    if (flip) hipLaunchKernelGGL((igemm_narrow_kernel<CI, TN, true>), dim3(blocks), dim3(256), 2 * WB, st, a);
    else hipLaunchKernelGGL((igemm_narrow_kernel<CI, TN, false>), dim3(blocks), dim3(256), 2 * WB, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

int igemm_narrow_launch(const NarrowArgs& a, int ci, int co_tiles, bool flip, hipStream_t st) {
    // 32-bit buffer offsets
    if ((int64_t)a.N * a.H * a.W * ci * 2 >= 0x80000000LL || (int64_t)a.N * a.H * a.W * a.CoStore * 2 >= 0x80000000LL ||
        (a.CoStore & 3))
        return E_UNSUPPORTED;
    if (ci == 32 && co_tiles == 1) return launch_narrow<32, 1>(a, flip, st);
    if (ci == 8 && co_tiles == 1) return launch_narrow<8, 1>(a, flip, st);
    if (ci == 8 && co_tiles == 2) return launch_narrow<8, 2>(a, flip, st);
    return E_UNSUPPORTED;
}

}  // namespace fmri
