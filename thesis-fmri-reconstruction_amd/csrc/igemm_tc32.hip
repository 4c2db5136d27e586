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
//
// Round 3: the epilogue of a row pair is software-pipelined behind the MFMAs of the NEXT row pair.  Round 2's loop
// ran [MFMAs(q) -> exchange -> barrier -> mask loads waited -> stores] four times per tile fully unrolled: every row
// pair exposed a global-memory round trip (the ReLU-mask loads, issued right in front of their use), the four unrolled
// bodies of the four roles were ~100 KB of code for eight waves to loop over, and `acc[tn][1 - h]` with the run-time
// chunk index h became movrel register indexing.  Now the row-pair loop is a real loop (one body per role: ~3 KB),
// stage q issues the mask loads of ITS rows first, runs its MFMAs, then finishes row pair q - 1 (partner's partial sum
// from LDS + the half it kept in registers, mask, store) -- whose mask loads have had a whole MFMA block to land -- and
// only then hands over its own partial sums; 4 barriers per tile instead of 5.  No bias / activation epilogue (no layer
// of this geometry has one: a BatchNorm or a data gradient follows); the launcher routes such calls elsewhere.
template <int CY, int CX, bool MASKED>
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
    // Loads go through a buffer descriptor: padding pixels carry an out-of-range offset and read as zero (no exec
    // branches around the loads).  The per-unit constants (window pixel, byte offset inside the pixel) are recomputed
    // per tile rather than kept in registers next to the weights.
    const int tpi = a.tiles_y * a.tiles_x;
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, (int)((uint32_t)a.N * (uint32_t)a.Hi * (uint32_t)a.Wi * 256u), 0x00020000);
    // two staging phases of NH units: half the window is in flight during row pairs 0-1, the other half during 2-3
    constexpr int NH = (NU + 1) / 2;
    h8 stg[NH];
    auto fetch = [&](int t, int ph) __attribute__((always_inline)) {
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / a.tiles_x, txi = r - tyi * a.tiles_x;
        const int y0 = tyi * 8 - 1, x0 = txi * 16 - 1;            // window origin = tile origin - 1
        // opaque copy of the thread index: otherwise the per-unit constants below are hoisted out of the tile loop into
        // ~20 registers the 9-tap role does not have (it then spills around these very loads)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int u = (ph * NH + e) * 512 + tid_o;
            const int chunk = u / (NPIX * 8);
            const int rr = u - chunk * (NPIX * 8);
            const int pix = rr >> 3;
            const int j = pix / WW, i = pix - j * WW;
            const int iy = y0 + j, ix = x0 + i;
            const bool ok = u < UNITS && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const uint32_t off = ok ? (uint32_t)((((n * a.Hi + iy) * a.Wi + ix) * 128 + chunk * 64 + (rr & 7) * 8) * 2)
                                    : 0x80000000u;
            stg[e] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (int)off, 0, 0));
        }
    };
    auto stash = [&](int buf, int ph) __attribute__((always_inline)) {
        char* dst = smem + buf * BUF;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int u = (ph * NH + e) * 512 + tid_o;
            const int chunk = u / (NPIX * 8);
            const int rr = u - chunk * (NPIX * 8);
            if (u < UNITS) *(h8*)(dst + chunk * CHB + (rr >> 3) * PITCH + (rr & 7) * 16) = stg[e];
        }
    };

    const int lane_off = h * CHB + frow * PITCH + fq * 16;
    const int co_lane = fq * 4;                       // + 16 tn: this lane's 4 output channels of channel tile tn

    // ---- row pair whose epilogue is pending: the half this wave keeps and where it goes (byte offset of the lane's
    // pixel in the output; 0x80000000 = nothing to store: a pixel outside the class grid, or no pending pair yet).
    // The epilogue is BRANCH-FREE: mask loads and stores go through buffer descriptors, whose range check returns zeros /
    // drops the store for the out-of-range offset.  With the loads inside `if (inside)` branches the compiler cannot
    // prove at the loop header that they have been waited for and drains the whole VMEM queue there (`s_waitcnt vmcnt(0)`:
    // the stores of the previous stage and the window prefetch) -- every stage.  The ReLU-mask values of the pending pair
    // are loaded at the START of the stage that finishes it (one MFMA block ahead of their use) and consumed in that same
    // stage: no loaded register is carried around the loop.
    const uint32_t out_bytes = (uint32_t)a.N * (uint32_t)a.Ho * (uint32_t)a.Wo * (uint32_t)a.CoStore * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_msk =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.relu_y, 0, MASKED ? (int)out_bytes : 0, 0x00020000);
    // channel tiles this lane stores (CoStore is 8, 16, 24 or 32: whole 4-channel groups)
    const bool tn_on[2] = {co_lane < a.CoStore, 16 + co_lane < a.CoStore};
    f4 keep[2];
    uint32_t poff = 0x80000000u;
    int pslot = 0;
    keep[0] = keep[1] = (f4){0.f, 0.f, 0.f, 0.f};
    // exchange area: [slot 2][wave pair 4][reader h 2] x 2 KB (lane x (tn 2 x f4))
    float* const xch0 = (float*)(smem + XCH_OFF + pair * 4096);
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    auto load_mask = [&](h4 (&pmk)[2]) __attribute__((always_inline)) {
        if constexpr (MASKED) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                pmk[tn] = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(
                                                     rsrc_msk, (int)(tn_on[tn] ? poff + (tn * 16 + co_lane) * 2 : 0x80000000u), 0, 0));
        } else {
            pmk[0] = pmk[1] = (h4)(half_t)0.f;
        }
    };
    auto finish = [&](const h4 (&pmk)[2]) __attribute__((always_inline)) {
        // partner's partial sums of the pending row pair (visible since the barrier that closed its stage)
        const float* src = xch0 + pslot * 4096 + h * 512 + lane * 8;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int co = tn * 16 + co_lane;
            const f4 o = *(const f4*)(src + tn * 4);
            h4 hv;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float f = co + rg < a.Co ? keep[tn][rg] + o[rg] : 0.f;
                if (MASKED && !((float)pmk[tn][rg] > 0.f)) f = 0.f;
                hv[rg] = (half_t)f;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, hv), rsrc_out,
                                                  (int)(tn_on[tn] ? poff + co * 2 : 0x80000000u), 0, 0);
        }
    };

    int t = blockIdx.x;
    fetch(t, 0);
    stash(0, 0);
    fetch(t, 1);
    stash(0, 1);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (; t < a.ntiles; t += gridDim.x) {
        const int tnext = t + gridDim.x;
        const bool more = tnext < a.ntiles;
        if (more) fetch(tnext, 0);                  // global loads in flight during the MFMAs below
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / a.tiles_x, txi = r - tyi * a.tiles_x;
        const int x = txi * 16 + frow;
        const char* Ws = smem + cur * BUF + lane_off;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {               // class-grid rows 2 q, 2 q + 1 of the tile
            // ReLU-mask values of the PENDING row pair: in flight during this stage's MFMAs
            h4 pmk[2];
            load_mask(pmk);
            // (without the fences the compiler hoists the epilogue below to the top of the MFMA block, right behind the
            // loads it then waits for)
            __builtin_amdgcn_sched_barrier(0);
            f4 acc[2][2];                           // [tn][tm]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
            const char* Wq = Ws + q * (2 * WW * PITCH);
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                const int ty = tp / TW, tx = tp - ty * TW;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 af[2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        af[tm] = *(const h8*)(Wq + ((tm + 2 - ty) * WW + (2 - tx)) * PITCH + ks * 64);
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tn][tm] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tp][ks][tn], af[tm], acc[tn][tm], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- next tile's window: the half fetched two stages ago goes to LDS (in front of this stage's stores, so
            // that the wait for its loads does not wait for them too)
            if (q == 1 && more) stash(cur ^ 1, 0);
            if (q == 3 && more) stash(cur ^ 1, 1);
            // ---- the row pair before this one leaves the block
            finish(pmk);
            if (q == 1 && more) fetch(tnext, 1);    // (behind the stores: the wait for the mask values must not cover it)
            // ---- the two chunk partial sums meet through LDS: wave h keeps row tm = h and hands over row 1 - h
            {
                float* dst = xch0 + (q & 1) * 4096 + (1 - h) * 512 + lane * 8;        // slot read by the partner
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    *(f4*)(dst + tn * 4) = h ? acc[tn][0] : acc[tn][1];
                    keep[tn] = h ? acc[tn][1] : acc[tn][0];
                }
            }
            // where this wave's rows of this pair go
            {
                const int y = tyi * 8 + 2 * q + h;
                const bool inside = y < c.Yc && x < c.Xc;
                poff = inside ? (uint32_t)(((n * a.Ho + (2 * y + CY)) * a.Wo + (2 * x + CX)) * a.CoStore) * 2u : 0x80000000u;
                pslot = q & 1;
            }
            __syncthreads();                        // partial sums handed over; after q == 3: next window in place too
        }
        cur ^= 1;
    }
    {
        h4 pmk[2];
        load_mask(pmk);
        finish(pmk);
    }
}

template <bool MASKED>
__global__ __launch_bounds__(512) void igemm_tc32_kernel(const Tc32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, h = wave & 1;
    // wave w runs on SIMD w % 4: pairs 0 / 2 (9 + 4 taps) share SIMDs 0, 1; pairs 1 / 3 (6 + 6 taps) share SIMDs 2, 3
    if (pair == 0) tc32_role<0, 0, MASKED>(a, smem, h, pair, lane);
    else if (pair == 1) tc32_role<0, 1, MASKED>(a, smem, h, pair, lane);
    else if (pair == 2) tc32_role<1, 1, MASKED>(a, smem, h, pair, lane);
    else tc32_role<1, 0, MASKED>(a, smem, h, pair, lane);
}

// no bias / activation; input and output below 2^32 bytes (32-bit buffer offsets)
int igemm_tc32_launch(const Tc32Args& a, int nblocks, hipStream_t st) {
    if (a.bias != nullptr || a.act != ACT_NONE) return E_UNSUPPORTED;
    if ((int64_t)a.N * a.Ho * a.Wo * a.CoStore * 2 >= 0x80000000LL || (int64_t)a.N * a.Hi * a.Wi * 256 >= 0x80000000LL ||
        (a.CoStore & 3))
        return E_UNSUPPORTED;
    if (route_probe("fmri::igemm_tc32_kernel<%s>", a.relu_y ? "true" : "false")) return OK;
    auto kern = a.relu_y ? igemm_tc32_kernel<true> : igemm_tc32_kernel<false>;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TC32_LDS) != hipSuccess) return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), TC32_LDS, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
