// Stride-2 convolution (k5 p2), Ci % 32 == 0, on MFMA (gfx950): WIDE form of igemm_c5.hip.  One 8-WAVE block owns a CU:
// waves 4-7 LOAD (window slices and weight tiles by buffer_load ... lds with counted vmcnt, the tile arithmetic, the
// output stores and the BatchNorm statistics of the stored values), waves 0-3 COMPUTE (fragment reads and MFMAs only);
// one s_barrier per K-step joins the two roles.  A block's tile is 16 x 16 output pixels of one image (PW = 16) or
// 8 x 8 pixels of four images (PW = 8) x 128 output channels; blocks are persistent over whole rounds of tiles.
//
// Replaces (reference models/vae_gan.py): every Conv2d(k5, s2, p2) forward with >= 32 input channels -- encoder.conv.1/2
// (:73-78), discriminator.conv.1/2/3 (:149-153) -- and every ConvTranspose2d(k5, s2, p2) data gradient (a stride-2
// convolution of the cotangent): decoder.conv.0/1/2 (:112-116).  Declines what needs igemm_c5.hip: the
// BatchNorm-backward epilogue, statistics groups that would share a four-image tile, Wo > 8 with Ho <= 8.
//
// Why (round 3): one wave per SIMD is ISSUE-bound -- ~60 cycles of a wave's instruction stream per
// LDS-DMA instruction, and nothing else on the SIMD to fill them; a finished tile's 64 KB of stores stalled the wave
// that multiplies.  With the roles split
//   * the compute wave's tile is 8 x 4 MFMA tiles (128 pixels x 64 channels: 0.375 ds_read_b128 per MFMA, 64 MFMAs per
//     K-step = two taps x 32 channels); the second tap slot of step t - 1 stays PENDING in registers and its 32 MFMAs
//     run right behind the barrier of step t, covering the first slot's 12 fragment reads; reads are interleaved one
//     per two MFMAs in both halves (sched_group_barrier);
//   * a finished tile goes to the loader waves as fp16 through a 32 KB LDS staging area in two rounds; they store it
//     (256 contiguous bytes per pixel, one or two buffer_store_dwordx4 per K-step of the NEXT tile, each followed by
//     FMRI_STORE_FENCE: common.h) and sum the statistics on the way;
//   * LDS 144-160 KB: two phase windows (10 / 12 slices of 4 KB), the 2 x 16 KB weight ring, the staging area.
// Everything else is igemm_c5.hip's: phase windows with the columns split into their parities (unit-stride conflict-free
// ds_read_b128 with the chunk swizzle 2*bit2(column)), weights straight out of the [co][tap * Ci + ci] matrix, buffer
// descriptor DMA with hardware zero fill, compile-time tap loops, eval-mode BatchNorm folded in (AffEpi).
#include "kernels.h"
#include <type_traits>

#ifndef C5W_NSTG
#define C5W_NSTG 2      // weight ring stages (16 KB each); deeper rings measured no faster
#endif

namespace fmri {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_w<I + 1, N>(f);
    }
}

__device__ __forceinline__ void wdma(v4i srd, uint32_t voff, uint32_t soff, uint32_t lds) {
    srd.x = __builtin_amdgcn_readfirstlane(srd.x);
    srd.y = __builtin_amdgcn_readfirstlane(srd.y);
    srd.z = __builtin_amdgcn_readfirstlane(srd.z);
    srd.w = __builtin_amdgcn_readfirstlane(srd.w);
    soff = __builtin_amdgcn_readfirstlane(soff);
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(srd), "s"(soff), "s"(lds)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmw() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the 25 taps in phase order: 15 taps of the even rows (ky = 0, 2, 4), then 10 of the odd rows (ky = 1, 3)
constexpr int w5_ky(int i) { return i < 15 ? 2 * (i / 5) : 1 + 2 * ((i - 15) / 5); }
constexpr int w5_kx(int i) { return i < 15 ? i % 5 : (i - 15) % 5; }

// ---- DMA schedule of one 32-channel sub-chunk (13 K-steps), ring of NSTG weight stages filled D = NSTG - 1 steps ahead.
// Issued behind the barrier of step t, in this order: the 4 weight pieces of step t + D (of the next sub-chunk past step 12),
// then window slices: the NSL slices of this sub-chunk's odd-row window (buffer 1, first read in step 7) over steps 0-4
// (2 each of 10; 3, 3, 2, 2, 2 of 12), those of the next sub-chunk's even-row window (buffer 0, last read in step 7) over steps
// 8-11 (3, 3, 2, 2 of 10; 3 each of 12).
constexpr int w5_od_n(int t, int NSL) { return t <= 4 ? NSL / 5 + (t < NSL % 5 ? 1 : 0) : 0; }
constexpr int w5_od_0(int t, int NSL) { return t * (NSL / 5) + (t < NSL % 5 ? t : NSL % 5); }
constexpr int w5_ev_n(int t, int NSL) { return (t >= 8 && t <= 11) ? NSL / 4 + (t - 8 < NSL % 4 ? 1 : 0) : 0; }
constexpr int w5_ev_0(int t, int NSL) { return (t - 8) * (NSL / 4) + (t - 8 < NSL % 4 ? t - 8 : NSL % 4); }
constexpr int w5_s(int t, bool more, int NSL) { return t <= 4 ? w5_od_n(t, NSL) : (more ? w5_ev_n(t, NSL) : 0); }
constexpr int w5_w(int t, bool more, int D) { return (t + D < 13 || more) ? 4 : 0; }
// pieces that may still be in flight at the barrier of step t (vmcnt counts in issue order): the weights of step t were
// issued at step t - D; index i < 0 = step i + 13 of the previous sub-chunk (which had a successor)
constexpr int w5_allow(int t, bool more, int D, int NSL) {
    int n = t - D < 0 ? w5_s(t - D + 13, true, NSL) : w5_s(t - D, more, NSL);
    for (int i = t - D + 1; i < t; ++i) n += i < 0 ? 4 + w5_s(i + 13, true, NSL) : w5_w(i, more, D) + w5_s(i, more, NSL);
    if (t == 7) {                       // the odd-row window (last slices issued at step 4) is read from here on
        int m = 0;
        for (int i = 5; i < 7; ++i) m += w5_w(i, more, D) + w5_s(i, more, NSL);
        n = m < n ? m : n;
    }
    if (t == 0) n = 4 < n ? 4 : n;      // the even-row window (last slices at step 11 of the previous sub-chunk) is read from here on
    return n;
}

// output stores of the previous tile issued behind the DMA of step t of a tile's first sub-chunk: 16 items over steps 0-11
constexpr int w5_nst(int t) { return t < 4 ? 2 : (t < 12 ? 1 : 0); }
constexpr int w5_st0(int t) { return t < 4 ? 2 * t : 8 + (t - 4); }

}  // namespace

// PW: 16 = tiles of 16 x 16 pixels of one image, 8 = 8 x 8 pixels of FOUR images (an MFMA row tile = one tile row of two
// images).  STATS: 0 none, 1 BatchNorm forward statistics (StatEpi)
template <int PW, int STATS>
__global__ __launch_bounds__(512, 1) void igemm_c5w_kernel(const C5Args a) {
    constexpr int BN = 128, WN = 2, TM = 8, TN = 4;
    constexpr int PH = PW, IPB = PW == 16 ? 1 : 4;
    constexpr int IH = PH + 2;                            // window rows per image
    constexpr int ROW = 2 * PW + 3;                       // window pixels per row: plane 0 (PW + 2), plane 1 (PW + 1)
    constexpr int NSL = (IPB * IH * ROW * 64 + 4095) / 4096;      // 4 KB DMA slices per window: 10 (40 320 B) / 12 (48 640 B)
    constexpr int WINB = NSL * 4096;
    constexpr int W_BYTES = 2 * 8192;                     // two tap slots of [128 co][32 ch]
    constexpr int WBUF0 = 2 * WINB;
    constexpr int NSTG = C5W_NSTG, D = NSTG - 1;          // weight ring stages, K-steps of weights in flight
    constexpr int STG0 = WBUF0 + NSTG * W_BYTES;          // output staging: 128 pixels x 128 channels fp16 (32 KB)
    static_assert(STG0 + 32768 <= 160 * 1024, "LDS");
    static_assert(PW == 16 || PW == 8, "tile");
    static_assert(WINB + (2 + 7) * ROW * 64 < 65536, "ds_read immediates");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= 4;                        // waves 4-7 move bytes, waves 0-3 multiply
    int bx, by;
    xcd_tile(bx, by);
    const int tile0 = bx * a.tpb;
    if (tile0 >= a.ntiles) return;
    const int tile1 = tile0 + a.tpb < a.ntiles ? tile0 + a.tpb : a.ntiles;
    const int co0 = by * BN;
    const int nsub = a.nsub;

    // ---- tile -> (image, tile origin)
    const int tpi = a.tiles_y * a.tiles_x;
    auto tile_geom = [&](int tile, int& g, int& yy, int& xx) __attribute__((always_inline)) {
        g = (int)fd_div((uint32_t)tile, a.fdTPI);
        const int trem = tile - g * tpi;
        const int tyi = (int)fd_div((uint32_t)trem, a.fdTX);
        yy = tyi * PH;
        xx = (trem - tyi * a.tiles_x) * PW;
    };
    // BatchNorm statistics: summed by the LOADER waves over the fp16 values they keep for the output stores (they idle
    // between DMA issues; in the compute waves the same sums cost ~1 500 instructions per tile, +50 % on a 13-step tile).
    // Loader thread tid holds channels co0 + 8*slot .. +7 (slot = (tid & 15) ^ (tid >> 4)) of pixel tid >> 4 of every item
    float lsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lsq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    int grp0, ty0, tx0;
    tile_geom(tile0, grp0, ty0, tx0);
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? (grp0 * IPB) / a.st.group_n : 0;   // statistics group of the block's tiles

    if (loader) {
        // =====================================================================================================
        // loader waves: per K-step, wait for the operands of this step, meet the compute waves at the barrier (they are
        // done with the ring stage and window rows about to be overwritten), issue the six DMA pieces of the step
        // =====================================================================================================
        const int tid = threadIdx.x & 255;
        const int lw = wave - 4;
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        v4i srd_in, srd_w;
        srd_in.x = (int)(uint32_t)(uintptr_t)a.in;
        srd_in.y = (int)(uint32_t)((uintptr_t)a.in >> 32);
        srd_in.z = (int)a.in_bytes;
        srd_in.w = 0x00020000;
        srd_w.x = (int)(uint32_t)(uintptr_t)a.w;
        srd_w.y = (int)(uint32_t)((uintptr_t)a.w >> 32);
        srd_w.z = (int)a.w_bytes;
        srd_w.w = 0x00020000;

        // ---- window DMA: 16-B unit q = e*256 + tid of a window buffer holds channels 8*cc .. 8*cc+7 (of the 32-channel
        // sub-chunk) of window pixel p = q >> 2 = image ip, row j, column position ii; cc = (q & 3) ^ 2*bit2(ii) (PW = 16) or
        // (q & 3) ^ 2*(ip & 1) (PW = 8): the fragment reads are bank-conflict free with these.
        // Column position ii < PW + 2: input column 2*x0 - 2 + 2*ii; else 2*x0 - 1 + 2*(ii - PW - 2).  Row j of phase rp:
        // input row 2*y0 - 2 + rp + 2*j (the odd phase has PH + 1 rows).
        uint32_t soff0[NSL], soff1[NSL];
        uint32_t wstat[NSL];                  // column term | row << 8 | cc << 13 | valid << 15 | image << 16
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            const int q = e * 256 + tid;
            const int p = q >> 2;
            const int ip = p / (IH * ROW);
            const int pr = p - ip * (IH * ROW);
            const int j = pr / ROW;
            const int ii = pr - j * ROW;
            const int cp = ii >= PW + 2 ? 1 : 0;
            const int m = ii - cp * (PW + 2);
            const int cc = (q & 3) ^ ((PW == 16 ? ((ii >> 2) & 1) : (ip & 1)) << 1);
            const int valid = ip < IPB ? 1 : 0;
            wstat[e] = (uint32_t)((2 * m + cp) | ((j & 31) << 8) | (cc << 13) | (valid << 15) | ((ip & 3) << 16));
        }
        auto tile_offsets = [&](int g, int yy, int xx) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < NSL; ++e) {
                soff0[e] = soff1[e] = 0x80000000u;             // out of range -> the DMA writes zeros
                const uint32_t ws = wstat[e];
                const int j = (ws >> 8) & 31, cc = (ws >> 13) & 3;
                const int n = g * IPB + (int)((ws >> 16) & 3);
                const int ix = 2 * xx - 2 + (int)(ws & 255);
                const int iy = 2 * yy - 2 + 2 * j;
                if (((ws >> 15) & 1) && n < a.N && (unsigned)ix < (unsigned)a.Wi) {
                    const uint32_t o = (uint32_t)((((n * a.Hi + iy) * a.Wi + ix) * a.Ci + cc * 8) * 2);
                    if ((unsigned)iy < (unsigned)a.Hi) soff0[e] = o;
                    if (j < PH + 1 && (unsigned)(iy + 1) < (unsigned)a.Hi) soff1[e] = o + (uint32_t)(a.Wi * a.Ci * 2);
                }
            }
        };
        tile_offsets(grp0, ty0, tx0);
        const uint32_t lds_wave = lds0 + lw * 1024;
        // slice E of sub-chunk `sub`, phase RP, into window buffer RP
        auto load_slice = [&](auto RP_, int sub, auto E_) __attribute__((always_inline)) {
            constexpr int rp = decltype(RP_)::value, e = decltype(E_)::value;
            if constexpr (e < NSL) wdma(srd_in, rp ? soff1[e] : soff0[e], (uint32_t)sub * 64u, lds_wave + rp * WINB + e * 4096);
        };
        // ---- weight DMA: tap slot = [128 co][32 ch] = 8 KB, 64 rows per instruction of the four waves; chunk swizzle
        // 2*bit2(row)
        const int trow = tid >> 2;
        const int wcc = (tid & 3) ^ (((trow >> 2) & 1) << 1);
        const uint32_t vw = (uint32_t)(((co0 + trow) * a.Kpad + wcc * 8) * 2);
        const uint32_t rs64 = (uint32_t)(a.Kpad * 128);      // 64 rows
        const int Ci2 = a.Ci * 2;
        // piece PC (tap slot * 2 + 64-row half) of taps [T0, T0 + 2) (those < 25) of sub-chunk `sub` into ring stage stg
        auto load_w_piece = [&](int stg, auto T0_, auto PC_, int sub) __attribute__((always_inline)) {
            constexpr int t0 = decltype(T0_)::value, pc = decltype(PC_)::value;
            constexpr int s = pc >> 1, i = pc & 1;
            if constexpr (t0 + s < 25) {
                constexpr int tap = w5_ky(t0 + s) * 5 + w5_kx(t0 + s);
                const uint32_t so = (uint32_t)(tap * Ci2 + sub * 64);
                wdma(srd_w, vw, so + i * rs64, lds_wave + WBUF0 + stg * W_BYTES + s * 8192 + i * 4096);
            }
        };
        // ---- outputs: the compute waves hand a finished tile over through LDS in two rounds (tile rows 0-7, 8-15; pixel
        // p = row * 16 + x of the round at p * 256 B, 16-B channel slot s at slot s ^ x); thread tid keeps item k = row k,
        // x = tid >> 4, slot tid & 15 of both rounds in registers and stores them one or two per K-step of the NEXT tile's
        // first sub-chunk (the last tile: at the end): 256 contiguous bytes per pixel, and no store ever waits in a compute wave
        const __amdgpu_buffer_rsrc_t srd_out = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.out, 0, (int)((uint32_t)a.N * (uint32_t)a.Ho * (uint32_t)a.Wo * (uint32_t)a.CoStore * 2u), 0x00020000);
        // (PW = 8: a round = two images, item k = tile row k of both, pixel tid >> 4 = image (bit 3), column)
        const uint32_t row_b = (uint32_t)(a.Wo * a.CoStore * 2);
        v4i oreg[2][8];
        uint32_t ovo[2] = {0x80000000u, 0x80000000u};
        int onrow[2] = {0, 0};
        auto take = [&](int g, int yy, int xx) __attribute__((always_inline)) {
            const int opx = tid >> 4;                      // pixel of the item
            const int slot = (tid & 15) ^ opx;
            const int ox = PW == 16 ? opx : (opx & 7);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                __builtin_amdgcn_s_barrier();                      // the round is in LDS
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) oreg[r][k] = *(const v4i*)(smem + STG0 + tid * 16 + k * 4096);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // ... and in registers: the staging area is free
                __builtin_amdgcn_sched_barrier(0);
                const int n = PW == 16 ? g : g * IPB + r * 2 + (opx >> 3);
                const int yb = PW == 16 ? yy + r * 8 : yy, x = xx + ox;
                const bool ok = n < a.N && x < a.Wo && co0 + slot * 8 < a.CoStore;
                ovo[r] = ok ? (uint32_t)((((n * a.Ho + yb) * a.Wo + x) * a.CoStore + co0 + slot * 8) * 2) : 0x80000000u;
                onrow[r] = a.Ho - yb;                              // tile rows k < onrow exist
            }
        };
        // item I of the kept tile (always issued, so that vmcnt counts stay compile-time: rows below the image carry an
        // out-of-range offset and are dropped)
        auto put = [&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, r = i >> 3, k = i & 7;
            const uint32_t vt = k < onrow[r] ? ovo[r] : 0x80000000u;
            if constexpr (STATS == 1) {
                // statistics of the STORED (fp16-rounded) values: what the consumers and the BN backward see
                const h8 hv = __builtin_bit_cast(h8, oreg[r][k]);
                const bool in = vt != 0x80000000u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = in ? (float)hv[j] : 0.f;
                    lsum[j] += f;
                    lsq[j] += f * f;
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, oreg[r][k]), srd_out, (int)vt, (int)(k * row_b), 0);
            FMRI_STORE_FENCE();        // SGPR-offset store: the compiler pads no wait states (common.h)
        };
        int stg = 0;                              // ring stage of the current step (wave-uniform)
        // one 32-channel sub-chunk: 13 steps (schedule: w5_allow's comment); `outs`: the previous tile's outputs leave
        // behind the DMA of steps 0-11 (and may stay in flight over the next barrier, like the window slices)
        auto feed_sub = [&](int sub, bool more, int nsubi, bool switch_tile, int ng, int ny, int nx, bool outs)
                            __attribute__((always_inline)) {
            static_for_w<0, 13>([&](auto T_) __attribute__((always_inline)) {
                constexpr int t = decltype(T_)::value;
                constexpr int n_more = w5_allow(t, true, D, NSL), n_last = w5_allow(t, false, D, NSL);
                constexpr int ns_prev = t > 0 ? w5_nst(t - 1) : 0;
                static_assert(D == 1 || C5W_NSTG == 2, "the store allowance below assumes one step of weights in flight");
                if (outs) {
                    if (more) wait_vmw<n_more + ns_prev>(); else wait_vmw<n_last + ns_prev>();
                } else {
                    if (more) wait_vmw<n_more>(); else wait_vmw<n_last>();
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                int sd = stg + D;
                if (sd >= NSTG) sd -= NSTG;
                static_for_w<0, 4>([&](auto K_) __attribute__((always_inline)) {
                    if constexpr (t + D < 13) {
                        load_w_piece(sd, std::integral_constant<int, 2 * (t + D)>{}, K_, sub);
                    } else {
                        if (more) load_w_piece(sd, std::integral_constant<int, 2 * (t + D - 13)>{}, K_, nsubi);
                    }
                });
                if constexpr (t <= 4) {
                    static_for_w<0, w5_od_n(t, NSL)>([&](auto J_) __attribute__((always_inline)) {
                        load_slice(std::integral_constant<int, 1>{}, sub, std::integral_constant<int, w5_od_0(t, NSL) + decltype(J_)::value>{});
                    });
                } else if constexpr (t == 6) {
                    if (switch_tile) tile_offsets(ng, ny, nx);
                } else if constexpr (w5_ev_n(t, NSL) > 0) {
                    if (more)
                        static_for_w<0, w5_ev_n(t, NSL)>([&](auto J_) __attribute__((always_inline)) {
                            load_slice(std::integral_constant<int, 0>{}, nsubi, std::integral_constant<int, w5_ev_0(t, NSL) + decltype(J_)::value>{});
                        });
                }
                if constexpr (w5_nst(t) > 0) {
                    if (outs) static_for_w<w5_st0(t), w5_st0(t) + w5_nst(t)>([&](auto I_) __attribute__((always_inline)) { put(I_); });
                }
                stg = stg + 1 == NSTG ? 0 : stg + 1;
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // prologue: even-row window of the first tile's sub-chunk 0 and the weights of the first D steps, all landed
        static_for_w<0, NSL>([&](auto E_) __attribute__((always_inline)) { load_slice(std::integral_constant<int, 0>{}, 0, E_); });
        static_for_w<0, D>([&](auto S_) __attribute__((always_inline)) {
            static_for_w<0, 4>([&](auto K_) __attribute__((always_inline)) {
                load_w_piece(decltype(S_)::value, std::integral_constant<int, 2 * decltype(S_)::value>{}, K_, 0);
            });
        });
        wait_vmw<0>();
        int tile = tile0, sub = 0;
        int cg = grp0, cy = ty0, cx = tx0;        // the tile being computed
        int ng = 0, ny = 0, nx = 0;
        if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
        while (tile < tile1) {
            const bool next_tile = tile + 1 < tile1;
            const bool last_sub = sub + 1 >= nsub;
            feed_sub(sub, !last_sub || next_tile, last_sub ? 0 : sub + 1, last_sub && next_tile, ng, ny, nx, sub == 0 && tile > tile0);
            ++sub;
            if (last_sub) {
                take(cg, cy, cx);
                cg = ng; cy = ny; cx = nx;
                sub = 0;
                ++tile;
                if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
            }
        }
        static_for_w<0, 16>([&](auto I_) __attribute__((always_inline)) { put(I_); });
    } else {
        // =====================================================================================================
        // compute waves: LDS fragment reads and MFMAs only
        // =====================================================================================================
        int grp = grp0, y0 = ty0, x0 = tx0;
        // ---- A fragment addresses: abase[kx] for row tile 0 of the wave (PW = 16: tile row wm*8, lane = column; PW = 8: tile
        // row 0 of images wm*2 + (frow >> 3), lane & 7 = column); row tile tm adds tm * ROW * 64, a window row shift sy adds
        // sy * ROW * 64: immediates
        uint32_t abase[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const int x = PW == 16 ? frow : (frow & 7);
            const int ii = (kx & 1) * (PW + 2) + x + (kx >> 1);
            const int p = (PW == 16 ? (wm * 8) * ROW : (wm * 2 + (frow >> 3)) * IH * ROW) + ii;
            const int swz = PW == 16 ? ((ii >> 2) & 1) : (frow >> 3);
            abase[kx] = (uint32_t)((p << 6) + ((fq ^ (swz << 1)) << 4));
        }
        // ---- B fragment address (row = wn*64 + tn*16 + frow)
        const uint32_t boff = (uint32_t)(WBUF0 + (wn * (BN / WN) + frow) * 64 + ((fq ^ (((frow >> 2) & 1) << 1)) << 4));

        f4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

        // ---- the pending second tap slot of the previous K-step (all zeros: nothing pending)
        h8 paf[TM], pbf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) paf[i] = (h8)(half_t)0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i) pbf[i] = (h8)(half_t)0.f;
        auto pending_mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pbf[tn], paf[tm], acc[tn][tm], 0, 0, 0);
        };
        auto clear_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) paf[i] = (h8)(half_t)0.f;
#pragma unroll
            for (int i = 0; i < TN; ++i) pbf[i] = (h8)(half_t)0.f;
        };

        // one K-step: taps T0, T0 + 1 (those < 25), weights in ring stage STG.  The first slot's 12 fragment reads go out
        // first and land under the pending slot's 32 MFMAs; the second slot's reads sit between the first slot's MFMAs
        auto step = [&](auto T0_, uint32_t wb) __attribute__((always_inline)) {
            constexpr int t0 = decltype(T0_)::value;
            constexpr int NS = t0 + 1 < 25 ? 2 : 1;
            h8 af0[TM], bf0[TN];
            {
                constexpr int ky = w5_ky(t0), kx = w5_kx(t0);
                const char* Ps = smem + (ky & 1) * WINB + (ky >> 1) * (ROW * 64);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) af0[tm] = *(const h8*)(Ps + abase[kx] + tm * (ROW * 64));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bf0[tn] = *(const h8*)(smem + (wb + tn * 1024));
            }
            // (t0 == 0: the previous step was the lone 25th tap or the start of the kernel -- nothing is pending, and 32 MFMAs
            // on zeros would only cover the first slot's read latency at twice its price)
            if constexpr (t0 != 0) pending_mfmas();
            if constexpr (t0 != 0) {
                // the reads one by one between the first MFMAs, not as a burst in front of them
#pragma unroll
                for (int i = 0; i < TM + TN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - 2 * (TM + TN), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NS == 2) {
                constexpr int ky = w5_ky(t0 + 1), kx = w5_kx(t0 + 1);
                const char* Ps = smem + (ky & 1) * WINB + (ky >> 1) * (ROW * 64);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) paf[tm] = *(const h8*)(Ps + abase[kx] + tm * (ROW * 64));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) pbf[tn] = *(const h8*)(smem + (wb + 8192 + tn * 1024));
            } else {
                clear_pending();
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[tn], af0[tm], acc[tn][tm], 0, 0, 0);
            if constexpr (NS == 2) {
#pragma unroll
                for (int i = 0; i < TM + TN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - 2 * (TM + TN), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int stg = 0;                              // ring stage of the current step (wave-uniform)
        auto run_sub = [&]() __attribute__((always_inline)) {
            static_for_w<0, 13>([&](auto T_) __attribute__((always_inline)) {
                constexpr int t = decltype(T_)::value;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                step(std::integral_constant<int, 2 * t>{}, boff + (uint32_t)(stg * W_BYTES));
                stg = stg + 1 == NSTG ? 0 : stg + 1;
            });
        };

        // ---- a finished tile: D[i = co][j = tile pixel] as fp16 into the staging area (no bias / activation: BatchNorm or a
        // data gradient follows), tile rows 0-7 (the wm = 0 waves) then 8-15 (wm = 1); the loader waves take each round into
        // registers and store it (see `take`).  Pixel p = tm * 16 + frow at p * 256 B, 16-B channel slot s = wn * 8 + tn * 2
        // + (fq >> 1) at slot s ^ frow (spreads the 16 pixels of a write over the banks).  Channels >= Co are written as zeros.
        const bool full_co = ((a.Co | a.CoStore) & 127) == 0;
        const uint32_t sbase = (uint32_t)(STG0 + frow * 256 + (((wn * 8 + (fq >> 1)) ^ frow) << 4) + (fq & 1) * 8);
        auto hand_body = [&](auto FULL_) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(FULL_)::value;
            // PW = 16: pixel (y0 + wm*8 + tm, x0 + frow) of image grp; PW = 8: pixel (y0 + tm, x0 + (frow & 7)) of image
            // grp*4 + wm*2 + (frow >> 3); pixels outside the image are handed over too (the loader waves drop them)
            const int cw = co0 + wn * (BN / WN) + fq * 4;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int co = cw + tn * 16;
                // AffEpi (eval-mode BatchNorm of the consumer folded in): scale / shift of the lane's four channels
                f4 asc = (f4){1.f, 1.f, 1.f, 1.f}, ash = (f4){0.f, 0.f, 0.f, 0.f};
                const bool aff = STATS == 0 && a.aff.scale != nullptr;
                if (aff && co < a.CoStore) {
                    asc = *(const f4*)(a.aff.scale + co);
                    ash = *(const f4*)(a.aff.shift + co);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    f4 v = acc[tn][tm];
                    if (aff) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            v[rg] = v[rg] * asc[rg] + ash[rg];
                            if (a.aff.relu) v[rg] = fmaxf(v[rg], 0.f);
                        }
                    }
                    h4 hv;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)((FULL || co + rg < a.Co) ? v[rg] : 0.f);
                    *(h4*)(smem + ((sbase ^ (uint32_t)(tn * 32)) + tm * 4096)) = hv;
                }
            }
        };
        auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (wm == r) {
                    if (full_co) hand_body(std::true_type{});
                    else hand_body(std::false_type{});
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // the round is in LDS
                __builtin_amdgcn_s_barrier();                      // ... and in the loader waves' registers
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        int tile = tile0, sub = 0;
        int ng = 0, ny = 0, nx = 0;
        if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
        while (tile < tile1) {
            const bool last_sub = sub + 1 >= nsub;
            run_sub();
            ++sub;
            if (last_sub) {
                epilogue();
                clear_pending();
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
                grp = ng; y0 = ny; x0 = nx;
                sub = 0;
                ++tile;
                if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
            }
        }
    }
    if constexpr (STATS != 0) {
        // one row per block; the blocks of one statistics group are contiguous (tpg[0] = blocks per group).  The 16 loader
        // threads that hold the same 8 channels (one per pixel column of an item) meet in LDS
        float* scratch = (float*)smem;
        __syncthreads();                                     // everyone is done with the operand tiles
        if (loader) {
            const int t = threadIdx.x & 255;
            float* dst = scratch + ((((t & 15) ^ (t >> 4)) * 16 + (t >> 4)) * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dst[j] = lsum[j]; dst[8 + j] = lsq[j]; }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x;                       // channel co0 + c = slot c >> 3, element c & 7
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                const float* src = scratch + ((c >> 3) * 16 + px) * 16 + (c & 7);
                s0 += src[0];
                s1 += src[8];
            }
            float* row = a.st.part + ((size_t)sgrp * a.st.rows_cap + (bx - sgrp * a.st.tpg[0])) * 2 * a.st.C;
            if (co0 + c < a.st.C) {
                row[co0 + c] = s0;
                row[a.st.C + co0 + c] = s1;
            }
        }
    }
}

template <int PW, int STATS>
static int launch_c5w(const C5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_c5w_kernel<PW, STATS>;
    constexpr int lds = 2 * (PW == 16 ? 10 : 12) * 4096 + C5W_NSTG * 16384 + 32768;
    if (route_probe("fmri::igemm_c5w_kernel<%d,%d>", PW, STATS)) return OK;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.ntiles + a.tpb - 1) / a.tpb, copad / 128, 1), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// k5 s2 p2, Ci % 32 == 0, 128-channel tiles, 16 x 16-pixel tiles of one image, no bias / activation, no BnBwdEpi.
// C5Args with tiles_y = ceil(Ho / 16), tiles_x = ceil(Wo / 16), ntiles = N * tiles_y * tiles_x.
int igemm_c5w_launch(const C5Args& a, int copad, hipStream_t st) {
    if (a.nsub < 1 || (copad & 127) || a.ntiles < 1 || a.tpb < 1 || a.bb.x || (a.aff.scale && a.st.part)) return E_UNSUPPORTED;
    if ((int64_t)a.N * a.Ho * a.Wo * a.CoStore * 2 >= 0x7fffffffLL) return E_UNSUPPORTED;      // 32-bit store offsets
    if (a.pw16) return a.st.part ? launch_c5w<16, 1>(a, copad, st) : launch_c5w<16, 0>(a, copad, st);
    return a.st.part ? launch_c5w<8, 1>(a, copad, st) : launch_c5w<8, 0>(a, copad, st);
}


}  // namespace fmri
