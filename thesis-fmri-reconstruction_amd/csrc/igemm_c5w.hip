// Stride-2 convolution (k5 p2) with >= 128 input channels on MFMA (gfx950), WIDE form of igemm_c5.hip: one 4-wave block
// per CU owns 16 x 16 output pixels x 128 channels (wave tile 128 pixels x 64 channels) and its K-step is pipelined by hand.
//
// Replaces (reference models/vae_gan.py): the forward of discriminator.conv.2 (:149-153) and the data gradient of
// decoder.conv.1 (:112-116 -- a stride-2 convolution of the 128-channel cotangent), i.e. the Ci >= 128, Wo >= 16
// launches of igemm_c5.hip.
//
// Why (round 3): igemm_c5 with ONE of its two blocks per CU resident still delivers 76 % of its
// throughput -- a lone wave per SIMD runs the K-step in ~1 500 cycles of which 512 are its MFMAs, the rest is the
// step's fixed chain (barrier, operand DMA round trip, fragment-read latency), and the second resident block only adds
// 30 %.  Here the fixed chain is paid per 64 MFMAs instead of 32: the wave tile is 8 x 4 MFMA tiles (0.375 LDS fragment
// reads per MFMA instead of 0.5; 22 KB of operand DMA per 256 MFMAs of the block instead of 20 KB per 128), and the
// wave's own instruction stream keeps the matrix pipe fed across the chain:
//   * the second tap slot of step t - 1 stays PENDING (fragments in registers); its 32 MFMAs run right behind the
//     barrier of step t with the step's six DMA instructions issued one by one between them and the first slot's 12
//     fragment reads in front (they land under those MFMAs); then the first slot's 32 MFMAs with the second slot's
//     reads between them;
//   * one block per CU: 512 registers per lane (no pressure from the 128 accumulators), 112 KB of LDS (two 40 KB phase
//     windows of (16 + 2) x 35 pixels x 32 channels, 2 x 16 KB weight ring).
// Everything else is igemm_c5.hip's: phase windows with the columns split into their parities (unit-stride conflict-free
// ds_read_b128 with the chunk swizzle 2*bit2(column)), weights straight out of the [co][tap * Ci + ci] matrix, buffer
// descriptor DMA with hardware zero fill, counted vmcnt, compile-time tap loops, persistent blocks over consecutive
// tiles, BatchNorm forward statistics of the stored values (StatEpi).
#include "kernels.h"
#include <type_traits>

namespace fmri {

#ifdef FMRI_STAMP
// Diagnostic build only (tools/probes/c5w_stamps.py; never shipped): [0] sync (vmcnt + barrier), [1] pending-slot phase
// (12 fragment reads, 32 MFMAs, 6 DMA pieces), [2] first-slot phase, [3] epilogue, [4] steps, [5] waves, [6] kernel
// cycles (s_memtime), [7] kernel 100 MHz ticks (s_memrealtime)
__device__ unsigned long long c5w_stamp_acc[8];
#define FMRI_STAMP_AT(v)                                                                     \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

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
    static_assert(N >= 0 && N <= 15, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the 25 taps in phase order: 15 taps of the even rows (ky = 0, 2, 4), then 10 of the odd rows (ky = 1, 3)
constexpr int w5_ky(int i) { return i < 15 ? 2 * (i / 5) : 1 + 2 * ((i - 15) / 5); }
constexpr int w5_kx(int i) { return i < 15 ? i % 5 : (i - 15) % 5; }

}  // namespace

// STATS: 0 none, 1 BatchNorm forward statistics (StatEpi)
template <int STATS>
__global__ __launch_bounds__(256, 1) void igemm_c5w_kernel(const C5Args a) {
    constexpr int BN = 128, WM = 2, WN = 2, TM = 8, TN = 4;
    constexpr int PW = 16, PH = 16;
    constexpr int ROW = 2 * PW + 3;                       // window pixels per row: plane 0 (PW + 2), plane 1 (PW + 1)
    constexpr int NSL = 10;                               // 4 KB DMA slices per window ((PH + 2) * ROW * 64 B = 40 320)
    constexpr int WINB = NSL * 4096;
    constexpr int W_BYTES = 2 * 8192;                     // two tap slots of [128 co][32 ch]
    constexpr int WBUF0 = 2 * WINB;
    static_assert((PH + 2) * ROW * 64 <= WINB, "window fits its slices");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int tile0 = bx * a.tpb;
    if (tile0 >= a.ntiles) return;
    const int tile1 = tile0 + a.tpb < a.ntiles ? tile0 + a.tpb : a.ntiles;
    const int co0 = by * BN;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- tile -> (image, tile origin)
    const int tpi = a.tiles_y * a.tiles_x;
    int grp, y0, x0;                      // of the tile being computed (epilogue)
    auto tile_geom = [&](int tile, int& g, int& yy, int& xx) __attribute__((always_inline)) {
        g = (int)fd_div((uint32_t)tile, a.fdTPI);
        const int trem = tile - g * tpi;
        const int tyi = (int)fd_div((uint32_t)trem, a.fdTX);
        yy = tyi * PH;
        xx = (trem - tyi * a.tiles_x) * PW;
    };
    tile_geom(tile0, grp, y0, x0);

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
    // sub-chunk) of window pixel p = q >> 2 = row j, column position ii; cc = (q & 3) ^ 2*bit2(ii).
    // Column position ii < PW + 2: input column 2*x0 - 2 + 2*ii; else 2*x0 - 1 + 2*(ii - PW - 2).  Row j of phase rp:
    // input row 2*y0 - 2 + rp + 2*j (the odd phase has PH + 1 rows).
    uint32_t soff0[NSL], soff1[NSL];
    uint32_t wstat[NSL];                  // column term | row << 8 | cc << 13 | valid << 15
#pragma unroll
    for (int e = 0; e < NSL; ++e) {
        const int q = e * 256 + tid;
        const int p = q >> 2;
        const int j = p / ROW;
        const int ii = p - j * ROW;
        const int cp = ii >= PW + 2 ? 1 : 0;
        const int m = ii - cp * (PW + 2);
        const int cc = (q & 3) ^ (((ii >> 2) & 1) << 1);
        const int valid = j < PH + 2 ? 1 : 0;
        wstat[e] = (uint32_t)((2 * m + cp) | ((j & 31) << 8) | (cc << 13) | (valid << 15));
    }
    auto tile_offsets = [&](int g, int yy, int xx) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            soff0[e] = soff1[e] = 0x80000000u;             // out of range -> the DMA writes zeros
            const uint32_t ws = wstat[e];
            const int j = (ws >> 8) & 31, cc = (ws >> 13) & 3;
            const int ix = 2 * xx - 2 + (int)(ws & 255);
            const int iy = 2 * yy - 2 + 2 * j;
            if ((ws >> 15) && g < a.N && (unsigned)ix < (unsigned)a.Wi) {
                const uint32_t o = (uint32_t)((((g * a.Hi + iy) * a.Wi + ix) * a.Ci + cc * 8) * 2);
                if ((unsigned)iy < (unsigned)a.Hi) soff0[e] = o;
                if (j < PH + 1 && (unsigned)(iy + 1) < (unsigned)a.Hi) soff1[e] = o + (uint32_t)(a.Wi * a.Ci * 2);
            }
        }
    };
    tile_offsets(grp, y0, x0);
    const uint32_t lds_wave = lds0 + wave * 1024;
    // slice E of sub-chunk `sub`, phase RP, into window buffer RP
    auto load_slice = [&](auto RP_, int sub, auto E_) __attribute__((always_inline)) {
        constexpr int rp = decltype(RP_)::value, e = decltype(E_)::value;
        if constexpr (e < NSL) wdma(srd_in, rp ? soff1[e] : soff0[e], (uint32_t)sub * 64u, lds_wave + rp * WINB + e * 4096);
    };

    // ---- weight DMA: tap slot = [128 co][32 ch] = 8 KB, 64 rows per block instruction; chunk swizzle 2*bit2(row)
    const int trow = tid >> 2;
    const int wcc = (tid & 3) ^ (((trow >> 2) & 1) << 1);
    const uint32_t vw = (uint32_t)(((co0 + trow) * a.Kpad + wcc * 8) * 2);
    const uint32_t rs64 = (uint32_t)(a.Kpad * 128);      // 64 rows
    const int Ci2 = a.Ci * 2;
    // piece PC (tap slot * 2 + 64-row half) of taps [T0, T0 + 2) (those < 25) of sub-chunk `sub` into ring stage STG
    auto load_w_piece = [&](auto STG_, auto T0_, auto PC_, int sub) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG_)::value, t0 = decltype(T0_)::value, pc = decltype(PC_)::value;
        constexpr int s = pc >> 1, i = pc & 1;
        if constexpr (t0 + s < 25) {
            constexpr int tap = w5_ky(t0 + s) * 5 + w5_kx(t0 + s);
            const uint32_t so = (uint32_t)(tap * Ci2 + sub * 64);
            wdma(srd_w, vw, so + i * rs64, lds_wave + WBUF0 + stg * W_BYTES + s * 8192 + i * 4096);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
#ifdef FMRI_STAMP
    unsigned long long st_sync = 0, st_pend = 0, st_first = 0, st_epi = 0, st_steps = 0, k0, k1, r0, r1;
    FMRI_STAMP_AT(k0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
#endif

    // ---- A fragment addresses: abase[kx] for tile row wm*8 (row tile tm adds tm * ROW * 64, a window row shift sy adds
    // sy * ROW * 64: immediates)
    uint32_t abase[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
        const int ii = (kx & 1) * (PW + 2) + frow + (kx >> 1);
        const int p = (wm * 8) * ROW + ii;
        abase[kx] = (uint32_t)((p << 6) + ((fq ^ (((ii >> 2) & 1) << 1)) << 4));
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
    auto pending_mfma = [&](auto M_) __attribute__((always_inline)) {
        constexpr int m = decltype(M_)::value, tn = m / TM, tm = m % TM;
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pbf[tn], paf[tm], acc[tn][tm], 0, 0, 0);
    };
    auto drain_pending = [&]() __attribute__((always_inline)) {
        static_for_w<0, TM * TN>([&](auto M_) __attribute__((always_inline)) { pending_mfma(M_); });
#pragma unroll
        for (int i = 0; i < TM; ++i) paf[i] = (h8)(half_t)0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i) pbf[i] = (h8)(half_t)0.f;
    };

    // one K-step: taps T0, T0 + 1 (those < 25), weights in ring stage STG; `piece(K)` issues DMA instruction K of the step
    auto step = [&](auto T0_, auto STG_, auto&& piece) __attribute__((always_inline)) {
        constexpr int t0 = decltype(T0_)::value, stg = decltype(STG_)::value;
        constexpr int NS = t0 + 1 < 25 ? 2 : 1;
        h8 af0[TM], bf0[TN];
        {
            constexpr int ky = w5_ky(t0), kx = w5_kx(t0);
            const char* Ps = smem + (ky & 1) * WINB + (ky >> 1) * (ROW * 64);
            const char* Ws = smem + stg * W_BYTES;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af0[tm] = *(const h8*)(Ps + abase[kx] + tm * (ROW * 64));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf0[tn] = *(const h8*)(Ws + (boff + tn * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef FMRI_STAMP
        unsigned long long tb, tc, td;
        FMRI_STAMP_AT(tb);
#endif
        // pending MFMAs with the six DMA pieces between them
        static_for_w<0, TM * TN>([&](auto M_) __attribute__((always_inline)) {
            constexpr int m = decltype(M_)::value;
            pending_mfma(M_);
            constexpr int k = m == 4 ? 0 : m == 9 ? 1 : m == 14 ? 2 : m == 19 ? 3 : m == 24 ? 4 : m == 29 ? 5 : -1;
            if constexpr (k >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                piece(std::integral_constant<int, k>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#ifdef FMRI_STAMP
        FMRI_STAMP_AT(tc);
#endif
        // first-slot MFMAs; the second slot's 12 fragment reads (into the pending registers) between the first 12
        if constexpr (NS == 2) {
            constexpr int ky = w5_ky(t0 + 1), kx = w5_kx(t0 + 1);
            const char* Ps = smem + (ky & 1) * WINB + (ky >> 1) * (ROW * 64);
            const char* Ws = smem + stg * W_BYTES + 8192;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) paf[tm] = *(const h8*)(Ps + abase[kx] + tm * (ROW * 64));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) pbf[tn] = *(const h8*)(Ws + (boff + tn * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) paf[i] = (h8)(half_t)0.f;
#pragma unroll
            for (int i = 0; i < TN; ++i) pbf[i] = (h8)(half_t)0.f;
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
#ifdef FMRI_STAMP
        FMRI_STAMP_AT(td);
        st_pend += tc - tb; st_first += td - tc; st_steps += 1;
#endif
    };

    // ---- one 32-channel sub-chunk: 13 steps.  P = parity of the sub-chunk (ring stage of step t = (P + t) & 1).
    // Window traffic: steps 0-4 bring this sub-chunk's odd-row window (buffer 1, first read in step 7), two slices per step;
    // steps 8-12 the next sub-chunk's even-row window (buffer 0, last read in step 7).  Waits: a step leaves the two window
    // slices issued behind the previous step's weight pieces in flight.
    const int nsub = a.nsub;
    auto run_sub = [&](auto P_, int sub, bool more, int nsubi, bool switch_tile, int ng, int ny, int nx, bool landed)
                       __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        static_for_w<0, 13>([&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value;
            constexpr int stg = (P + t) & 1;
            constexpr int prev_n = (t >= 1 && t <= 5) ? 2 : ((t >= 9) ? 2 : 0);
            constexpr bool prev_cond = t >= 9;             // ... only when another (tile, sub-chunk) follows
#ifdef FMRI_STAMP
            unsigned long long ta, tb0;
            FMRI_STAMP_AT(ta);
#endif
            if constexpr (t == 0) { if (!landed) wait_vmw<0>(); }
            else if constexpr (prev_n == 0) wait_vmw<0>();
            else if constexpr (prev_cond) { if (more) wait_vmw<prev_n>(); else wait_vmw<0>(); }
            else wait_vmw<prev_n>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef FMRI_STAMP
            FMRI_STAMP_AT(tb0);
            st_sync += tb0 - ta;
#endif
            // DMA pieces of this step, in issue order: 4 weight pieces of the next step, then 2 window slices
            auto piece = [&](auto K_) __attribute__((always_inline)) {
                constexpr int k = decltype(K_)::value;
                if constexpr (k < 4) {
                    if constexpr (t < 12) {
                        load_w_piece(std::integral_constant<int, stg ^ 1>{}, std::integral_constant<int, 2 * t + 2>{}, K_, sub);
                    } else {
                        if (more) load_w_piece(std::integral_constant<int, stg ^ 1>{}, std::integral_constant<int, 0>{}, K_, nsubi);
                    }
                } else {
                    constexpr int j = k - 4;                      // 0, 1
                    if constexpr (t <= 4) {
                        load_slice(std::integral_constant<int, 1>{}, sub, std::integral_constant<int, 2 * t + j>{});
                    } else if constexpr (t == 6) {
                        if constexpr (j == 0) { if (switch_tile) tile_offsets(ng, ny, nx); }
                    } else if constexpr (t >= 8) {
                        if (more) load_slice(std::integral_constant<int, 0>{}, nsubi, std::integral_constant<int, 2 * (t - 8) + j>{});
                    }
                }
            };
            step(std::integral_constant<int, 2 * t>{}, std::integral_constant<int, stg>{}, piece);
        });
    };

    // ---- epilogue: D[i = co][j = tile pixel] -> NHWC fp16 (no bias / activation: BatchNorm or a data gradient follows)
    float vsum = 0.f, vsq = 0.f;         // over all tiles of the block
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? grp / a.st.group_n : 0;     // statistics group of the block's tiles
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int x = x0 + frow;
        const bool xok = grp < a.N && x < a.Wo;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int co = co0 + wn * (BN / WN) + tn * 16 + fq * 4;
            if (co >= a.CoStore) continue;
            f4 s0 = (f4){0.f, 0.f, 0.f, 0.f}, s1 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int y = y0 + wm * 8 + tm;
                if (!xok || y >= a.Ho) continue;
                const f4 v = acc[tn][tm];
                h4 hv;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)(co + rg < a.Co ? v[rg] : 0.f);
                if constexpr (STATS == 1) {
                    // statistics of the STORED (fp16-rounded) values: what the consumers and the BN backward see
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float f = (float)hv[rg];
                        s0[rg] += f;
                        s1[rg] += f * f;
                    }
                }
                *(h4*)(a.out + (((int64_t)grp * a.Ho + y) * a.Wo + x) * a.CoStore + co) = hv;
            }
            if constexpr (STATS != 0) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float ra = row16_sum(s0[rg]);
                    const float rb = row16_sum(s1[rg]);
                    if (frow == tn * 4 + rg) { vsum += ra; vsq += rb; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // prologue: even-row window of the first tile's sub-chunk 0 and the first weight tiles
    static_for_w<0, NSL>([&](auto E_) __attribute__((always_inline)) { load_slice(std::integral_constant<int, 0>{}, 0, E_); });
    static_for_w<0, 4>([&](auto K_) __attribute__((always_inline)) {
        load_w_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, K_, 0);
    });
    int tile = tile0, sub = 0;
    bool landed = false;
    int ng = 0, ny = 0, nx = 0;
    if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
    auto one = [&](auto P_) __attribute__((always_inline)) {
        const bool next_tile = tile + 1 < tile1;
        const bool last_sub = sub + 1 >= nsub;
        run_sub(P_, sub, !last_sub || next_tile, last_sub ? 0 : sub + 1, last_sub && next_tile, ng, ny, nx, landed);
        ++sub;
        landed = false;
        if (last_sub) {
            drain_pending();                                     // the last step's second tap slot
            if (next_tile) { wait_vmw<0>(); landed = true; }     // the next tile's first window and weights
#ifdef FMRI_STAMP
            unsigned long long te0, te1;
            FMRI_STAMP_AT(te0);
#endif
            epilogue();
#ifdef FMRI_STAMP
            FMRI_STAMP_AT(te1);
            st_epi += te1 - te0;
#endif
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
            grp = ng; y0 = ny; x0 = nx;
            sub = 0;
            ++tile;
            if (tile + 1 < tile1) tile_geom(tile + 1, ng, ny, nx);
        }
    };
    while (tile < tile1) {
        one(std::integral_constant<int, 0>{});
        if (tile < tile1) one(std::integral_constant<int, 1>{});
    }
#ifdef FMRI_STAMP
    FMRI_STAMP_AT(k1);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    if (lane == 0) {
        atomicAdd(&c5w_stamp_acc[0], st_sync); atomicAdd(&c5w_stamp_acc[1], st_pend); atomicAdd(&c5w_stamp_acc[2], st_first);
        atomicAdd(&c5w_stamp_acc[3], st_epi); atomicAdd(&c5w_stamp_acc[4], st_steps); atomicAdd(&c5w_stamp_acc[5], 1ull);
        atomicAdd(&c5w_stamp_acc[6], k1 - k0); atomicAdd(&c5w_stamp_acc[7], r1 - r0);
    }
#endif
    if constexpr (STATS != 0) {
        // one row per block; the blocks of one statistics group are contiguous (tpg[0] = blocks per group).  The lane that
        // owns channel (frow >> 2)*16 + fq*4 + (frow & 3) of the wave's 64 holds its sums; the two pixel halves (wm) meet
        // in LDS (stat_store)
        const int prow = bx - sgrp * a.st.tpg[0];
        stat_store<TN, WM, WN>(vsum, vsq, lane, wm, wn, co0, (float*)smem,
                               a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C, a.st.C);
    }
}

template <int STATS>
static int launch_c5w(const C5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_c5w_kernel<STATS>;
    constexpr int lds = 2 * 10 * 4096 + 2 * 16384;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.ntiles + a.tpb - 1) / a.tpb, copad / 128, 1), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// k5 s2 p2, Ci % 32 == 0, 128-channel tiles, 16 x 16-pixel tiles of one image, no bias / activation, no BnBwdEpi.
// C5Args with tiles_y = ceil(Ho / 16), tiles_x = ceil(Wo / 16), ntiles = N * tiles_y * tiles_x.
int igemm_c5w_launch(const C5Args& a, int copad, hipStream_t st) {
    if (a.nsub < 1 || (copad & 127) || a.ntiles < 1 || a.tpb < 1 || a.bb.x) return E_UNSUPPORTED;
    return a.st.part ? launch_c5w<1>(a, copad, st) : launch_c5w<0>(a, copad, st);
}

#ifdef FMRI_STAMP
extern "C" int fmri_debug_c5w_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(c5w_stamp_acc), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(c5w_stamp_acc), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace fmri
