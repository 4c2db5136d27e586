// Stride-2 transposed convolution (k5 p2), Ci % 128 == 0, 128-channel tiles, class grids wider than 8 on MFMA (gfx950):
// WIDE form of igemm_tc5.hip -- one 8-wave block per CU owns 16 x 16 class-grid positions x 128 channels, four loader
// waves feed four compute waves (wave tile 128 positions x 64 channels).
//
// Replaces (reference models/vae_gan.py): ConvTranspose2d(k5, s2, p2, output_padding 1) forward of decoder.conv.1
// (:46-53, :112-116) and the data gradient of discriminator.conv.2 (:149-153), i.e. the 16-wide-tile launches of
// igemm_tc5.hip with 128 output channels per block.
//
// Same recipe as igemm_c5w.hip (round 3): with two 4-wave blocks per CU every wave issued its own operand
// DMA (~60 cycles of its instruction stream per 1 KB piece, 5 pieces per 32 MFMAs) and fetched 0.5 LDS fragments per
// MFMA.  Here
//   * waves 4-7 only move bytes (window slices, weight tiles: buffer_load ... lds, counted vmcnt), waves 0-3 only read
//     fragments and issue MFMAs; one s_barrier per K-step joins them;
//   * the wave tile is 8 x 4 MFMA tiles: 0.375 fragment reads per MFMA, 64 MFMAs per wave and K-step (one tap x 64 channels);
//   * the second 32-channel half of a K-step stays pending in registers across the barrier: its 32 MFMAs cover the
//     first half's fragment reads of the next step;
//   * the window swizzle depends on the window COLUMN only, so that a tile row (and a tap's row shift) is an immediate
//     of the ds_read: 6 address registers instead of igemm_tc5's 36;
//   * outputs leave through buffer stores (one 32-bit offset per lane and class, the tile row as the scalar offset).
// Everything else is igemm_tc5.hip's: the four parity classes (3x3, 3x2, 2x3, 2x2 unit-shift taps) run back to back over
// the same (16 + 2)^2 window of 64-channel chunks (double-buffered), weights straight out of the per-class packed
// matrices, BatchNorm forward statistics of the stored values summed over the four classes (StatEpi).
#include "kernels.h"
#include <type_traits>

namespace fmri {


namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// output stores of the previous class issued behind the DMA of step J of a class with T taps (first chunk pair: 2T steps, 16 items)
constexpr int t5_nst(int j, int T) { return 16 / (2 * T) + (j < 16 % (2 * T) ? 1 : 0); }
constexpr int t5_st0(int j, int T) { return j * (16 / (2 * T)) + (j < 16 % (2 * T) ? j : 16 % (2 * T)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for_t(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_t<I + 1, N>(f);
    }
}

__device__ __forceinline__ void tdma(v4i srd, uint32_t voff, uint32_t soff, uint32_t lds) {
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
__device__ __forceinline__ void wait_vmt() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// PW: 16 = tiles of 16 x 16 class-grid positions of one image; 8 = 8 x 8 positions (class grids and inputs of at most 8 x 8)
// of FOUR images, an MFMA row tile = one tile row of two images.  STATS: 0 none, 1 BatchNorm forward statistics (StatEpi)
// SOLO: one parity class per block (grid.z = 4), for launches of few tiles
template <int PW, int STATS, bool SOLO = false>
__global__ __launch_bounds__(512, 1) void igemm_tc5w_kernel(const Tc5Args a) {
    constexpr int BN = 128, WN = 2, TM = 8, TN = 4;
    constexpr int PH = PW, IPB = PW == 16 ? 1 : 4;
    constexpr int IW = PW + 2, IH = PH + 2;              // window of one image, pixels (128 B each: one 64-channel chunk)
    constexpr int ROWB = IW * 128;
    constexpr int NSL = (IPB * IH * IW * 128 + 4095) / 4096;      // 4 KB DMA slices per window chunk: 11 (41 472 B) / 13 (51 200 B)
    constexpr int WINB = NSL * 4096;
    constexpr int W_BYTES = BN * 128;                    // one tap x 64 channels of [128 co]
    constexpr int WBUF0 = 2 * WINB;
    // output staging: 128 positions x 128 channels fp16 (32 KB).  PW = 8: inside window buffer 1 -- the buffer of a class's
    // last (odd) chunk is free from its last fragment read to the first DMA of the next class's second chunk, and the hand-over
    // sits in between
    constexpr int STG0 = PW == 16 ? WBUF0 + 2 * W_BYTES : WINB;
    static_assert(PW == 16 || PW == 8, "tile");
    static_assert((2 + 7) * ROWB < 65536 && 32768 <= WINB, "ds_read immediates / staging");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= 4;                        // waves 4-7 move bytes, waves 0-3 multiply
    int bx, by;
    xcd_tile(bx, by);
    if (bx >= a.ntiles) return;
    const int co0 = by * BN;
    // launches of few tiles (the encoder's N = 256 layers: 64 tiles for 256 CUs): one parity class per block, grid.z = 4
    constexpr bool solo = SOLO;
    const int cls0 = solo ? (int)blockIdx.z : 0;

    // ---- tile -> (image, tile row, tile column) of the class grid
    const int tpi = a.tiles_y * a.tiles_x;
    const int grp = (int)fd_div((uint32_t)bx, a.fdTPI);
    const int trem = bx - grp * tpi;
    const int tyi = (int)fd_div((uint32_t)trem, a.fdTX);
    const int y0 = tyi * PH, x0 = (trem - tyi * a.tiles_x) * PW;
    const int nch = a.nchunks;

    // BatchNorm statistics: summed by the LOADER waves over the fp16 values they keep for the output stores (they idle
    // between DMA issues; in the compute waves the same sums cost ~1 500 instructions per class).  Loader thread tid holds
    // channels co0 + 8*slot .. +7 (slot = (tid & 15) ^ (tid >> 4)) of position tid >> 4 of every item
    float lsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lsq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? (grp * IPB) / a.st.group_n : 0;     // statistics group of the tile

    if (loader) {
        // =====================================================================================================
        // loader waves: per K-step, wait for the operands of this step, meet the compute waves at the barrier (they are
        // done with the ring stage and the window buffer about to be overwritten), issue the DMA of the step
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

        // ---- window DMA: 16-B unit q = e*256 + tid of a window buffer holds channels 8*cc .. 8*cc+7 (of the 64-channel
        // chunk) of window pixel q >> 3 = image ip, row j, column i; cc = (q & 7) ^ (i & 6).  Window origin = tile origin - 1.
        uint32_t soff[NSL];
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            soff[e] = 0x80000000u;                         // out of range -> the DMA writes zeros
            const int q = e * 256 + tid;
            const int pixel = q >> 3;
            const int ip = pixel / (IH * IW);
            const int pr = pixel - ip * (IH * IW);
            const int j = pr / IW;
            const int i = pr - j * IW;
            const int n = grp * IPB + ip;
            const int iy = y0 - 1 + j, ix = x0 - 1 + i;
            if (ip < IPB && n < a.N && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                soff[e] = (uint32_t)((((n * a.Hi + iy) * a.Wi + ix) * a.Ci + (((q & 7) ^ (i & 6)) << 3)) * 2);
        }
        const uint32_t lds_wave = lds0 + lw * 1024;
        // slices [LO, HI) of channel chunk `chunk` into window buffer BUF
        auto load_slices = [&](auto BUF_, int chunk, auto LO_, auto HI_) __attribute__((always_inline)) {
            constexpr int buf = decltype(BUF_)::value, lo = decltype(LO_)::value, hi = decltype(HI_)::value;
            const uint32_t so = (uint32_t)chunk * 128u;
            static_for_t<lo, (hi < NSL ? hi : NSL)>([&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                tdma(srd_in, soff[e], so, lds_wave + buf * WINB + e * 4096);
            });
        };
        // ---- weight tile DMA (rows = co, 64 k-values per step), XOR swizzled like igemm.hip.  Per class: per-lane offset
        // vw (row, 16-B column), scalar offset of the class matrix sw and of 32 rows rs.
        const int trow = tid >> 3;
        const int clog = (tid & 7) ^ ((trow >> 1) & 7);
        auto class_w = [&](int cls, uint32_t& vw, uint32_t& sw, uint32_t& rs) __attribute__((always_inline)) {
            const int kp = a.cls[cls].Kpad;
            vw = (uint32_t)(((co0 + trow) * kp + clog * 8) * 2);
            sw = (uint32_t)(a.cls[cls].w_off * 2);
            rs = (uint32_t)(kp * 64);
        };
        auto load_w = [&](auto STG_, uint32_t vw, uint32_t so, uint32_t rs) __attribute__((always_inline)) {
            constexpr int stg = decltype(STG_)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tdma(srd_w, vw, so + i * rs, lds_wave + WBUF0 + stg * W_BYTES + i * 4096);
            }
        };
        const int Ci2 = a.Ci * 2;
        uint32_t vw, sw, rs;
        class_w(0, vw, sw, rs);

        // ---- outputs: the compute waves hand a finished class over through LDS in two rounds (tile rows 0-7, 8-15; position
        // p = row * 16 + x of the round at p * 256 B, 16-B channel slot s at slot s ^ x); thread tid keeps item k = row k,
        // x = tid >> 4, slot tid & 15 of both rounds in registers and stores them one or two per K-step of the NEXT class
        // (the last class: at the end): 256 contiguous bytes per position, and no store ever waits in a compute wave
        const __amdgpu_buffer_rsrc_t srd_out = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.out, 0, (int)((uint32_t)a.N * (uint32_t)a.Ho * (uint32_t)a.Wo * (uint32_t)a.CoStore * 2u), 0x00020000);
        const uint32_t row2_b = (uint32_t)(2 * a.Wo * a.CoStore * 2);       // two output rows = one class-grid row
        v4i oreg[2][8];
        uint32_t ovo[2];
        int onrow[2];
        auto take = [&](int cls) __attribute__((always_inline)) {
            // (PW = 8: a round = two images, item k = class-grid row k of both, position tid >> 4 = image (bit 3), column)
            const int opx = tid >> 4;
            const int slot = (tid & 15) ^ opx;
            const int ox = PW == 16 ? opx : (opx & 7);
            const int cy = cls >> 1, cx = cls & 1;
            const int Yc = a.cls[cls].Yc, Xc = a.cls[cls].Xc;
            if constexpr (PW == 8) __builtin_amdgcn_s_barrier();   // (every compute wave is done with window buffer 1 = the staging area)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                __builtin_amdgcn_s_barrier();                      // the round is in LDS
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) oreg[r][k] = *(const v4i*)(smem + STG0 + tid * 16 + k * 4096);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // ... and in registers: the staging area is free
                __builtin_amdgcn_sched_barrier(0);
                const int n = PW == 16 ? grp : grp * IPB + r * 2 + (opx >> 3);
                const int yb = PW == 16 ? y0 + r * 8 : y0, x = x0 + ox;
                const bool ok = n < a.N && x < Xc && co0 + slot * 8 < a.CoStore;
                ovo[r] = ok ? (uint32_t)((((n * a.Ho + 2 * yb + cy) * a.Wo + 2 * x + cx) * a.CoStore + co0 + slot * 8) * 2) : 0x80000000u;
                onrow[r] = Yc - yb;                                // tile rows k < onrow exist in this class
            }
        };
        // item I of the kept class (always issued, so that vmcnt counts stay compile-time: rows outside the class grid carry
        // an out-of-range offset and are dropped)
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
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, oreg[r][k]), srd_out, (int)vt, (int)(k * row2_b), 0);
            FMRI_STORE_FENCE();        // SGPR-offset store: the compiler pads no wait states (common.h)
        };
        auto feed_class = [&](auto CLS_) __attribute__((always_inline)) {
            constexpr int cls = decltype(CLS_)::value;
            constexpr int TH = (cls >> 1) ? 2 : 3, TW = (cls & 1) ? 2 : 3, T = TH * TW;
            constexpr int SPT = (NSL + T - 1) / T;            // window slices issued per tap
            constexpr bool LAST = solo || cls == 3;           // no class follows in this block
            uint32_t vwn = 0, swn = 0, rsn = 0;
            if constexpr (!LAST) class_w(cls + 1, vwn, swn, rsn);
            for (int chunk = 0; chunk < nch; chunk += 2) {
                static_for_t<0, 2>([&](auto PB_) __attribute__((always_inline)) {
                    constexpr int pb = decltype(PB_)::value;
                    const int ch = chunk + pb;
                    const bool last_chunk = ch + 1 >= nch;
                    const bool more_win = !(LAST && last_chunk);       // another (class, chunk) window follows
                    static_for_t<0, T>([&](auto TAP_) __attribute__((always_inline)) {
                        constexpr int t = decltype(TAP_)::value;
                        constexpr int stg = (pb * T + t) & 1;
                        // slices issued behind the weight tile of the PREVIOUS step (window of the next chunk)
                        constexpr int prev_lo = t == 0 ? 0 : (t - 1) * SPT;
                        constexpr int prev_n = t == 0 ? 0
                                                      : ((prev_lo >= NSL) ? 0 : ((prev_lo + SPT > NSL ? NSL : prev_lo + SPT) - prev_lo));
                        // weights of this step landed; at the first tap of a chunk the whole window must have landed too.
                        // In flight may stay: the slices and the output stores issued behind the previous step's weights
                        constexpr int j = pb * T + t;
                        constexpr int ns_prev = (cls > 0 && t > 0) ? t5_nst(j - 1, T) : 0;
                        if constexpr (t == 0) wait_vmt<0>();
                        else {
                            const bool win_prev = !LAST || more_win;
                            if (chunk == 0 && !solo) { if (win_prev) wait_vmt<prev_n + ns_prev>(); else wait_vmt<ns_prev>(); }
                            else { if (win_prev) wait_vmt<prev_n>(); else wait_vmt<0>(); }
                        }
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        // next step's weight tile
                        if constexpr (t + 1 < T) {
                            load_w(std::integral_constant<int, stg ^ 1>{}, vw, sw + (uint32_t)((t + 1) * Ci2 + ch * 128), rs);
                        } else {
                            if (!last_chunk) load_w(std::integral_constant<int, stg ^ 1>{}, vw, sw + (uint32_t)((ch + 1) * 128), rs);
                            else if constexpr (!LAST) load_w(std::integral_constant<int, stg ^ 1>{}, vwn, swn, rsn);
                        }
                        // window of the next (class, chunk), spread over the taps
                        if (more_win)
                            load_slices(std::integral_constant<int, pb ^ 1>{}, last_chunk ? 0 : ch + 1,
                                        std::integral_constant<int, t * SPT>{}, std::integral_constant<int, t * SPT + SPT>{});
                        // outputs of the previous class
                        if constexpr (cls > 0) {
                            if (chunk == 0 && !solo)
                                static_for_t<t5_st0(j, T), t5_st0(j, T) + t5_nst(j, T)>([&](auto I_) __attribute__((always_inline)) { put(I_); });
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
            vw = vwn; sw = swn; rs = rsn;
            take(cls);
        };
        // prologue: window of (class 0, chunk 0) and the first weight tile
        if constexpr (solo) class_w(cls0, vw, sw, rs);
        load_slices(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NSL>{});
        load_w(std::integral_constant<int, 0>{}, vw, sw, rs);
        if constexpr (solo) {
            if (cls0 == 0) feed_class(std::integral_constant<int, 0>{});
            else if (cls0 == 1) feed_class(std::integral_constant<int, 1>{});
            else if (cls0 == 2) feed_class(std::integral_constant<int, 2>{});
            else feed_class(std::integral_constant<int, 3>{});
        } else {
            feed_class(std::integral_constant<int, 0>{});
            feed_class(std::integral_constant<int, 1>{});
            feed_class(std::integral_constant<int, 2>{});
            feed_class(std::integral_constant<int, 3>{});
        }
        static_for_t<0, 16>([&](auto I_) __attribute__((always_inline)) { put(I_); });
    } else {
        // =====================================================================================================
        // compute waves: LDS fragment reads and MFMAs only
        // =====================================================================================================
        // ---- A fragment addresses of window column shift sx, first 32-channel half, window buffer 0, row tile 0 of the wave
        // (PW = 16: tile row wm*8, lane = column; PW = 8: row 0 of images wm*2 + (frow >> 3), lane & 7 = column):
        // row tile tm and row shift sy add (tm + sy) * ROWB (immediates), the second half is ^ 64, buffer 1 is + WINB
        uint32_t abase[2][3];
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int col = (PW == 16 ? frow : (frow & 7)) + sx;
            const int pix0 = PW == 16 ? (wm * 8) * IW : (wm * 2 + (frow >> 3)) * IH * IW;
            abase[0][sx] = (uint32_t)((pix0 + col) * 128 + ((fq ^ (col & 6)) << 4));
            abase[1][sx] = abase[0][sx] + WINB;          // (the immediates of buffer 1 would pass 16 bits)
        }
        // ---- B fragment address (row = wn*64 + tn*16 + frow; the swizzle term does not depend on tn or wn)
        const uint32_t boff = (uint32_t)(WBUF0 + (wn * (BN / WN) + frow) * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));

        f4 acc[TN][TM];
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        };
        zero_acc();

        // ---- the pending second half of the previous K-step (all zeros: nothing pending)
        h8 paf[TM], pbf[TN];
        auto clear_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) paf[i] = (h8)(half_t)0.f;
#pragma unroll
            for (int i = 0; i < TN; ++i) pbf[i] = (h8)(half_t)0.f;
        };
        clear_pending();
        auto pending_mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pbf[tn], paf[tm], acc[tn][tm], 0, 0, 0);
        };
        auto interleave = [&]() __attribute__((always_inline)) {
            // the 12 fragment reads one by one between the first MFMAs
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - 2 * (TM + TN), 0);
        };

        // one K-step: window shift (SY, SX), window buffer PB, ring stage STG
        auto step = [&](auto SY_, auto SX_, auto PB_, auto STG_) __attribute__((always_inline)) {
            constexpr int sy = decltype(SY_)::value, sx = decltype(SX_)::value, pb = decltype(PB_)::value, stg = decltype(STG_)::value;
            const char* Ps = smem + sy * ROWB;
            const char* Ws = smem + stg * W_BYTES;
            h8 af0[TM], bf0[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af0[tm] = *(const h8*)(Ps + abase[pb][sx] + tm * ROWB);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf0[tn] = *(const h8*)(Ws + (boff + tn * 2048));
            pending_mfmas();
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            {
                // the second half: one XOR each (volatile: the compiler would otherwise keep both address sets live)
                uint32_t ax, bxo;
                asm volatile("v_xor_b32 %0, 64, %1" : "=v"(ax) : "v"(abase[pb][sx]));
                asm volatile("v_xor_b32 %0, 64, %1" : "=v"(bxo) : "v"(boff));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) paf[tm] = *(const h8*)(Ps + ax + tm * ROWB);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) pbf[tn] = *(const h8*)(Ws + (bxo + tn * 2048));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[tn], af0[tm], acc[tn][tm], 0, 0, 0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        };

        // ---- a finished class: D[i = co][j = class-grid position] as fp16 into the staging area, tile rows 0-7 (the wm = 0 waves)
        // then 8-15 (wm = 1); the loader waves take each round into registers and store it (see `take`).  Position p = tm * 16
        // + frow at p * 256 B, 16-B channel slot s = wn * 8 + tn * 2 + (fq >> 1) at slot s ^ frow (spreads the 16 positions
        // of a write over the banks).  Channels >= Co are written as zeros.
        const bool full_co = ((a.Co | a.CoStore) & 127) == 0;
        const uint32_t sbase = (uint32_t)(STG0 + frow * 256 + (((wn * 8 + (fq >> 1)) ^ frow) << 4) + (fq & 1) * 8);
        const bool aff_on = STATS == 0 && a.aff.scale != nullptr, aff_relu = a.aff.relu != 0;
        auto hand_body = [&](int cls, auto FULL_) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(FULL_)::value;
            (void)cls;                                       // positions outside the class grid are handed over too (the loader waves drop them)
            const int cw = co0 + wn * (BN / WN) + fq * 4;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int co = cw + tn * 16;
                // AffEpi (eval-mode BatchNorm of the consumer folded in): scale / shift of the lane's four channels
                f4 asc = (f4){1.f, 1.f, 1.f, 1.f}, ash = (f4){0.f, 0.f, 0.f, 0.f};
                const bool aff = aff_on;
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
                            if (aff_relu) v[rg] = fmaxf(v[rg], 0.f);
                        }
                    }
                    h4 hv;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)((FULL || co + rg < a.Co) ? v[rg] : 0.f);
                    *(h4*)(smem + ((sbase ^ (uint32_t)(tn * 32)) + tm * 4096)) = hv;
                }
            }
        };
        auto hand_over = [&](int cls) __attribute__((always_inline)) {
            if constexpr (PW == 8) {
                // the staging area lies in window buffer 1: every compute wave has to be done with the class's last fragments
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (wm == r) {
                    if (full_co) hand_body(cls, std::true_type{});
                    else hand_body(cls, std::false_type{});
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // the round is in LDS
                __builtin_amdgcn_s_barrier();                      // ... and in the loader waves' registers
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        auto run_class = [&](auto CLS_) __attribute__((always_inline)) {
            constexpr int cls = decltype(CLS_)::value;
            constexpr int TH = (cls >> 1) ? 2 : 3, TW = (cls & 1) ? 2 : 3, T = TH * TW;
            for (int chunk = 0; chunk < nch; chunk += 2) {
                static_for_t<0, 2>([&](auto PB_) __attribute__((always_inline)) {
                    constexpr int pb = decltype(PB_)::value;
                    static_for_t<0, T>([&](auto TAP_) __attribute__((always_inline)) {
                        constexpr int t = decltype(TAP_)::value;
                        constexpr int ty = t / TW, tx = t % TW;
                        constexpr int stg = (pb * T + t) & 1;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        step(std::integral_constant<int, 2 - ty>{}, std::integral_constant<int, 2 - tx>{}, PB_,
                             std::integral_constant<int, stg>{});
                    });
                });
            }
            pending_mfmas();                                     // the last step's second half
            hand_over(cls);
            clear_pending();
            zero_acc();
        };
        if constexpr (solo) {
            if (cls0 == 0) run_class(std::integral_constant<int, 0>{});
            else if (cls0 == 1) run_class(std::integral_constant<int, 1>{});
            else if (cls0 == 2) run_class(std::integral_constant<int, 2>{});
            else run_class(std::integral_constant<int, 3>{});
        } else {
            run_class(std::integral_constant<int, 0>{});
            run_class(std::integral_constant<int, 1>{});
            run_class(std::integral_constant<int, 2>{});
            run_class(std::integral_constant<int, 3>{});
        }
    }

    // ---- BatchNorm statistics of the block (all four classes): its own row of the partial buffer.  The 16 loader threads
    // that hold the same 8 channels (one per position column of an item) meet in LDS
    if constexpr (STATS != 0) {
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
            // (solo: four rows per tile, one per class block; tpg[0] counts rows)
            const int prow = solo ? (bx * 4 + cls0) - sgrp * a.st.tpg[0] : bx - sgrp * a.st.tpg[0];
            float* row = a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C;
            if (co0 + c < a.st.C) {
                row[co0 + c] = s0;
                row[a.st.C + co0 + c] = s1;
            }
        }
    }
}

template <int PW, int STATS, bool SOLO>
static int launch_tc5w_(const Tc5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_tc5w_kernel<PW, STATS, SOLO>;
    constexpr int lds = PW == 16 ? 2 * 11 * 4096 + 2 * 128 * 128 + 32768 : 2 * 13 * 4096 + 2 * 128 * 128;
    if (route_probe("fmri::igemm_tc5w_kernel<%d,%d,%s>", PW, STATS, SOLO ? "true" : "false")) return OK;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ntiles, copad / 128, SOLO ? 4 : 1), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

template <int PW, int STATS>
static int launch_tc5w(const Tc5Args& a, int copad, hipStream_t st) {
    if (a.solo) return launch_tc5w_<PW, STATS, true>(a, copad, st);
    return launch_tc5w_<PW, STATS, false>(a, copad, st);
}

// Tc5Args with 16 x 16-position tiles of one image (IPB = 1: tiles_y = ceil(Yc0 / 16), tiles_x = ceil(Xc0 / 16), ntiles = N *
// tiles_y * tiles_x) or, for class grids and inputs of at most 8 x 8, one 8 x 8 tile of four images (IPB = 4: ntiles =
// ceil(N / 4)); nchunks even; no bias / activation, no BnBwdEpi
int igemm_tc5w_launch(const Tc5Args& a, int copad, hipStream_t st) {
    if (a.bias != nullptr || a.act != ACT_NONE || a.bb.x || (copad & 127) || a.ntiles < 1 || (a.nchunks & 1) ||
        (a.aff.scale && a.st.part))
        return E_UNSUPPORTED;
    if ((int64_t)a.N * a.Ho * a.Wo * a.CoStore * 2 >= 0x7fffffffLL) return E_UNSUPPORTED;      // 32-bit store offsets
    if (a.IPB == 4) {
        if (a.Hi > 8 || a.Wi > 8 || a.tiles_x != 1 || a.tiles_y != 1) return E_BADARG;
        return a.st.part ? launch_tc5w<8, 1>(a, copad, st) : launch_tc5w<8, 0>(a, copad, st);
    }
    return a.st.part ? launch_tc5w<16, 1>(a, copad, st) : launch_tc5w<16, 0>(a, copad, st);
}


}  // namespace fmri
