// igemm_tc5.hip's contraction (all four parity classes of a k5 s2 p2 transposed convolution per block, Ci % 128 == 0)
// as ONE 8-wave block per CU that owns TWO 128-position tiles and shares every weight tile between them, with the two
// wave groups running half a K-step apart ("ping-pong").
//
// igemm_tc5.hip puts two independent 4-wave blocks on a CU.  Each streams its own copy of every 16 KB weight tile
// (5 LDS-DMA pieces per wave and K-step, ~100 cycles of issue each) and the two waves of a SIMD interleave only by
// chance: measured 0.52 of the matrix pipes' cycles busy.  Here
//   * waves 0-3 (group 0) and 4-7 (group 1) each own one tile (its window double-buffered as before); a K-step's weight
//     tile is DMA'd ONCE for both, 2 pieces per wave, into a 3-stage ring, two K-steps ahead;
//   * a K-step is two phases separated by workgroup barriers: L = {issue DMA, issue the 16 fragment reads} and
//     M = {32 MFMAs}.  Group 1 starts one barrier late, so on every SIMD one wave is in its L phase while the other is in
//     its M phase: the matrix pipe always has a wave issuing MFMAs while the other wave's DMA / LDS issue overlaps;
//   * counted vmcnt: at the end of an L phase a wave leaves only the pieces it issued in that very phase in flight, so
//     every DMA has a full K-step to land, and a tile issued by group 1 in phase 2s+1 is complete (and behind a barrier)
//     before group 0 reads it in phase 2s+4.
// Everything else (descriptor DMA with hardware zero fill, compile-time tap loops, address tables, statistics epilogues)
// is igemm_tc5.hip's.
#include "kernels.h"
#include <type_traits>

namespace fmri {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for_b(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_b<I + 1, N>(f);
    }
}

__device__ __forceinline__ void bdma16b(v4i srd, uint32_t voff, uint32_t soff, uint32_t lds) {
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
__device__ __forceinline__ void wait_vmb() {
    static_assert(N >= 0 && N <= 15, "pieces issued per L phase");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

// NSL: 4 KB DMA slices per window chunk (6, 7; 4 = dense 8 x 8 windows, see igemm_tc5.hip).  STATS as in igemm_tc5.hip.
template <int NSL, int STATS>
__global__ __launch_bounds__(512, 2) void igemm_tc5b_kernel(const Tc5Args a) {
    constexpr int BN = 128, BM = 128, WM = 2, WN = 2;
    constexpr int TM = 4, TN = 4;
    constexpr int W_BYTES = BN * 128;                // 16 KB weight tile
    constexpr int WSTAGES = 3;
    constexpr bool DENSE = NSL == 4;
    constexpr int WINB = DENSE ? 4 * 4096 + 128 : NSL * 4096;
    constexpr int ZERO_OFF = 4 * 4096;
    constexpr int WBUF0 = 4 * WINB;                  // [group][buffer] windows, then the weight ring
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp2 = wave >> 2;                      // wave group = tile of the block
    const int gw = wave & 3;                         // wave within the group
    const int gtid = tid & 255;
    int bx, by;
    xcd_tile(bx, by);
    const int co0 = by * BN;
    // the block's two tiles; an odd tile count leaves the last block's second group on a copy of its first tile (it takes
    // part in every barrier and weight DMA but stores nothing)
    int tile = 2 * bx + grp2;
    const bool live = tile < a.ntiles;
    if (!live) tile = a.ntiles - 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    const int tpi = a.tiles_y * a.tiles_x;
    const int grp = (int)fd_div((uint32_t)tile, a.fdTPI);
    const int trem = tile - grp * tpi;
    const int tyi = (int)fd_div((uint32_t)trem, a.fdTX);
    const int txi = trem - tyi * a.tiles_x;
    const int PW = 1 << a.pw_log2;
    const int y0 = tyi * a.PH, x0 = txi * PW;
    const int IHW = a.IH * a.IW;

    v4i srd_in, srd_w;
    srd_in.x = (int)(uint32_t)(uintptr_t)a.in;
    srd_in.y = (int)(uint32_t)((uintptr_t)a.in >> 32);
    srd_in.z = (int)a.in_bytes;
    srd_in.w = 0x00020000;
    srd_w.x = (int)(uint32_t)(uintptr_t)a.w;
    srd_w.y = (int)(uint32_t)((uintptr_t)a.w >> 32);
    srd_w.z = (int)a.w_bytes;
    srd_w.w = 0x00020000;

    // ---- window DMA of the group's tile (units of the group's 256 threads, igemm_tc5.hip's layout)
    uint32_t soff[NSL];
    {
        const FastDiv fIHW = a.fdIHW, fIW = a.fdIW;
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            soff[e] = 0x80000000u;
            const int q = e * 256 + gtid;
            const int pixel = q >> 3;
            int ip, j, i;
            if constexpr (DENSE) {
                ip = pixel >> 6; j = ((pixel >> 3) & 7) + 1; i = (pixel & 7) + 1;
            } else {
                ip = (int)fd_div((uint32_t)pixel, fIHW);
                const int rem = pixel - ip * IHW;
                j = (int)fd_div((uint32_t)rem, fIW);
                i = rem - j * a.IW;
            }
            const int n = grp * a.IPB + ip;
            const int iy = y0 - 1 + j, ix = x0 - 1 + i;
            if (ip < a.IPB && n < a.N && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                soff[e] = (uint32_t)((((n * a.Hi + iy) * a.Wi + ix) * a.Ci + (((q & 7) ^ (pixel & 6)) << 3)) * 2);
        }
    }
    const uint32_t lds_win = lds0 + grp2 * 2 * WINB + gw * 1024;
    auto load_slices = [&](auto BUF_, int chunk, auto LO_, auto HI_) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF_)::value, lo = decltype(LO_)::value, hi = decltype(HI_)::value;
        const uint32_t so = (uint32_t)chunk * 128u;
#pragma unroll
        for (int e = lo; e < hi; ++e)
            if (e < NSL) bdma16b(srd_in, soff[e], so, lds_win + buf * WINB + e * 4096);
    };

    // ---- weight tile DMA by all 512 threads: thread loads 16 B of row trow + 64 i (i < 2), swizzled like igemm.hip
    const int trow = tid >> 3;                       // 0 .. 63
    const int clog = (tid & 7) ^ ((trow >> 1) & 7);
    auto class_w = [&](int cls, uint32_t& vw, uint32_t& sw, uint32_t& rs) __attribute__((always_inline)) {
        const int kp = a.cls[cls].Kpad;
        vw = (uint32_t)(((co0 + trow) * kp + clog * 8) * 2);
        sw = (uint32_t)(a.cls[cls].w_off * 2);
        rs = (uint32_t)(kp * 128);                   // 64 rows
    };
    const uint32_t lds_w = lds0 + WBUF0 + wave * 1024;
    auto load_w = [&](int stage, uint32_t vw, uint32_t so, uint32_t rs) __attribute__((always_inline)) {
        const uint32_t dst = lds_w + (uint32_t)stage * W_BYTES;
        bdma16b(srd_w, vw, so, dst);
        bdma16b(srd_w, vw, so + rs, dst + 8192);
    };

    const int wm = gw >> 1, wn = gw & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int tp_log2 = a.pw_log2 + a.ph_log2;
    const int rotIW = (a.pw_log2 == 3 && !DENSE) ? a.IW : 0;
    auto tile_x = [&](int rr) __attribute__((always_inline)) { return (rr - (rr >> 3) * rotIW) & (PW - 1); };

    // ---- A fragment addresses (igemm_tc5.hip), relative to the group's window buffers
    uint32_t aoff[9][TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int ip = r >> tp_log2;
        const int rr = r & ((1 << tp_log2) - 1);
        const int base = ip * IHW + (rr >> a.pw_log2) * a.IW + tile_x(rr);
#pragma unroll
        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                uint32_t v;
                if constexpr (DENSE) {
                    const int yy = (rr >> 3) + sy - 1, xx = (rr & 7) + sx - 1;
                    const int pix = ip * 64 + yy * 8 + xx;
                    const bool in = (unsigned)yy < (unsigned)a.Hi && (unsigned)xx < (unsigned)a.Wi;
                    v = in ? (uint32_t)((pix << 7) + ((fq ^ (pix & 6)) << 4)) : (uint32_t)(ZERO_OFF + (fq << 4));
                } else {
                    const int pix = base + sy * a.IW + sx;
                    v = (uint32_t)((pix << 7) + ((fq ^ (pix & 6)) << 4));
                }
                aoff[sy * 3 + sx][tm] = v + (uint32_t)(grp2 * 2 * WINB);
            }
    }
    const uint32_t boff = (uint32_t)(WBUF0 + (wn * (BN / WN) + frow) * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));

    f4 acc[TN][TM];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    float vsum = 0.f, vsq = 0.f;
    h8 af[2][TM], bf[2][TN];                         // fragments of the step: read in its L phase, used in its M phase

    auto frag_reads = [&](auto SHIFT_, auto PB_, int stage) __attribute__((always_inline)) {
        constexpr int sh = decltype(SHIFT_)::value, pb = decltype(PB_)::value;
        const char* Ps = smem + pb * WINB;
        const uint32_t wb = boff + (uint32_t)stage * W_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                uint32_t ao = aoff[sh][tm];
                if (ks == 1) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(ao) : "v"(aoff[sh][tm]));
                af[ks][tm] = *(const h8*)(Ps + ao);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[ks][tn] = *(const h8*)(smem + ((wb ^ (ks * 64)) + tn * 2048));
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][tn], af[ks][tm], acc[tn][tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of one class (igemm_tc5.hip)
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? (grp * a.IPB) / a.st.group_n : 0;
    auto epilogue = [&](int cls) __attribute__((always_inline)) {
        const int cy = cls >> 1, cx = cls & 1;
        const int Yc = a.cls[cls].Yc, Xc = a.cls[cls].Xc;
        int64_t opix[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int r = wm * (BM / WM) + tm * 16 + frow;
            const int ip = r >> tp_log2;
            const int rr = r & ((1 << tp_log2) - 1);
            const int n = grp * a.IPB + ip;
            const int y = y0 + (rr >> a.pw_log2), x = x0 + tile_x(rr);
            opix[tm] = (!live || n >= a.N || y >= Yc || x >= Xc) ? -1
                                                                 : ((int64_t)n * a.Ho + (y * 2 + cy)) * a.Wo + (x * 2 + cx);
        }
        const float* gmean = nullptr;
        const float* grstd = nullptr;
        int gimg0 = 0;
        if constexpr (STATS == 2) bn_bwd_group(a.bb, sgrp, gmean, grstd, gimg0);
        const int64_t xshift = STATS == 2 ? (int64_t)(gimg0 - sgrp * a.st.group_n) * a.Ho * a.Wo : 0;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int co = co0 + wn * (BN / WN) + tn * 16 + fq * 4;
            if (co >= a.CoStore) continue;
            f4 mu, rs, ga, be;
            f4 s0 = (f4){0.f, 0.f, 0.f, 0.f}, s1 = (f4){0.f, 0.f, 0.f, 0.f};
            if constexpr (STATS == 2) {
                mu = *(const f4*)(gmean + co);
                rs = *(const f4*)(grstd + co);
                ga = *(const f4*)(a.bb.gamma + co);
                be = *(const f4*)(a.bb.beta + co);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if (opix[tm] < 0) continue;
                const f4 v = acc[tn][tm];
                h4 hv;
                if constexpr (STATS == 2) {
                    const h4 xr = *(const h4*)(a.bb.x + (opix[tm] + xshift) * a.CoStore + co);
                    hv = bn_bwd_mask4(v, xr, mu, rs, ga, be, a.bb.relu, s0, s1);
                } else {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)(co + rg < a.Co ? v[rg] : 0.f);
                    if constexpr (STATS == 1) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const float f = (float)hv[rg];
                            s0[rg] += f;
                            s1[rg] += f * f;
                        }
                    }
                }
                *(h4*)(a.out + opix[tm] * a.CoStore + co) = hv;
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

    // ---- the K-step sequence.  Weight tile of global step g lives in ring stage g % 3 and is issued two steps ahead.
    const int nch = a.nchunks;
    const int Ci2 = a.Ci * 2;
    int wst = 0;                                     // ring stage of the current step
    uint32_t vw, sw, rs;
    class_w(0, vw, sw, rs);

    // prologue: both windows' first chunk, weight tiles of steps 0 and 1 (class 0: taps 0 and 1 of chunk 0)
    if constexpr (DENSE) {
        if (gtid < 16)
            *(f4*)(smem + grp2 * 2 * WINB + (gtid >> 3) * WINB + ZERO_OFF + (gtid & 7) * 16) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    load_slices(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NSL>{});
    load_w(0, vw, sw, rs);
    load_w(1, vw, sw + (uint32_t)Ci2, rs);
    wait_vmb<0>();
    phase_barrier();
    if (grp2 == 1) phase_barrier();                  // group 1 runs one phase behind

    auto run_class = [&](auto CLS_) __attribute__((always_inline)) {
        constexpr int cls = decltype(CLS_)::value;
        constexpr int TH = (cls >> 1) ? 2 : 3, TW = (cls & 1) ? 2 : 3, T = TH * TW;
        constexpr int SPT = (8 + T - 2) / (T - 1);        // window slices per tap, none behind the last tap of a chunk
        constexpr bool LAST = cls == 3;
        constexpr int TNEXT = LAST ? 1 : (((cls + 1) & 1) ? 2 : 3);     // taps per tap row of the next class (unused)
        (void)TNEXT;
        uint32_t vwn = 0, swn = 0, rsn = 0;
        if constexpr (!LAST) class_w(cls + 1, vwn, swn, rsn);
        for (int chunk = 0; chunk < nch; chunk += 2) {
            static_for_b<0, 2>([&](auto PB_) __attribute__((always_inline)) {
                constexpr int pb = decltype(PB_)::value;
                const int ch = chunk + pb;
                const bool last_chunk = ch + 1 >= nch;
                const bool more_win = !(LAST && last_chunk);
                static_for_b<0, T>([&](auto TAP_) __attribute__((always_inline)) {
                    constexpr int t = decltype(TAP_)::value;
                    constexpr int ty = t / TW, tx = t % TW;
                    // ================= L phase: DMA for step + 2 and for the next window, then this step's fragments
                    int wst2 = wst + 2;
                    if (wst2 >= WSTAGES) wst2 -= WSTAGES;
                    bool wissued = true;
                    if constexpr (t + 2 < T) {
                        load_w(wst2, vw, sw + (uint32_t)((t + 2) * Ci2 + ch * 128), rs);
                    } else {
                        constexpr int t2 = t + 2 - T;                        // tap of the following chunk / class
                        if (!last_chunk) load_w(wst2, vw, sw + (uint32_t)(t2 * Ci2 + (ch + 1) * 128), rs);
                        else if constexpr (!LAST) load_w(wst2, vwn, swn + (uint32_t)(t2 * Ci2), rsn);
                        else wissued = false;
                    }
                    constexpr int lo = t * SPT;
                    constexpr int nsl = (t == T - 1 || lo >= NSL) ? 0 : ((lo + SPT > NSL ? NSL : lo + SPT) - lo);
                    if constexpr (nsl > 0) {
                        if (more_win)
                            load_slices(std::integral_constant<int, pb ^ 1>{}, last_chunk ? 0 : ch + 1,
                                        std::integral_constant<int, lo>{}, std::integral_constant<int, lo + nsl>{});
                    }
                    frag_reads(std::integral_constant<int, (2 - ty) * 3 + (2 - tx)>{}, std::integral_constant<int, pb>{}, wst);
                    // leave only this phase's own pieces in flight: everything older has a full K-step behind it
                    if (wissued) {
                        if constexpr (nsl > 0) { if (more_win) wait_vmb<2 + nsl>(); else wait_vmb<2>(); }
                        else wait_vmb<2>();
                    } else {
                        wait_vmb<0>();
                    }
                    phase_barrier();
                    // ================= M phase
                    mfmas();
                    __builtin_amdgcn_sched_barrier(0);
                    phase_barrier();
                    if (++wst == WSTAGES) wst = 0;
                });
            });
        }
        epilogue(cls);
        zero_acc();
        vw = vwn; sw = swn; rs = rsn;
    };
    run_class(std::integral_constant<int, 0>{});
    run_class(std::integral_constant<int, 1>{});
    run_class(std::integral_constant<int, 2>{});
    run_class(std::integral_constant<int, 3>{});
    if (grp2 == 0) phase_barrier();                  // matches group 1's extra barrier at the start

    if constexpr (STATS != 0) {
        // every thread takes part (block barriers inside); only live tiles write their row
        const int prow = tile - sgrp * a.st.tpg[0];
        float* row = a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C;
        float* scratch = (float*)smem + grp2 * 512;
        const int frw = lane & 15, fqq = lane >> 4;
        const int ch = (frw >> 2) * 16 + fqq * 4 + (frw & 3);
        __syncthreads();
        if (wm > 0) { scratch[wn * 128 + ch] = vsum; scratch[wn * 128 + 64 + ch] = vsq; }
        __syncthreads();
        if (wm == 0 && live) {
            const float s = vsum + scratch[wn * 128 + ch], q = vsq + scratch[wn * 128 + 64 + ch];
            const int co = co0 + wn * 64 + ch;
            if (co < a.st.C) { row[co] = s; row[a.st.C + co] = q; }
        }
    }
}

template <int NSL, int STATS>
static int launch_tc5b(const Tc5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_tc5b_kernel<NSL, STATS>;
    constexpr int lds = 4 * (NSL == 4 ? 4 * 4096 + 128 : NSL * 4096) + 3 * 128 * 128;
    static_assert(lds <= 160 * 1024, "one block per CU");
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.ntiles + 1) / 2, copad / 128, 1), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

template <int NSL>
static int launch_tc5bs(const Tc5Args& a, int copad, hipStream_t st) {
    if (!a.st.part) return launch_tc5b<NSL, 0>(a, copad, st);
    return a.bb.x ? launch_tc5b<NSL, 2>(a, copad, st) : launch_tc5b<NSL, 1>(a, copad, st);
}

// 128-channel tiles only, an even number of 64-channel chunks; otherwise E_UNSUPPORTED (igemm_tc5.hip takes it)
int igemm_tc5b_launch(const Tc5Args& a, int bn_tile, int copad, hipStream_t st) {
    if (bn_tile != 128 || a.bias != nullptr || a.act != ACT_NONE || (a.nchunks & 1) || a.ntiles < 2) return E_UNSUPPORTED;
    if (a.nslice == 4) {
        if (a.Hi > 8 || a.Wi > 8 || a.IPB != 2 || a.tiles_x != 1 || a.tiles_y != 1) return E_BADARG;
        return launch_tc5bs<4>(a, copad, st);
    }
    if (a.nslice == 6) return launch_tc5bs<6>(a, copad, st);
    if (a.nslice == 7) return launch_tc5bs<7>(a, copad, st);
    return E_UNSUPPORTED;
}

}  // namespace fmri
