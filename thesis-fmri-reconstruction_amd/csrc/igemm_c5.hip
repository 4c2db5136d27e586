// Stride-2 convolution (k5 p2), Ci % 32 == 0, 128-channel output tiles, on MFMA (gfx950): the input WINDOW of a tile of
// 128 output pixels stays in LDS and is read at stride 1, only the weight tiles stream.
//
// Replaces (reference models/vae_gan.py): the forward of every Conv2d(k5, s2, p2) with >= 32 input channels
// (encoder.conv.1/2 :18-20, discriminator.conv.1/2/3 :149-153) and the data gradient of every
// ConvTranspose2d(k5, s2, p2) (decoder.conv.0/1/2 :46-53, :112-116 -- a stride-2 convolution of the cotangent).
//
// The tap-list kernel (igemm.hip) re-gathers the A operand for each of the 25 taps: half of its L2 -> LDS bytes and DMA
// instructions are the same input pixels again, read at a 2-pixel stride (64 useful bytes per 128-byte line at
// Ci = 32).  Here
//   * out(y, x) = sum over (ky, kx) of in(2y + ky - 2, 2x + kx - 2): rows of one parity (ky even: 15 taps, ky odd: 10)
//     are one PHASE; a phase's window = the (PH + 2) input rows of that parity x (2 PW + 3) columns x 32 channels
//     (<= 24 KB), with the columns split into their two parities inside a row, so that tap (ky, kx) is the 16
//     consecutive pixels  [row y + ky/2][plane kx & 1][x + kx/2 ...]  of the window: unit-stride, conflict-free
//     ds_read_b128 (16-B chunk ^ 2*bit2(column position); checked for all taps and both tile shapes);
//   * the window of the next phase / next 32-channel sub-chunk is DMA'd in 4 KB slices behind the weight tiles of the
//     current one (two window buffers, counted vmcnt); a K-step = 2 taps x 32 channels = 16 KB of weights for 32 MFMAs
//     per wave, 13 steps per sub-chunk; weights come straight out of the [co][tap * Ci + ci] matrix igemm.hip uses;
//   * tap loops, ring stage and window buffer are compile-time constants: every ds_read address is one of 20 per-lane
//     registers (5 kx x 4 row tiles) plus an immediate; all DMA through buffer descriptors with hardware zero fill
//     (out-of-image pixels carry an out-of-range offset), as in igemm_tc5.hip;
//   * BatchNorm batch statistics of the stored values leave the block as one row of the partial-sum buffer (StatEpi).
// Tiles: 8 x 16 output pixels of one image (Wo > 8) or 8 x 8 of two images; 4 waves (2 x 2), two blocks per CU.
#include "kernels.h"
#include <type_traits>

namespace fmri {


namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for_c(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_c<I + 1, N>(f);
    }
}

// 16-byte buffer -> LDS DMA (see igemm_tc5.hip::bdma16)
__device__ __forceinline__ void cdma16(v4i srd, uint32_t voff, uint32_t soff, uint32_t lds) {
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
__device__ __forceinline__ void wait_vmc() {
    static_assert(N >= 0 && N <= 15, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the 25 taps in phase order: 15 taps of the even rows (ky = 0, 2, 4), then 10 of the odd rows (ky = 1, 3)
constexpr int c5_ky(int i) { return i < 15 ? 2 * (i / 5) : 1 + 2 * ((i - 15) / 5); }
constexpr int c5_kx(int i) { return i < 15 ? i % 5 : (i - 15) % 5; }

}  // namespace

// PW: tile width (16: one image, 8: two images).  STATS: 0 none, 1 BatchNorm forward statistics (StatEpi),
// 2 BatchNorm backward statistics + ReLU mask (BnBwdEpi).
template <int PW, int STATS>
__global__ __launch_bounds__(256, 2) void igemm_c5_kernel(const C5Args a) {
    constexpr int BM = 128, BN = 128, WM = 2, WN = 2, TM = 4, TN = 4;
    constexpr int PH = 8;
    constexpr int ROW = 2 * PW + 3;                       // window pixels per row: plane 0 (PW + 2), plane 1 (PW + 1)
    constexpr int IPB = PW == 16 ? 1 : 2;
    constexpr int IMG = PW == 16 ? (PH + 2) * ROW : 192;  // window pixels per image (two images: a multiple of 8)
    constexpr int NSL = 6;                                // 4 KB DMA slices per window
    constexpr int WINB = NSL * 4096;
    constexpr int W_BYTES = 2 * 8192;                     // two tap slots of [128 co][32 ch]
    constexpr int WBUF0 = 2 * WINB;
    static_assert(IPB * IMG * 64 <= WINB, "window fits 6 slices");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    // the block's contiguous tile range (persistent over tpb tiles: the next tile's window and weights are prefetched
    // behind the current tile's last steps, the store epilogue overlaps the other resident block's MFMAs)
    const int tile0 = bx * a.tpb;
    if (tile0 >= a.ntiles) return;
    const int tile1 = tile0 + a.tpb < a.ntiles ? tile0 + a.tpb : a.ntiles;
    const int co0 = by * BN;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- tile -> (image group, tile origin)
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
    // sub-chunk) of window pixel p = q >> 2 = image ip, row j, column position ii; cc = (q & 3) ^ 2*bit2(ii).
    // Column position ii < PW + 2: input column 2*x0 - 2 + 2*ii; else 2*x0 - 1 + 2*(ii - PW - 2).  Row j of phase rp:
    // input row 2*y0 - 2 + rp + 2*j (the odd phase has PH + 1 rows).
    uint32_t soff0[NSL], soff1[NSL];
    uint32_t wstat[NSL];                  // tile-independent part: column term | row << 8 | image << 12 | cc << 13 | valid << 15
#pragma unroll
    for (int e = 0; e < NSL; ++e) {
        const int q = e * 256 + tid;
        const int p = q >> 2;
        const int ip = p / IMG;
        const int rem = p - ip * IMG;
        const int j = rem / ROW;
        const int ii = rem - j * ROW;
        const int cp = ii >= PW + 2 ? 1 : 0;
        const int m = ii - cp * (PW + 2);
        const int cc = (q & 3) ^ (((ii >> 2) & 1) << 1);
        const int valid = (ip < IPB && j < PH + 2) ? 1 : 0;
        wstat[e] = (uint32_t)((2 * m + cp) | (j << 8) | ((ip & 1) << 12) | (cc << 13) | (valid << 15));
    }
    auto tile_offsets = [&](int g, int yy, int xx) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            soff0[e] = soff1[e] = 0x80000000u;             // out of range -> the DMA writes zeros
            const uint32_t ws = wstat[e];
            const int j = (ws >> 8) & 15, ip = (ws >> 12) & 1, cc = (ws >> 13) & 3;
            const int n = g * IPB + ip;
            const int ix = 2 * xx - 2 + (int)(ws & 255);
            const int iy = 2 * yy - 2 + 2 * j;
            if ((ws >> 15) && n < a.N && (unsigned)ix < (unsigned)a.Wi) {
                const uint32_t o = (uint32_t)((((n * a.Hi + iy) * a.Wi + ix) * a.Ci + cc * 8) * 2);
                if ((unsigned)iy < (unsigned)a.Hi) soff0[e] = o;
                if (j < PH + 1 && (unsigned)(iy + 1) < (unsigned)a.Hi) soff1[e] = o + (uint32_t)(a.Wi * a.Ci * 2);
            }
        }
    };
    tile_offsets(grp, y0, x0);
    const uint32_t lds_wave = lds0 + wave * 1024;
    // slices [lo, hi) of sub-chunk `sub`, phase RP, into window buffer RP
    auto load_slices = [&](auto RP_, int sub, auto LO_, auto HI_) __attribute__((always_inline)) {
        constexpr int rp = decltype(RP_)::value, lo = decltype(LO_)::value, hi = decltype(HI_)::value;
        const uint32_t so = (uint32_t)sub * 64u;
#pragma unroll
        for (int e = lo; e < hi; ++e)
            if (e < NSL) cdma16(srd_in, rp ? soff1[e] : soff0[e], so, lds_wave + rp * WINB + e * 4096);
    };

    // ---- weight DMA: tap slot = [128 co][32 ch] = 8 KB, 64 rows per block instruction; chunk swizzle 2*bit2(row)
    const int trow = tid >> 2;
    const int wcc = (tid & 3) ^ (((trow >> 2) & 1) << 1);
    const uint32_t vw = (uint32_t)(((co0 + trow) * a.Kpad + wcc * 8) * 2);
    const uint32_t rs64 = (uint32_t)(a.Kpad * 128);      // 64 rows
    const int Ci2 = a.Ci * 2;
    // taps [T0, T0 + 2) (those < 25) of sub-chunk `sub` into ring stage STG
    auto load_w = [&](auto STG_, auto T0_, int sub) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG_)::value, t0 = decltype(T0_)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (t0 + s < 25) {
                const int tap = c5_ky(t0 + s) * 5 + c5_kx(t0 + s);
                const uint32_t so = (uint32_t)(tap * Ci2 + sub * 64);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    cdma16(srd_w, vw, so + i * rs64, lds_wave + WBUF0 + stg * W_BYTES + s * 8192 + i * 4096);
            }
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- A fragment addresses: abase[kx][tm] for window row shift 0; row shift sy adds the immediate sy * ROW * 64
    uint32_t abase[5][TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int y = r >> 4;
        const int ip = PW == 16 ? 0 : (r >> 3) & 1;
        const int x = PW == 16 ? (r & 15) : (r & 7);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const int ii = (kx & 1) * (PW + 2) + x + (kx >> 1);
            const int p = ip * IMG + y * ROW + ii;
            abase[kx][tm] = (uint32_t)((p << 6) + ((fq ^ (((ii >> 2) & 1) << 1)) << 4));
        }
    }
    // ---- B fragment address (row = wn*64 + tn*16 + frow)
    const uint32_t boff = (uint32_t)(WBUF0 + (wn * (BN / WN) + frow) * 64 + ((fq ^ (((frow >> 2) & 1) << 1)) << 4));

    f4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    // one K-step: taps T0, T0 + 1 (those < 25), weights in ring stage STG
    auto compute = [&](auto T0_, auto STG_) __attribute__((always_inline)) {
        constexpr int t0 = decltype(T0_)::value, stg = decltype(STG_)::value;
        constexpr int NS = t0 + 1 < 25 ? 2 : 1;
        h8 af[NS][TM], bf[NS][TN];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ky = c5_ky(t0 + s), kx = c5_kx(t0 + s);
            const char* Ps = smem + (ky & 1) * WINB + (ky >> 1) * (ROW * 64);
            const char* Ws = smem + stg * W_BYTES + s * 8192;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[s][tm] = *(const h8*)(Ps + abase[kx][tm]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[s][tn] = *(const h8*)(Ws + (boff + tn * 1024));
            if (s == 0) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s][tn], af[s][tm], acc[tn][tm], 0, 0, 0);
            if (s == 0 && NS == 2) {
                // the second slot's 8 fragment reads one by one between the first slot's MFMAs
#pragma unroll
                for (int i = 0; i < TM + TN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- one 32-channel sub-chunk: 13 steps.  P = parity of the sub-chunk (ring stage of step t = (P + t) & 1).
    // Window traffic: steps 0-5 bring this sub-chunk's odd-row window (buffer 1, first read in step 7); steps 8-11 the
    // next sub-chunk's even-row window (buffer 0, last read in step 7).
    const int nsub = a.nsub;
    // `more`: another (tile, sub-chunk) follows; `nsubi` = its sub-chunk index; `ng/ny/nx` = geometry of the next tile
    // when this is the last sub-chunk of a tile that has a successor (switch: the window offsets are recomputed in step 6,
    // after this tile's last odd-row slice has been issued)
    // `landed`: the DMA this sub-chunk's first step waits for was already waited for in front of the previous tile's
    // store epilogue, so that the stores (which count in vmcnt too) get this step to complete instead of stalling it
    auto run_sub = [&](auto P_, int sub, bool more, int nsubi, bool switch_tile, int ng, int ny, int nx, bool landed)
                       __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        static_for_c<0, 13>([&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value;
            constexpr int stg = (P + t) & 1;
            // window slices issued behind the weight tiles of the PREVIOUS step
            constexpr int prev_n = (t >= 1 && t <= 6) ? 1 : ((t == 9 || t == 10) ? 2 : ((t == 11 || t == 12) ? 1 : 0));
            constexpr bool prev_cond = t >= 9;             // ... only when another (tile, sub-chunk) follows
            if constexpr (t == 0) { if (!landed) wait_vmc<0>(); }
            else if constexpr (prev_n == 0) wait_vmc<0>();
            else if constexpr (prev_cond) { if (more) wait_vmc<prev_n>(); else wait_vmc<0>(); }
            else wait_vmc<prev_n>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // next step's weight tiles
            if constexpr (t < 12) load_w(std::integral_constant<int, stg ^ 1>{}, std::integral_constant<int, 2 * t + 2>{}, sub);
            else if (more) load_w(std::integral_constant<int, stg ^ 1>{}, std::integral_constant<int, 0>{}, nsubi);
            // window slices (a whole window in one step instead: 6 % faster at Ci = 32, whose windows miss L2, and 4 %
            // slower at Ci >= 128, where the burst delays the weight tiles)
            if constexpr (t <= 5) {
                load_slices(std::integral_constant<int, 1>{}, sub, std::integral_constant<int, t>{}, std::integral_constant<int, t + 1>{});
            } else if constexpr (t == 6) {
                if (switch_tile) tile_offsets(ng, ny, nx);
            } else if constexpr (t == 8 || t == 9) {
                if (more)
                    load_slices(std::integral_constant<int, 0>{}, nsubi, std::integral_constant<int, (t - 8) * 2>{},
                                std::integral_constant<int, (t - 8) * 2 + 2>{});
            } else if constexpr (t == 10 || t == 11) {
                if (more)
                    load_slices(std::integral_constant<int, 0>{}, nsubi, std::integral_constant<int, t - 6>{},
                                std::integral_constant<int, t - 5>{});
            }
            compute(std::integral_constant<int, 2 * t>{}, std::integral_constant<int, stg>{});
        });
    };

    // ---- epilogue: D[i = co][j = tile pixel] -> NHWC fp16 (no bias / activation: BatchNorm or a data gradient follows)
    float vsum = 0.f, vsq = 0.f;         // over all tiles of the block
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? (grp * IPB) / a.st.group_n : 0;     // statistics group of the tile
    auto epilogue = [&]() __attribute__((always_inline)) {
        int64_t opix[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int r = wm * (BM / WM) + tm * 16 + frow;
            const int ip = PW == 16 ? 0 : (r >> 3) & 1;
            const int n = grp * IPB + ip;
            const int y = y0 + (r >> 4), x = x0 + (PW == 16 ? (r & 15) : (r & 7));
            opix[tm] = (n >= a.N || y >= a.Ho || x >= a.Wo) ? -1 : ((int64_t)n * a.Ho + y) * a.Wo + x;
        }
        // BnBwdEpi: the saved forward tensor starts x_img0[group] images in, the cotangent group_n * group images
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
                        // statistics of the STORED (fp16-rounded) values: what the consumers and the BN backward see
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

    // prologue: even-row window of the first tile's sub-chunk 0 and the first weight tiles
    load_slices(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NSL>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
    // (tile, sub-chunk) pairs in one sequence, two per iteration: 13 steps per sub-chunk, so the ring stage parity
    // alternates along the sequence whatever nsub is
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
            if (next_tile) { wait_vmc<0>(); landed = true; }     // the next tile's first window and weights
            epilogue();
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
    if constexpr (STATS != 0) {
        // one row per block; the blocks of one statistics group are contiguous (tpg[0] = blocks per group)
        const int prow = bx - sgrp * a.st.tpg[0];
        stat_store<TN, WM, WN>(vsum, vsq, lane, wm, wn, co0, (float*)smem,
                               a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C, a.st.C);
    }
}

template <int PW, int STATS>
static int launch_c5(const C5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_c5_kernel<PW, STATS>;
    constexpr int lds = 2 * 6 * 4096 + 2 * 16384;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    if (route_probe("fmri::igemm_c5_kernel<%d,%d>", PW, STATS)) return OK;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.ntiles + a.tpb - 1) / a.tpb, copad / 128, 1), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// k5 s2 p2, Ci % 32 == 0, 128-channel tiles, no bias / activation; pw16 selects the tile shape
int igemm_c5_launch(const C5Args& a, int copad, hipStream_t st) {
    if (a.nsub < 1 || (copad & 127) || a.ntiles < 1 || a.tpb < 1) return E_UNSUPPORTED;
    if (a.pw16) {
        if (!a.st.part) return launch_c5<16, 0>(a, copad, st);
        return a.bb.x ? launch_c5<16, 2>(a, copad, st) : launch_c5<16, 1>(a, copad, st);
    }
    if (!a.st.part) return launch_c5<8, 0>(a, copad, st);
    return a.bb.x ? launch_c5<8, 2>(a, copad, st) : launch_c5<8, 1>(a, copad, st);
}


}  // namespace fmri
