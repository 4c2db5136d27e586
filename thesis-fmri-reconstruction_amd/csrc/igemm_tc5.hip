// Stride-2 transposed convolution (k5 p2) with Ci % 128 == 0 and >= 64 output channels on MFMA (gfx950): all four
// output-parity classes of a tile in ONE block, tap loops resolved at compile time.
//
// Replaces (reference models/vae_gan.py): ConvTranspose2d(k5, s2, p2, output_padding 0/1) forward of
// decoder.conv.0/1 (:46-53, :112-116) and the data gradient of every Conv2d(k5, s2, p2) with >= 128 output channels
// (encoder.conv.1/2 :18-20, discriminator.conv.2/3 :149-153).
//
// igemm_win.hip ran one parity class per block (grid.z = class): every class streamed the input again from HBM
// (4.3 x input re-read measured) and its K loop spent ~5.5 scalar / vector instructions per MFMA on tap bookkeeping,
// swizzled window addresses recomputed per tap and 64-bit DMA addresses -- two waves per SIMD could not keep the
// matrix pipe fed (60 % of what a bare MFMA loop sustains).  Here
//   * the four classes (3x3, 3x2, 2x3, 2x2 unit-shift taps) read the SAME (PH+2) x (PW+2)-pixel window of 128
//     class-grid positions; a 256-thread block runs them back to back over one continuous DMA pipeline (weights: 2-stage
//     ring, window: 64-channel chunks double-buffered, slices spread behind the weight tiles, counted vmcnt): the input
//     leaves HBM once, the later classes' window DMAs hit in the XCD's L2;
//   * classes, taps, ring stage and window buffer are template / unrolled-loop constants, so every ds_read address is
//     one of 36 per-lane registers computed once (9 window shifts x 4 row tiles) plus an immediate, the second 32-wide
//     half of a K-step is one XOR away, and the K loop carries no per-tap vector arithmetic;
//   * all DMA goes through buffer descriptors (buffer_load_dwordx4 ... lds): per-lane 32-bit offsets that never change,
//     scalar offsets for (class, tap, chunk), and the hardware's bounds check returns the zeros of the padding ring
//     (out-of-image window pixels carry an out-of-range offset) -- no zero page, no selects, no 64-bit adds;
//   * BatchNorm batch statistics (sum x, sum x^2 per output channel over the valid pixels of all four classes) are
//     accumulated in registers across the classes and leave the block as one row of a partial-sum buffer.
#include "kernels.h"
#include <type_traits>

namespace fmri {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// 16-byte buffer -> LDS DMA: LDS destination = wave-uniform `lds` + lane*16, source = descriptor base + voff + soff.
// Offsets >= num_records read as zero.  Issued from inline asm so that the compiler does not serialise later LDS reads
// behind it (see glds16_raw in common.h); the caller owns the vmcnt / barrier protocol.
__device__ __forceinline__ void bdma16(v4i srd, uint32_t voff, uint32_t soff, uint32_t lds) {
    // under scalar-register pressure the compiler parks the descriptor in vector registers and would hand those to the
    // "s" operand: name every word wave-uniform (free when it already sits in SGPRs)
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
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}

}  // namespace

// NSL = 4 KB DMA slices per window chunk: 6 -> two window buffers (80 KB LDS with the 128-row weight ring),
// 7 (two 10 x 10 image windows per tile) -> one buffer, reloaded between chunks; 4 -> dense 8 x 8 windows (see DENSE).
// STATS: 0 none, 1 BatchNorm forward statistics (StatEpi), 2 BatchNorm backward statistics + ReLU mask (BnBwdEpi)
template <int BN, int NSL, int STATS>
__global__ __launch_bounds__(256, 2) void igemm_tc5_kernel(const Tc5Args a) {
    constexpr int BM = 128, WM = 2, WN = 2;
    constexpr int TM = BM / WM / 16;            // 4
    constexpr int TN = BN / WN / 16;            // 4 (BN = 128) or 2 (BN = 64)
    constexpr int BROWS = BN / 32;
    constexpr int W_BYTES = BN * 128;
    // NSL == 4 ("dense"): the class grid fits one 8 x 8 tile per image (input <= 8 x 8), so the 1-pixel halo of the window is
    // all padding: only the 8 x 8 interior (2 images x 64 pixels x 128 B = 4 slices) is kept, and fragment addresses that
    // fall into the halo point at a 128-byte block of zeros behind each window buffer.
    constexpr bool DENSE = NSL == 4;
    constexpr int WINB = DENSE ? 4 * 4096 + 128 : NSL * 4096;
    constexpr int ZERO_OFF = 4 * 4096;               // dense: zeros at [ZERO_OFF, +128) of every window buffer
    constexpr int PBUFS = NSL <= 6 ? 2 : 1;
    constexpr int WBUF0 = PBUFS * WINB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    if (bx >= a.ntiles) return;
    const int co0 = by * BN;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- tile -> (image group, tile row, tile column) of the class grid
    const int tpi = a.tiles_y * a.tiles_x;
    const int grp = (int)fd_div((uint32_t)bx, a.fdTPI);
    const int trem = bx - grp * tpi;
    const int tyi = (int)fd_div((uint32_t)trem, a.fdTX);
    const int txi = trem - tyi * a.tiles_x;
    const int PW = 1 << a.pw_log2;
    const int y0 = tyi * a.PH, x0 = txi * PW;
    const int IHW = a.IH * a.IW;

    // ---- descriptors
    v4i srd_in, srd_w;
    srd_in.x = (int)(uint32_t)(uintptr_t)a.in;
    srd_in.y = (int)(uint32_t)((uintptr_t)a.in >> 32);
    srd_in.z = (int)a.in_bytes;
    srd_in.w = 0x00020000;
    srd_w.x = (int)(uint32_t)(uintptr_t)a.w;
    srd_w.y = (int)(uint32_t)((uintptr_t)a.w >> 32);
    srd_w.z = (int)a.w_bytes;
    srd_w.w = 0x00020000;

    // ---- window DMA: slice e covers LDS bytes [e*4096, +4096) of a window buffer; 16-B unit q = e*256 + tid holds
    // channels 8*cc .. 8*cc+7 of window pixel q >> 3, cc = (q & 7) ^ (pixel & 6)  (conflict-free for every tap shift,
    // see igemm_win.hip).  Window origin = tile origin - 1 (the union window of the four classes).
    uint32_t soff[NSL];
    {
        const FastDiv fIHW = a.fdIHW, fIW = a.fdIW;
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            soff[e] = 0x80000000u;                         // out of range -> the DMA writes zeros
            const int q = e * 256 + tid;
            const int pixel = q >> 3;
            int ip, j, i;
            if constexpr (DENSE) {
                ip = pixel >> 6; j = ((pixel >> 3) & 7) + 1; i = (pixel & 7) + 1;     // interior only, origin (0, 0)
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
    const uint32_t lds_wave = lds0 + wave * 1024;
    // slices [lo, hi) of channel chunk `chunk` into window buffer `buf`
    auto load_slices = [&](auto BUF_, int chunk, auto LO_, auto HI_) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF_)::value, lo = decltype(LO_)::value, hi = decltype(HI_)::value;
        const uint32_t so = (uint32_t)chunk * 128u;
#pragma unroll
        for (int e = lo; e < hi; ++e)
            if (e < NSL) bdma16(srd_in, soff[e], so, lds_wave + buf * WINB + e * 4096);
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
        for (int i = 0; i < BROWS; ++i) bdma16(srd_w, vw, so + i * rs, lds_wave + WBUF0 + stg * W_BYTES + i * 4096);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int tp_log2 = a.pw_log2 + a.ph_log2;
    // GEMM row rr of an image's tile -> tile column (8-wide tiles: rows rotated by -y*IW, see igemm_win.hip)
    const int rotIW = (a.pw_log2 == 3 && !DENSE) ? a.IW : 0;     // dense 8-pixel rows are conflict-free as they are
    auto tile_x = [&](int rr) __attribute__((always_inline)) { return (rr - (rr >> 3) * rotIW) & (PW - 1); };

    // ---- A fragment addresses: window shift (sy, sx) in 0..2 of row tile tm; ks = 1 is the same address ^ 64
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
                if constexpr (DENSE) {
                    const int yy = (rr >> 3) + sy - 1, xx = (rr & 7) + sx - 1;
                    const int pix = ip * 64 + yy * 8 + xx;
                    const bool in = (unsigned)yy < (unsigned)a.Hi && (unsigned)xx < (unsigned)a.Wi;
                    aoff[sy * 3 + sx][tm] = in ? (uint32_t)((pix << 7) + ((fq ^ (pix & 6)) << 4))
                                               : (uint32_t)(ZERO_OFF + (fq << 4));
                } else {
                    const int pix = base + sy * a.IW + sx;
                    aoff[sy * 3 + sx][tm] = (uint32_t)((pix << 7) + ((fq ^ (pix & 6)) << 4));
                }
            }
    }
    // ---- B fragment address (row = wn*(BN/WN) + tn*16 + frow; the swizzle term does not depend on tn or wn)
    const uint32_t boff = (uint32_t)((wn * (BN / WN) + frow) * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));

    f4 acc[TN][TM];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // BatchNorm statistics of the lane's channel (see stat_lane_add) over the block's valid pixels, all classes
    float vsum = 0.f, vsq = 0.f;

    auto compute = [&](auto SHIFT_, auto PB_, auto STG_) __attribute__((always_inline)) {
        constexpr int sh = decltype(SHIFT_)::value, pb = decltype(PB_)::value, stg = decltype(STG_)::value;
        const char* Ps = smem + pb * WINB;
        const char* Ws = smem + WBUF0 + stg * W_BYTES;
        h8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                uint32_t ao = aoff[sh][tm];
                // the second 32-wide half: one XOR, issued here (volatile: otherwise the compiler precomputes all 36
                // XORed addresses outside the loop and runs out of registers)
                if (ks == 1) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(ao) : "v"(aoff[sh][tm]));
                af[ks][tm] = *(const h8*)(Ps + ao);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[ks][tn] = *(const h8*)(Ws + ((boff ^ (ks * 64)) + tn * 2048));
            if (ks == 0) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][tn], af[ks][tm], acc[tn][tm], 0, 0, 0);
            if (ks == 0) {
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

    // ---- epilogue of one class: D[i = co][j = class-grid position] -> output pixel (2y + cy, 2x + cx).  No bias /
    // activation here: every layer of this geometry is followed by BatchNorm or is a data gradient (the launcher routes
    // anything else to igemm_win.hip).
    const int sgrp = (STATS != 0 && a.st.group_n > 0) ? (grp * a.IPB) / a.st.group_n : 0;     // statistics group of the tile
    auto epilogue = [&](int cls) __attribute__((always_inline)) {
        const int cy = cls >> 1, cx = cls & 1;
        const int Yc = a.cls[cls].Yc, Xc = a.cls[cls].Xc;
        // output pixel (element offset / CoStore) of row tile tm, -1: outside
        int64_t opix[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int r = wm * (BM / WM) + tm * 16 + frow;
            const int ip = r >> tp_log2;
            const int rr = r & ((1 << tp_log2) - 1);
            const int n = grp * a.IPB + ip;
            const int y = y0 + (rr >> a.pw_log2), x = x0 + tile_x(rr);
            opix[tm] = (n >= a.N || y >= Yc || x >= Xc) ? -1 : ((int64_t)n * a.Ho + (y * 2 + cy)) * a.Wo + (x * 2 + cx);
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
            // sums of this lane's 4 channels over its pixels: reduced to the lane's one channel right after the tile
            // column (keeps the epilogue's register peak below what would spill loop-carried values)
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
                // 16-lane row sums; lane (fq, frow) owns channel (frow >> 2)*16 + fq*4 + (frow & 3) (stat_store)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float ra = row16_sum(s0[rg]);
                    const float rb = row16_sum(s1[rg]);
                    if (frow == tn * 4 + rg) { vsum += ra; vsq += rb; }
                }
                // one tile column at a time: without this the compiler issues every column's loads up front and the
                // registers they occupy push loop-carried values into scratch
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- one class: chunk pairs x taps, everything but the chunk counter a compile-time constant
    const int nch = a.nchunks;
    const int Ci2 = a.Ci * 2;
    uint32_t vw, sw, rs;
    class_w(0, vw, sw, rs);
    auto run_class = [&](auto CLS_) __attribute__((always_inline)) {
        constexpr int cls = decltype(CLS_)::value;
        constexpr int TH = (cls >> 1) ? 2 : 3, TW = (cls & 1) ? 2 : 3, T = TH * TW;
        constexpr int SPT = (8 + T - 1) / T;              // window slices issued per tap (covers NSL <= 8)
        constexpr bool LAST = cls == 3;
        uint32_t vwn = 0, swn = 0, rsn = 0;
        if constexpr (!LAST) class_w(cls + 1, vwn, swn, rsn);
        for (int chunk = 0; chunk < nch; chunk += 2) {
            static_for<0, 2>([&](auto PB_) __attribute__((always_inline)) {
                constexpr int pb = decltype(PB_)::value;
                const int ch = chunk + pb;
                const bool last_chunk = ch + 1 >= nch;
                const bool more_win = !(LAST && last_chunk);       // another (class, chunk) window follows
                static_for<0, T>([&](auto TAP_) __attribute__((always_inline)) {
                    constexpr int t = decltype(TAP_)::value;
                    constexpr int ty = t / TW, tx = t % TW;
                    constexpr int stg = (pb * T + t) & 1;
                    // slices issued behind the weight tile of the PREVIOUS step (window of the next chunk)
                    constexpr int prev_lo = t == 0 ? 0 : (t - 1) * SPT;
                    constexpr int prev_n = t == 0 ? 0
                                                  : ((prev_lo >= NSL) ? 0 : ((prev_lo + SPT > NSL ? NSL : prev_lo + SPT) - prev_lo));
                    // weights of this step landed; at the first tap of a chunk the whole window must have landed too
                    // (in the last (class, chunk) no window follows: nothing was issued behind the weight tile)
                    if constexpr (t == 0 || PBUFS == 1 || prev_n == 0) wait_vm<0>();
                    else if constexpr (LAST) { if (more_win) wait_vm<prev_n>(); else wait_vm<0>(); }
                    else wait_vm<prev_n>();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
                    if constexpr (PBUFS == 2) {
                        if (more_win)
                            load_slices(std::integral_constant<int, pb ^ 1>{}, last_chunk ? 0 : ch + 1,
                                        std::integral_constant<int, t * SPT>{}, std::integral_constant<int, t * SPT + SPT>{});
                    }
                    compute(std::integral_constant<int, (2 - ty) * 3 + (2 - tx)>{}, std::integral_constant<int, PBUFS == 2 ? pb : 0>{},
                            std::integral_constant<int, stg>{});
                    if constexpr (PBUFS == 1 && t == T - 1) {
                        // single window buffer: everyone is done with the old window, then reload (exposed once per chunk;
                        // the other block resident on the CU keeps the MFMAs busy meanwhile)
                        if (more_win) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            load_slices(std::integral_constant<int, 0>{}, last_chunk ? 0 : ch + 1, std::integral_constant<int, 0>{},
                                        std::integral_constant<int, NSL>{});
                        }
                    }
                });
            });
        }
        epilogue(cls);
        zero_acc();
        vw = vwn; sw = swn; rs = rsn;
    };

    if constexpr (DENSE) {
        // the zero blocks behind the two window buffers (visible after the first step's barrier)
        if (tid < 16) *(f4*)(smem + (tid >> 3) * WINB + ZERO_OFF + (tid & 7) * 16) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    // prologue: window of (class 0, chunk 0) and the first weight tile
    load_slices(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NSL>{});
    load_w(std::integral_constant<int, 0>{}, vw, sw, rs);
    run_class(std::integral_constant<int, 0>{});
    run_class(std::integral_constant<int, 1>{});
    run_class(std::integral_constant<int, 2>{});
    run_class(std::integral_constant<int, 3>{});

    // ---- BatchNorm statistics of the block (all four classes): its own row of the partial buffer
    if constexpr (STATS != 0) {
        const int prow = bx - sgrp * a.st.tpg[0];
        float* row = a.st.part + ((size_t)sgrp * a.st.rows_cap + prow) * 2 * a.st.C;
        stat_store<TN, WM, WN>(vsum, vsq, lane, wm, wn, co0, (float*)smem, row, a.st.C);
    }
}

template <int BN, int NSL, int STATS>
static int launch_tc5(const Tc5Args& a, int copad, hipStream_t st) {
    auto kern = igemm_tc5_kernel<BN, NSL, STATS>;
    constexpr int PBUFS = NSL <= 6 ? 2 : 1;
    constexpr int lds = PBUFS * (NSL == 4 ? 4 * 4096 + 128 : NSL * 4096) + 2 * BN * 128;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    if (route_probe("fmri::igemm_tc5_kernel<%d,%d,%d>", BN, NSL, STATS)) return OK;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ntiles, copad / BN, 1), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

template <int BN, int NSL>
static int launch_tc5s(const Tc5Args& a, int copad, hipStream_t st) {
    if (!a.st.part) return launch_tc5<BN, NSL, 0>(a, copad, st);
    return a.bb.x ? launch_tc5<BN, NSL, 2>(a, copad, st) : launch_tc5<BN, NSL, 1>(a, copad, st);
}

// bias == null and act == none only (see the epilogue); nslice in {6, 7}; bn_tile in {64, 128}
int igemm_tc5_launch(const Tc5Args& a, int bn_tile, int copad, hipStream_t st) {
    if ((a.nslice != 4 && a.nslice != 6 && a.nslice != 7) || a.bias != nullptr || a.act != ACT_NONE) return E_UNSUPPORTED;
    if (a.nslice == 4 && (a.Hi > 8 || a.Wi > 8 || a.IPB != 2 || a.tiles_x != 1 || a.tiles_y != 1)) return E_BADARG;
    if (bn_tile == 128) {
        if (a.nslice == 4) return launch_tc5s<128, 4>(a, copad, st);
        return a.nslice == 6 ? launch_tc5s<128, 6>(a, copad, st) : launch_tc5s<128, 7>(a, copad, st);
    }
    if (bn_tile == 64) {
        if (a.nslice == 4) return launch_tc5s<64, 4>(a, copad, st);
        return a.nslice == 6 ? launch_tc5s<64, 6>(a, copad, st) : launch_tc5s<64, 7>(a, copad, st);
    }
    return E_UNSUPPORTED;
}

}  // namespace fmri
