// Window-resident weight gradient for stride-2 sampling (gfx950).
//
//   dW[a][tap*Bc + b] += sum_{m=(n,y,x)} P[m][a] * Q[n, 2y + ty - pad, 2x + tx - pad, b]          (tap = ty*k + tx)
//
// (Conv2d stride 2: P = dY, Q = X; ConvTranspose2d stride 2: P = X, Q = dY -- wgrad.hip's contract.)  wgrad.hip
// streams a 64 x 128 P tile AND a gathered 64 x 128 Q tile from L2 per K-step, one Q tile per (tap, 8-channel) column
// block, although neighbouring taps read almost the same pixels.  Here Q is viewed as its four parity planes
// Qp[r][c] = Q[2r+py][2c+px]: for the taps of one parity, tap (ty, tx) of output pixel (y, x) is plane pixel
// (y + t_y, x + t_x) with t in {-1, 0, 1} -- a UNIT shift.  A block owns 128 rows a, 32 channels b and ALL (2..3)^2
// shifts of one plane; per K-step (an 8x8 tile of output pixels) it DMAs the P tile (16 KB) and the (8+2)^2-pixel
// window of the plane (6.4 KB) once and runs 8 * shifts MFMAs per wave from them: 22 KB per 4.7 MFLOP instead of
// 32 KB per 2.1 MFLOP.  The MFMA K index is the pixel, so both operands are read with transposing LDS reads
// (ds_read_b64_tr_b16); the window fragments of a shift are the same reads at a compile-time byte offset.
#include "kernels.h"
#include <type_traits>


namespace fmri {


namespace {
typedef int v4i __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for_g(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_g<I + 1, N>(f);
    }
}

// 16-byte buffer -> LDS DMA (see igemm_tc5.hip::bdma16): offsets >= num_records read as zero
__device__ __forceinline__ void wdma16(v4i srd, uint32_t voff, uint32_t lds) {
    srd.x = __builtin_amdgcn_readfirstlane(srd.x);
    srd.y = __builtin_amdgcn_readfirstlane(srd.y);
    srd.z = __builtin_amdgcn_readfirstlane(srd.z);
    srd.w = __builtin_amdgcn_readfirstlane(srd.w);
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(srd), "s"(lds)
                 : "memory");
}
}  // namespace

// P tile: 64 m-rows x 256 B, swizzled exactly like wgrad.hip.  Window: WH x WW pixels x 64 B, linear.
template <int NSY, int NSX, bool SLABS>
__device__ __forceinline__ void wgrad_win_body(const WgradWinArgs& a, char* smem, int py, int px, int a_tile,
                                               int b_tile, int split) {
    constexpr int WH = 8 + NSY - 1, WW = 8 + NSX - 1;
    constexpr int P_BYTES = 64 * 256;
    constexpr int W_BYTES = ((WH * WW * 64 + 1023) / 1024) * 1024;
    constexpr int STAGE = P_BYTES + W_BYTES;
    constexpr int NS = NSY * NSX;
    constexpr int TA = 4;             // 2 x 2 waves: wave tile = 64 rows a x 16 channels b

    // Round 3: 8 waves, one block per CU.  Waves 4-7 only move bytes (the per-step tile arithmetic, bounds checks and DMA
    // instructions: ~170 of the ~310 instructions a wave used to issue per K-step around its 32-72 MFMAs -- with one wave per
    // SIMD the step was bound by instruction issue, ~1 300 cycles per step whatever the MFMA count), waves 0-3 only read
    // fragments and multiply; one s_barrier per K-step joins them.
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave_all >= 4;
    const int wave = wave_all & 3;                // loader: DMA slice index; compute: tile quadrant
    const int tid = threadIdx.x & 255;
    const int a0 = a_tile * 128;
    const int b0 = b_tile * 32;
    const int tminy = a.tmin[py], tminx = a.tmin[px];

    // ---- P source: thread loads the 16-B chunk `clog` of rows trow + 16 i (i < 4), wgrad.hip's swizzle.
    // Round 3: both operands are DMA'd through buffer descriptors with 32-bit offsets = (per-lane constant) + (scalar
    // of the tile); out-of-image lanes carry the out-of-range offset and read as zero.  Round 2's form built a 64-bit
    // address (and a select against the zero page) per lane and DMA instruction: 330-680 vector + 350-590 scalar
    // instructions per K-step around its 32-72 MFMAs -- the kernel was bound by VALU issue, not by the matrix pipe.
    const int trow = tid >> 4;
    const int cphys = tid & 15;
    const int fsw = (((trow & 3) | (((trow >> 3) & 1) << 2)) << 1);
    const int clog = cphys ^ fsw;
    const bool p_on = a0 + clog * 8 < a.A;
    const int prow = trow >> 3, pcol = trow & 7;                     // tile pixel of row trow (+ 2 pixel rows per i)
    const uint32_t lane_p = (uint32_t)(((prow * a.Xc + pcol) * a.A + a0 + clog * 8) * 2);

    // ---- window source: units u = tid, tid + 256 -> window pixel u >> 2, channels 8 * (u & 3)
    int wj[2], wi[2];
    bool won[2];
    uint32_t lane_q[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int u = tid + 256 * e;
        const int pix = u >> 2;
        won[e] = pix < WH * WW;
        wj[e] = pix / WW;
        wi[e] = pix - wj[e] * WW;
        lane_q[e] = (uint32_t)(((2 * wj[e] * a.Wq + 2 * wi[e]) * a.Bc + b0 + (tid & 3) * 8) * 2);
    }
    v4i srdP, srdQ;
    srdP.x = (int)(uint32_t)(uintptr_t)a.P;
    srdP.y = (int)(uint32_t)((uintptr_t)a.P >> 32);
    srdP.z = (int)((uint32_t)a.N * (uint32_t)a.Yc * (uint32_t)a.Xc * (uint32_t)a.A * 2u);
    srdP.w = 0x00020000;
    srdQ.x = (int)(uint32_t)(uintptr_t)a.Q;
    srdQ.y = (int)(uint32_t)((uintptr_t)a.Q >> 32);
    srdQ.z = (int)((uint32_t)a.N * (uint32_t)a.Hq * (uint32_t)a.Wq * (uint32_t)a.Bc * 2u);
    srdQ.w = 0x00020000;

    // the K range (8x8 pixel tiles) is cut into equal pieces, the same for every plane (see api.hip)
    const int tps = a.plane_tps[py * 2 + px];
    const int t0 = split * tps;
    int t1 = t0 + tps;
    if (t1 > a.ntiles) t1 = a.ntiles;
    const int nsteps = t1 > t0 ? t1 - t0 : 0;
    if (nsteps == 0 && a.slab_stride == 0) return;
    const int tpi = a.tiles_y * a.tiles_x;

    // loop-invariant scalars of the K loop and the epilogue, pinned in SGPRs (see FMRI_KEEP)
    int kYc = a.Yc, kXc = a.Xc, kA = a.A, kHq = a.Hq, kWq = a.Wq, kBc = a.Bc, ktx = a.tiles_x, ktpi = tpi;
    uint32_t tpi_magic = a.fdTPI.magic, tpi_sh = a.fdTPI.sh, tx_magic = a.fdTX.magic, tx_sh = a.fdTX.sh;
    FMRI_KEEP(kYc); FMRI_KEEP(kXc); FMRI_KEEP(kA); FMRI_KEEP(kHq); FMRI_KEEP(kWq); FMRI_KEEP(kBc); FMRI_KEEP(ktx);
    FMRI_KEEP(ktpi); FMRI_KEEP(tpi_magic); FMRI_KEEP(tpi_sh); FMRI_KEEP(tx_magic); FMRI_KEEP(tx_sh);
    const FastDiv fTPI{tpi_magic, tpi_sh, 0, 0}, fTX{tx_magic, tx_sh, 0, 0};
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t prow_step = (uint32_t)(2 * kXc * kA * 2);          // two pixel rows of P

    auto stage_load = [&](int buf, int t) __attribute__((always_inline)) {
        // tile -> (image, tile row, tile col): wave-uniform
        const int n = (int)fd_div((uint32_t)t, fTPI);
        const int trem = t - n * ktpi;
        const int tyi = (int)fd_div((uint32_t)trem, fTX);
        const int txi = trem - tyi * ktx;
        const int y0 = tyi * 8, x0 = txi * 8;
        const uint32_t tile_p = (uint32_t)(((n * kYc + y0) * kXc + x0) * kA * 2);
        const bool pcol_ok = p_on && x0 + pcol < kXc;
        const uint32_t dstP = lds0 + buf * STAGE + wave * (4 * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = pcol_ok && y0 + prow + 2 * i < kYc;
            wdma16(srdP, ok ? tile_p + lane_p + i * prow_step : 0x80000000u, dstP + i * (16 * 256));
        }
        // window origin in Q (may lie outside the image: the per-lane checks below) and its byte offset (signed)
        const int by = 2 * (y0 + tminy) + py, bx = 2 * (x0 + tminx) + px;
        const int tile_q = ((n * kHq + by) * kWq + bx) * kBc * 2;
        const uint32_t dstW = lds0 + buf * STAGE + P_BYTES + wave * 1024;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e * 4096 + wave * 1024 < W_BYTES) {      // wave-uniform
                const int iy = by + 2 * wj[e], ix = bx + 2 * wi[e];
                const bool ok = won[e] && (unsigned)iy < (unsigned)kHq && (unsigned)ix < (unsigned)kWq;
                wdma16(srdQ, ok ? (uint32_t)(tile_q + (int)lane_q[e]) : 0x80000000u, dstW + e * 4096);
            }
        }
    };

    f4 acc[NS][TA];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < TA; ++i) acc[s][i] = (f4){0.f, 0.f, 0.f, 0.f};

    const int wa = wave >> 1, wb = wave & 1;
    // transposing-read lane roles (wgrad.hip): group g = lane>>4 covers K rows 8g..8g+7 of a 32-row half; lane 4q+p of
    // the group addresses row q (and q+4), columns 4p..4p+3 of a 16-column block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int fr = ((q | ((g & 1) << 2)) << 1);
    const int rowoff = (8 * g + q) * 256 + (p & 1) * 8;
    // window: K row r = 32 ks + 8 g + q (+4) is tile pixel (4 ks + g, q (+4)); shift (sy, sx) adds (sy*WW + sx) pixels
    const int woff = (g * WW + q) * 64 + wb * 32 + p * 8;

    // One K-step: 2 K-halves x NS shifts "items" of TA MFMAs each.  With one wave per SIMD (the launch budget leaves the other
    // half of the CU to the main stream's kernels) nothing hides an LDS read behind another wave's MFMAs, so the reads are
    // pipelined by hand: the window fragment of item i + 2 (and, spread over the first half's last items, the P fragments of
    // the second half) is requested before the MFMAs of item i; sched_group_barrier pins that order.
    auto compute = [&](int buf) {
        const char* Ps = smem + buf * STAGE;
        const char* Ws = Ps + P_BYTES + woff;
        constexpr int NI = 2 * NS;
        auto read_a = [&](int ks, int ta) __attribute__((always_inline)) -> h8 {
            const int blk = wa * 4 + ta;
            const int ch = (2 * blk + (p >> 1)) ^ fr;
            const char* ad = Ps + ks * (32 * 256) + rowoff + ch * 16;
            union { s4v s[2]; h8 h; } u;
            u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
            u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 256));
            return u.h;
        };
        auto read_w = [&](int i) __attribute__((always_inline)) -> h8 {
            const int ks = i / NS, sh = i % NS, sy = sh / NSX, sx = sh % NSX;
            const char* ad = Ws + ((ks * 4 + sy) * WW + sx) * 64;
            union { s4v s[2]; h8 h; } u;
            u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
            u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 64));
            return u.h;
        };
        h8 af[2][TA], wf[3];
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) af[0][ta] = read_a(0, ta);
        wf[0] = read_w(0);
        wf[1] = read_w(1);
        __builtin_amdgcn_sched_barrier(0);
        static_for_g<0, NI>([&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            constexpr int ks = i / NS, sh = i % NS;
            constexpr bool rd_w = i + 2 < NI, rd_a = i >= NS - TA && i < NS;
            if constexpr (rd_w) wf[(i + 2) % 3] = read_w(i + 2);
            // the second half's P fragments ride on the last TA items of the first half
            if constexpr (rd_a) af[1][i - (NS - TA)] = read_a(1, i - (NS - TA));
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
                acc[sh][ta] = SLABS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i % 3], af[ks][ta], acc[sh][ta], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][ta], wf[i % 3], acc[sh][ta], 0, 0, 0);
            if constexpr (rd_w || rd_a) __builtin_amdgcn_sched_group_barrier(0x100, (rd_w ? 2 : 0) + (rd_a ? 2 : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TA, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    // 3-stage ring: the DMAs of step it+2 are issued at the top of step it, so a stage has two full steps to land.
    // Every wave issues the same number of DMA instructions per stage (4 + 2, wave 3: 4 + 1 when the window ends
    // inside its slice), so "all but the newest stage" is a constant s_waitcnt vmcnt(NW) per wave.
    if (loader) {
        const bool short_wave = 4096 + wave * 1024 >= W_BYTES;
        if (nsteps > 0) stage_load(0, t0);
        if (nsteps > 1) stage_load(1, t0 + 1);
        int nxt = 2;
        for (int it = 0; it < nsteps; ++it) {
            if (it + 1 < nsteps) {
                if (short_wave) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // raw barrier (a __syncthreads() would drain vmcnt): the current stage has landed for every loader wave, and
            // every compute wave is done reading stage `nxt` (it was the current stage of step it-1)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < nsteps) stage_load(nxt, t0 + it + 2);
            if (++nxt == 3) nxt = 0;
        }
        return;
    }
    int cur = 0;
    for (int it = 0; it < nsteps; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        if (++cur == 3) cur = 0;
    }

    // Per-split slabs (SLABS: slab_stride != 0; every split writes its own fp32 slab with plain stores -- each slab
    // element is written exactly once, the four planes cover all taps -- and fmri_unpack_grad sums the slabs): the window
    // fragment is the FIRST MFMA operand, D[i = b][j = a], so a lane owns row a (lane&15) and the four consecutive
    // channels b = (lane>>4)*4 .. +3: one 16-byte store per accumulator tile instead of four dword stores.
    // Atomic mode (many splits over a small matrix, one pre-zeroed matrix): D[i = a][j = b], a lane owns channel b
    // (lane&15) and rows (lane>>4)*4 .. +3, so that the 16 lanes of an atomic instruction fall into one 64-byte request
    // per row (the transposed ownership would spread them over 16 rows: 4 x the atomic requests, measured 1.6-2.7 x
    // slower on the 32-channel layers).  Column of shift (sy, sx) = tap*Bc + b.
    int kldo = a.ldo, kpad = a.pad, kTW = a.TW;
    float* slab = a.out + (int64_t)split * a.slab_stride;
    FMRI_KEEP(kldo); FMRI_KEEP(kpad); FMRI_KEEP(kTW); FMRI_KEEP(slab);
    const int bcol = b0 + wb * 16 + (SLABS ? (lane >> 4) * 4 : (lane & 15));
    if (bcol >= kBc) return;
    const int nfill = (SLABS && split + 1 == a.plane_pieces[py * 2 + px]) ? a.splits - split : 1;
#pragma unroll
    for (int sy = 0; sy < NSY; ++sy)
#pragma unroll
        for (int sx = 0; sx < NSX; ++sx) {
            const int ty = 2 * (tminy + sy) + py + kpad;
            const int tx = 2 * (tminx + sx) + px + kpad;
            const int col = (ty * kTW + tx) * kBc + bcol;
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                const f4 v = acc[sy * NSX + sx][ta];
                if constexpr (SLABS) {
                    const int arow = a0 + wa * 64 + ta * 16 + (lane & 15);
                    *(f4*)(slab + (int64_t)arow * kldo + col) = v;
                    // the plane's last piece also clears its columns in the slabs only other planes have pieces for
                    for (int sl = 1; sl < nfill; ++sl)
                        *(f4*)(slab + sl * a.slab_stride + (int64_t)arow * kldo + col) = (f4){0.f, 0.f, 0.f, 0.f};
                } else {
                    const int arow = a0 + wa * 64 + ta * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(slab + (int64_t)(arow + r) * kldo + col, v[r]);
                }
            }
        }
}

template <bool SLABS>
__global__ __launch_bounds__(512, 1) void wgrad_win_kernel(const WgradWinArgs a) {
    if (a.gate && *a.gate == 0) return;       // the sub-network is not trained in this step (fmri_wgrad_if)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware block -> work map.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own
    // L2); the blocks that read the same P tiles and Q windows at the same time -- all (column block, plane, row
    // block) combinations of one K split -- get consecutive LOGICAL ids, and logical ids are laid out so that a run of
    // gridDim.x/8 of them shares one XCD: every tile is then fetched from HBM once per XCD-resident group and the
    // other 15..31 readers hit in that XCD's L2.  (Speed only: any mapping is correct.)
    const int nb = gridDim.x;
    int logical = blockIdx.x;
    if ((nb & 7) == 0) logical = (logical & 7) * (nb >> 3) + (logical >> 3);
    const int nbt = a.Bc >> 5;                 // 32-channel column blocks
    const int groups = nbt * a.a_tiles;        // (row block, column block) pairs: the same K piece of the same plane adjacent
    int piece = logical / groups;
    int rem = logical - piece * groups;
    const int b_tile = rem % nbt;
    const int a_tile = rem / nbt;
    int plane = 0;
    while (plane < 3 && piece >= a.plane_pieces[plane]) { piece -= a.plane_pieces[plane]; ++plane; }
    const int split = piece;
    const int py = plane >> 1, px = plane & 1;
    const int nsy = a.nsy[py], nsx = a.nsx[px];
    if (nsy == 3 && nsx == 3) wgrad_win_body<3, 3, SLABS>(a, smem, py, px, a_tile, b_tile, split);
    else if (nsy == 3 && nsx == 2) wgrad_win_body<3, 2, SLABS>(a, smem, py, px, a_tile, b_tile, split);
    else if (nsy == 2 && nsx == 3) wgrad_win_body<2, 3, SLABS>(a, smem, py, px, a_tile, b_tile, split);
    else wgrad_win_body<2, 2, SLABS>(a, smem, py, px, a_tile, b_tile, split);
}

int wgrad_win_launch(const WgradWinArgs& a, int apad, hipStream_t st) {
    // 32-bit buffer offsets
    if ((int64_t)a.N * a.Yc * a.Xc * a.A * 2 >= 0x80000000LL || (int64_t)a.N * a.Hq * a.Wq * a.Bc * 2 >= 0x80000000LL)
        return E_UNSUPPORTED;
    dim3 grid((a.Bc / 32) * a.a_tiles * (a.plane_pieces[0] + a.plane_pieces[1] + a.plane_pieces[2] + a.plane_pieces[3]));
    const int lds = 3 * (64 * 256 + 7 * 1024);
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)wgrad_win_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return E_LAUNCH;
    if (hipFuncSetAttribute((const void*)wgrad_win_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return E_LAUNCH;
    if (a.slab_stride != 0) hipLaunchKernelGGL(wgrad_win_kernel<true>, grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL(wgrad_win_kernel<false>, grid, dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}


}  // namespace fmri
