// Window-resident implicit GEMM for unit-stride sampling with Ci % 64 == 0 (gfx950).
//
// Same contraction as igemm.hip,
//   out[n, y*os+oy0, x*os+ox0, co] = act(bias[co] + sum_{tap,ci} in[n, y+dy(tap), x+dx(tap), ci] * w[co][tap*Ci+ci]),
// for the geometries whose input is sampled with stride 1: the parity classes of a stride-2 transposed convolution
// (ConvTranspose2d forward, Conv2d-stride-2 data gradient) and stride-1 convolutions.
//
// igemm.hip streams a 128x64 A tile AND a BNx64 B tile per K-step from L2 into LDS and is bound by that stream
// (~53 GB/s per CU measured, 32 KB per 2.1 MFLOP step).  Here a 256-thread block owns 128 output pixels laid out as
// IPB image windows of PH x PW pixels and keeps their input WINDOW ((PH+TH-1) x (PW+TW-1) pixels x 64 channels)
// resident in LDS for all T taps of a 64-channel chunk: every tap reads its MFMA A fragments from the window at a
// shifted pixel offset, so only the BN x 64 weight tile streams per K-step (16 KB for BN = 128) plus 1/T of a window.
// The K loop is igemm.hip's: 2-stage weight ring, one barrier per K-step, two blocks per CU (<= 80 KB LDS each) so
// that one block's DMA waits overlap the other's MFMAs.  The next chunk's window is DMA'd in slices spread over the
// taps of the current chunk (second window buffer) so that no K-step ever waits for more than its own weight tile
// plus the one or two window slices issued behind it (a counted s_waitcnt vmcnt).
#include "kernels.h"

namespace fmri {

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_win_kernel(const WinArgs a) {
    constexpr int BM = 128;
    constexpr int W_BYTES = BN * 128;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int BROWS = BN / 32;
    constexpr int MAXE = 8;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WinClass& c = a.cls[blockIdx.z];
    int bx, by;
    xcd_tile(bx, by);
    if (bx >= c.ntiles) return;
    const int co0 = by * BN;

    char* const win0 = smem;
    char* const wbuf0 = smem + a.pbufs * a.win_bytes;

    // ---- tile -> (image group, window row, window col)
    const int tpi = c.tiles_y * c.tiles_x;
    const int grp = (int)fd_div((uint32_t)bx, c.fdTPI);
    const int trem = bx - grp * tpi;
    const int tyi = (int)fd_div((uint32_t)trem, c.fdTX);
    const int txi = trem - tyi * c.tiles_x;
    const int PW = 1 << c.pw_log2;
    const int y0 = tyi * c.PH, x0 = txi * PW;
    const int IHW = c.IH * c.IW;

    // ---- window DMA slices of this thread (slice e covers LDS bytes [e*4096, +4096) of a window buffer; 16-B unit
    // q = e*256 + tid holds channels 8*cc..8*cc+7 of window pixel q>>3, cc = (q&7) ^ (pixel&6)).
    // Why pixel&6: a ds_read_b128 is served in 16-lane groups that mix two k-chunks c and c^1 (fq 0/1 or 2/3), each from
    // 8 of 16 consecutive window pixels -- whatever the tap shift, those are all 8 residues mod 8.  Bank slot of (pixel,
    // c) = 8*(pixel&1) + (c ^ pixel&6): per pixel parity the four pixels give c^{0,2,4,6} and (c^1)^{0,2,4,6} = all 8
    // slots, so every group touches 16 distinct 16-B slots at ANY alignment (the former (pixel>>1)&7 was conflict-free
    // only for shifts that are multiples of 4: 1.75-2 x the LDS cycles over the taps of a 5x5 kernel).
    const int total_units = (c.IPB * IHW) << 3;
    int soff[MAXE];
    {
        // pinned (FMRI_KEEP): otherwise every unrolled iteration re-loads these inside its own `if`
        int pIHW = IHW, pIW = c.IW, pIPB = c.IPB, pN = a.N, pHi = a.Hi, pWi = a.Wi, pCi = a.Ci;
        int py0 = y0 + c.dymin, px0 = x0 + c.dxmin;
        uint32_t ihw_magic = c.fdIHW.magic, ihw_sh = c.fdIHW.sh, iw_magic = c.fdIW.magic, iw_sh = c.fdIW.sh;
        FMRI_KEEP(pIHW); FMRI_KEEP(pIW); FMRI_KEEP(pIPB); FMRI_KEEP(pN); FMRI_KEEP(pHi); FMRI_KEEP(pWi); FMRI_KEEP(pCi);
        FMRI_KEEP(py0); FMRI_KEEP(px0); FMRI_KEEP(ihw_magic); FMRI_KEEP(ihw_sh); FMRI_KEEP(iw_magic); FMRI_KEEP(iw_sh);
        const FastDiv fIHW{ihw_magic, ihw_sh, 0, 0}, fIW{iw_magic, iw_sh, 0, 0};
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            soff[e] = -1;
            const int q = e * 256 + tid;
            if (q < total_units) {
                const int pixel = q >> 3;
                const int ip = (int)fd_div((uint32_t)pixel, fIHW);
                const int rem = pixel - ip * pIHW;
                const int j = (int)fd_div((uint32_t)rem, fIW);
                const int i = rem - j * pIW;
                const int n = grp * pIPB + ip;
                const int iy = py0 + j, ix = px0 + i;
                if (n < pN && (unsigned)iy < (unsigned)pHi && (unsigned)ix < (unsigned)pWi)
                    soff[e] = ((n * pHi + iy) * pWi + ix) * pCi + (((q & 7) ^ (pixel & 6)) << 3);
            }
        }
    }
    // loop-invariant scalars of the K loop, pinned in SGPRs (see FMRI_KEEP)
    int nsl = c.nslice;                // slices per window (wave-uniform), <= MAXE
    int kwin = a.win_bytes, kKpad = c.Kpad, kCi = a.Ci, knch = a.nchunks, kpb = a.pbufs, kdstep = c.dstep, kTW = c.TW;
    int kIW = c.IW;
    const half_t* kin = a.in;
    const half_t* kzero = a.zero;
    FMRI_KEEP(nsl); FMRI_KEEP(kwin); FMRI_KEEP(kKpad); FMRI_KEEP(kCi); FMRI_KEEP(knch); FMRI_KEEP(kpb);
    FMRI_KEEP(kdstep); FMRI_KEEP(kTW); FMRI_KEEP(kIW); FMRI_KEEP(kin); FMRI_KEEP(kzero);
    auto load_slices = [&](int buf, int chunk, int lo, int hi) {
        char* dst = win0 + buf * kwin + wave * 1024;
        const half_t* base = kin + chunk * 64;
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e >= lo && e < hi) glds16_raw(soff[e] >= 0 ? base + soff[e] : kzero, dst + e * 4096);
    };

    // ---- weight tile DMA (rows = co, 64 k-values per step), XOR swizzled like igemm.hip
    const int trow = tid >> 3;
    const int clog = (tid & 7) ^ ((trow >> 1) & 7);
    const half_t* wrow = a.w + c.w_off + (int64_t)(co0 + trow) * c.Kpad + clog * 8;
    auto load_w = [&](int buf, int k0) {
        char* dst = wbuf0 + buf * W_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) glds16_raw(wrow + k0 + (int64_t)i * 32 * kKpad, dst + i * 4096);
    };

    f4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int tp_log2 = c.pw_log2 + c.ph_log2;
    // GEMM row rr of an image's tile -> tile column.  16-wide tiles: plain row-major.  8-wide tiles hold two tile rows
    // per 16-lane fragment; rotating row y by -y*IW makes the window pixels (mod 8) of the two rows complementary, which
    // is what keeps the ds_read_b128 of A conflict-free for every tap shift (see the swizzle note above).
    const int rotIW = c.pw_log2 == 3 ? c.IW : 0;
    auto tile_x = [&](int rr) { return (rr - (rr >> 3) * rotIW) & (PW - 1); };

    // window pixel (tap delta 0) of the TM output pixels this lane feeds
    int base_pix[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int ip = r >> tp_log2;
        const int rr = r & ((1 << tp_log2) - 1);
        base_pix[tm] = ip * IHW + (rr >> c.pw_log2) * c.IW + tile_x(rr);
    }

    auto compute = [&](int wb, int pb, int dlt) {
        const char* Ws = wbuf0 + wb * W_BYTES;
        const char* Ps = win0 + pb * kwin;
        h8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int pix = base_pix[tm] + dlt;
                af[ks][tm] = *(const h8*)(Ps + (pix << 7) + (((ks * 4 + fq) ^ (pix & 6)) << 4));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int row = wn * (BN / WN) + tn * 16 + frow;
                const int ph = (ks * 4 + fq) ^ ((row >> 1) & 7);
                bf[ks][tn] = *(const h8*)(Ws + row * 128 + ph * 16);
            }
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

    // ---- K loop: chunk-major over 64-channel chunks, taps inner.  Weight row offset of (chunk, tap) = tap*Ci + chunk*64.
    int T = c.T, spt = c.spt;        // spt = window slices issued per tap (second buffer only)
    FMRI_KEEP(T); FMRI_KEEP(spt);
    const int nsteps = knch * T;
    load_slices(0, 0, 0, nsl);
    load_w(0, 0);
    int chunk = 0, tap = 0, tx = 0;
    int dlt = (c.dy0 - c.dymin) * c.IW + (c.dx0 - c.dxmin);       // window delta of tap 0
    const int dlt0 = dlt;
    const int drow = kdstep * kIW - kTW * kdstep;              // delta correction at the end of a tap row
    int pend = 0;                                                  // slice DMAs issued behind the newest weight tile
    for (int s = 0; s < nsteps; ++s) {
        // weights of step s landed; at the first tap of a chunk the whole window must have landed too
        if (tap == 0 || pend == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        // raw barrier: __syncthreads() would add s_waitcnt vmcnt(0) and drain the slice DMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == T) { ntap = 0; ++nchunk; }
        if (s + 1 < nsteps) load_w((s + 1) & 1, ntap * kCi + nchunk * 64);
        pend = 0;
        if (kpb == 2 && chunk + 1 < knch) {
            const int lo = tap * spt;
            int hi = lo + spt;
            if (hi > nsl) hi = nsl;
            if (lo < hi) {
                load_slices((chunk + 1) & 1, chunk + 1, lo, hi);
                pend = hi - lo;
            }
        }
        compute(s & 1, kpb == 2 ? (chunk & 1) : 0, dlt);
        // advance the tap (scalar): next column, or first column of the next tap row
        dlt += kdstep;
        if (++tx == kTW) { tx = 0; dlt += drow; }
        tap = ntap;
        if (tap == 0) {
            chunk = nchunk;
            dlt = dlt0;
            tx = 0;
            if (kpb == 1 && chunk < knch) {
                // single window buffer: everyone is done with the old window, then reload (exposed once per chunk;
                // the other block resident on the CU keeps the MFMAs busy meanwhile)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                load_slices(0, chunk, 0, nsl);
            }
        }
    }

    // ---- epilogue: D[i = co][j = output pixel]
    const bool plain = a.bias == nullptr && a.act == ACT_NONE;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int ip = r >> tp_log2;
        const int rr = r & ((1 << tp_log2) - 1);
        const int n = grp * c.IPB + ip;
        const int y = y0 + (rr >> c.pw_log2), x = x0 + tile_x(rr);
        if (n >= a.N || y >= c.Yc || x >= c.Xc) continue;
        const int64_t opix = ((int64_t)n * a.Ho + (y * a.os + c.oy0)) * a.Wo + (x * a.os + c.ox0);
        half_t* orow = a.out + opix * a.CoStore;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int co = co0 + wn * (BN / WN) + tn * 16 + fq * 4;
            if (co >= a.CoStore) continue;
            const f4 v = acc[tn][tm];
            h4 hv;
            if (plain && co + 3 < a.Co) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) hv[rg] = (half_t)v[rg];
            } else {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float f = v[rg];
                    if (co + rg < a.Co) {
                        if (a.bias) f += a.bias[co + rg];
                        f = act_apply(f, a.act);
                    } else {
                        f = 0.f;
                    }
                    hv[rg] = (half_t)f;
                }
            }
            *(h4*)(orow + co) = hv;
        }
    }
}

template <int BN, int WM, int WN>
static int launch_win(const WinArgs& a, int max_tiles, int copad, int lds, hipStream_t st) {
    auto kern = igemm_win_kernel<BN, WM, WN>;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(max_tiles, copad / BN, a.ncls), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// Two window buffers when they fit beside the weight ring in 80 KB (two blocks per CU), else one.
int igemm_win_launch(WinArgs& a, int max_tiles, int bn_tile, int copad, hipStream_t st) {
    const int wring = 2 * bn_tile * 128;
    a.pbufs = (a.nchunks > 1 && 2 * a.win_bytes + wring <= 80 * 1024) ? 2 : 1;
    const int lds = a.pbufs * a.win_bytes + wring;
    if (lds > 80 * 1024) return E_UNSUPPORTED;
    switch (bn_tile) {
        case 128: return launch_win<128, 2, 2>(a, max_tiles, copad, lds, st);
        case 64: return launch_win<64, 2, 2>(a, max_tiles, copad, lds, st);
        case 32: return launch_win<32, 4, 1>(a, max_tiles, copad, lds, st);
        default: return E_UNSUPPORTED;
    }
}

}  // namespace fmri
