// Patch-resident implicit GEMM for unit-stride sampling (gfx950).
//
// Same contraction as igemm.hip,
//   out[n, y*os+oy0, x*os+ox0, co] = act(bias[co] + sum_{tap,ci} in[n, y+dy(tap), x+dx(tap), ci] * w[co][tap*Ci+ci]),
// for the geometries whose input is sampled with stride 1: the 4 parity classes of a stride-2 transposed
// convolution (ConvTranspose2d forward, Conv2d-stride-2 data gradient) and stride-1 convolutions /
// their data gradients.
//
// A 512-thread block (8 waves, 2 per SIMD) owns 256 output pixels arranged as IPB image tiles of PH x PW
// pixels and keeps the whole input PATCH ((PH+TH-1) x (PW+TW-1) pixels x 64 channels) resident in LDS: it is
// DMA'd once per 64-channel chunk and every tap reads its MFMA A-fragments from it at a shifted pixel offset.
// Only the weight tile (BN x 64) is streamed per tap, through a ring of WS (3-4) LDS stages that keeps
// WS-1 K-steps of LDS-DMA in flight: a 2-stage ring exposes the ~1 us L2->LDS DMA round trip on every
// ~0.45 us K-step, which is what bounded igemm.hip at ~30 % of the MFMA peak.  Every wave counts its own
// outstanding DMA instructions and waits with a counted s_waitcnt vmcnt(N) -- never vmcnt(0) in the loop.
#include "kernels.h"

namespace fmri {

__device__ __forceinline__ void wait_vmcnt(int n) {
    // n is wave-uniform; the immediate must be a literal
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}

template <int BN, int WM, int WN>
__global__ __launch_bounds__(512) void igemm_patch_kernel(const PatchArgs a) {
    constexpr int BM = 256;
    constexpr int W_BYTES = BN * 128;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int MAXE = 7;
    static_assert(WM * WN == 8, "8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PatchClass& c = a.cls[blockIdx.z];
    if ((int)blockIdx.x >= c.ntiles) return;
    const int co0 = blockIdx.y * BN;
    const int WS = a.wstages;           // weight ring stages; WS - 1 K-steps of DMA stay in flight

    char* const patch0 = smem;
    char* const wbuf0 = smem + a.pbufs * a.patch_bytes;
    int* const tapd = (int*)(wbuf0 + WS * W_BYTES);

    // ---- tile -> (image group, tile row, tile col)
    const int tpi = c.tiles_y * c.tiles_x;
    const int grp = blockIdx.x / tpi;
    const int trem = blockIdx.x - grp * tpi;
    const int tyi = trem / c.tiles_x;
    const int txi = trem - tyi * c.tiles_x;
    const int PW = 1 << c.pw_log2;
    const int y0 = tyi * c.PH, x0 = txi * PW;
    const int IHW = c.IH * c.IW;
    const int cpp_log2 = a.cpp_log2;
    const int kshift = cpp_log2 == 3 ? 1 : 2;
    const int kmask = cpp_log2 == 3 ? 7 : (cpp_log2 == 2 ? 3 : 0);
    const int pix_shift = 4 + cpp_log2;

    // ---- tap -> patch pixel delta table
    if (tid < 32) {
        const int tt = tid < c.T ? tid : c.T - 1;
        const int ty = tt / c.TW, tx = tt - ty * c.TW;
        tapd[tid] = (c.dy0 + ty * c.dstep - c.dymin) * c.IW + (c.dx0 + tx * c.dstep - c.dxmin);
    }

    // ---- patch DMA entries of this thread (pixel/channel-slot -> source offset), fixed for all chunks.
    // Entry e of wave w writes LDS bytes [e*8192 + w*1024, +1024); entries past the patch buffer are skipped
    // (a wave-uniform decision, so every wave knows how many DMA instructions it issues: nmine).
    const int total_chunks = (c.IPB * IHW) << cpp_log2;
    int soff[MAXE];
    int nmine = 0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        soff[e] = -1;
        if ((e * 512 + wave * 64) * 16 < a.patch_bytes) {
            ++nmine;
            const int q = e * 512 + tid;
            const int pixel = q >> cpp_log2;
            const int phys = q & ((1 << cpp_log2) - 1);
            if (q < total_chunks) {
                const int ip = pixel / IHW;
                const int rem = pixel - ip * IHW;
                const int j = rem / c.IW;
                const int i = rem - j * c.IW;
                const int n = grp * c.IPB + ip;
                const int iy = y0 + c.dymin + j, ix = x0 + c.dxmin + i;
                if (n < a.N && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) {
                    const int cc = phys ^ ((pixel >> kshift) & kmask);
                    soff[e] = ((n * a.Hi + iy) * a.Wi + ix) * a.Ci + cc * 8;
                }
            }
        }
    }
    nmine = __builtin_amdgcn_readfirstlane(nmine);

    auto load_patch = [&](int buf, int chunk) {
        char* dst = patch0 + buf * a.patch_bytes + wave * 1024;
        const half_t* base = a.in + chunk * 64;
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e < nmine) glds16(soff[e] >= 0 ? base + soff[e] : a.zero, dst + e * 8192);
    };

    // ---- weight tile DMA (rows = co, 64 k-values per step), XOR swizzled like igemm.hip
    const int trow = tid >> 3;
    const int clog = (tid & 7) ^ ((trow >> 1) & 7);
    const half_t* wrow = a.w + c.w_off + (int64_t)(co0 + trow) * c.Kpad + clog * 8;
    const int wmine = BN >= 64 ? BN / 64 : (wave < BN / 8 ? 1 : 0);   // DMA instructions per wave per stage
    auto load_w = [&](int buf, int k0) {
        char* dst = wbuf0 + buf * W_BYTES + wave * 1024;
        if (BN >= 64) {
#pragma unroll
            for (int i = 0; i < (BN >= 64 ? BN / 64 : 1); ++i)
                glds16(wrow + k0 + (int64_t)i * 64 * c.Kpad, dst + i * 8192);
        } else {
            if (wave < BN / 8) glds16(wrow + k0, dst);
        }
    };

    f4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int tile_px = c.PH << c.pw_log2;

    // patch pixel (tap delta 0) of the TM output rows this lane feeds
    int base_pix[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int ip = r / tile_px;
        const int rr = r - ip * tile_px;
        base_pix[tm] = ip * IHW + (rr >> c.pw_log2) * c.IW + (rr & (PW - 1));
    }

    const int nsteps = c.Kpad >> 6;
    const int T = c.T;

    auto compute = [&](int wb, int pb, int it, int tapbase) {
        const char* Ws = wbuf0 + wb * W_BYTES;
        const char* Ps = patch0 + pb * a.patch_bytes;
        h8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sl = ks * 4 + fq;
            int tap, cc;
            if (cpp_log2 == 3) { tap = tapbase; cc = sl; }
            else if (cpp_log2 == 2) { tap = it * 2 + (sl >> 2); cc = sl & 3; }
            else { tap = it * 8 + sl; cc = 0; }
            const int dlt = tapd[tap < 31 ? tap : 31];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int pix = base_pix[tm] + dlt;
                af[ks][tm] = *(const h8*)(Ps + (pix << pix_shift) + ((cc ^ ((pix >> kshift) & kmask)) << 4));
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

    // k offset of K-step `s` in the packed weight rows: chunk-major over 64-channel chunks, taps inner
    auto k_of = [&](int s) {
        if (cpp_log2 != 3) return s * 64;
        const int ch = s / T;
        return (s - ch * T) * a.Ci + ch * 64;
    };

    // ---- prologue: patch of chunk 0, then the first WS-1 weight stages.  `issued` counts this wave's DMA
    // instructions; q0..q2 hold the value of `issued` right after the weights of steps it, it+1, it+2.
    int issued = 0;
    load_patch(0, 0);
    issued += nmine;
    const int D = WS - 1;
    int q0 = 0, q1 = 0, q2 = 0;
    for (int s = 0; s < D; ++s) {
        if (s < nsteps) { load_w(s, k_of(s)); issued += wmine; }
        if (s == 0) q0 = issued; else if (s == 1) q1 = issued; else q2 = issued;
    }
    int chunk = 0, tap = 0, stage = 0, nstage = D;   // stage = it % WS, nstage = (it + D) % WS
    for (int it = 0; it < nsteps; ++it) {
        wait_vmcnt(issued - q0);     // weights of this step (and everything older, incl. its patch) have landed
        // raw barrier: __syncthreads() would add s_waitcnt vmcnt(0) and drain the DMA ring
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (cpp_log2 == 3 && tap == 0 && chunk + 1 < a.nchunks) {
            load_patch((chunk + 1) & 1, chunk + 1);
            issued += nmine;
        }
        if (it + D < nsteps) { load_w(nstage, k_of(it + D)); issued += wmine; }
        q0 = q1;
        if (D == 3) { q1 = q2; q2 = issued; } else { q1 = issued; }
        compute(stage, a.pbufs == 2 ? (chunk & 1) : 0, it, tap);
        if (++stage == WS) stage = 0;
        if (++nstage == WS) nstage = 0;
        if (cpp_log2 == 3) {
            if (++tap == T) { tap = 0; ++chunk; }
        }
    }

    // ---- epilogue: D[i = co][j = output pixel]
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * (BM / WM) + tm * 16 + frow;
        const int ip = r / tile_px;
        const int rr = r - ip * tile_px;
        const int n = grp * c.IPB + ip;
        const int y = y0 + (rr >> c.pw_log2), x = x0 + (rr & (PW - 1));
        if (n >= a.N || y >= c.Yc || x >= c.Xc) continue;
        const int64_t opix = ((int64_t)n * a.Ho + (y * a.os + c.oy0)) * a.Wo + (x * a.os + c.ox0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int co = co0 + wn * (BN / WN) + tn * 16 + fq * 4;
            if (co >= a.CoStore) continue;
            const f4 v = acc[tn][tm];
            h4 hv;
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
            *(h4*)(a.out + opix * a.CoStore + co) = hv;
        }
    }
}

template <int BN, int WM, int WN>
static int launch_patch(const PatchArgs& a, int max_tiles, int copad, int lds, hipStream_t st) {
    auto kern = igemm_patch_kernel<BN, WM, WN>;
    // raising the dynamic-LDS limit is idempotent; every call sets it (no library-global state)
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(max_tiles, copad / BN, a.ncls), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

// Chooses the deepest weight ring (4, else 3 stages) that fits 160 KB of LDS with the patch buffers.
int igemm_patch_launch(PatchArgs& a, int max_tiles, int bn_tile, int copad, hipStream_t st) {
    int lds = 0;
    for (a.wstages = 4; a.wstages >= 3; --a.wstages) {
        lds = a.pbufs * a.patch_bytes + a.wstages * bn_tile * 128 + 128;
        if (lds <= 160 * 1024) break;
    }
    if (a.wstages < 3) return E_UNSUPPORTED;
    switch (bn_tile) {
        case 128: return launch_patch<128, 4, 2>(a, max_tiles, copad, lds, st);
        case 64: return launch_patch<64, 4, 2>(a, max_tiles, copad, lds, st);
        case 32: return launch_patch<32, 8, 1>(a, max_tiles, copad, lds, st);
        default: return E_UNSUPPORTED;
    }
}

}  // namespace fmri
