// Weight-gradient implicit GEMM on MFMA for gfx950.
//
//   dW[a][tap*Bc + b] (+)= sum_{m=(n,y,x)} P[m][a] * Q[n, y*s+dy(tap), x*s+dx(tap), b]
//
// Replaces the autograd weight-gradients of nn.Conv2d / nn.ConvTranspose2d / nn.Linear on the hot path
// (reference models/vae_gan.py:18-20, :46-53, :79, :107, :156; driven by loss.backward() in
// train/train_vgan_stage1.py:412-430):
//   conv   : P = dY (a = c_out), Q = X  (b = c_in)
//   deconv : P = X  (a = c_in),  Q = dY (b = c_out)      (same gather, roles swapped)
//   dense  : T = 1, Yc = Xc = 1: dW[n_out][k] = sum_batch dY[m][n_out] * X[m][k]
//
// The reduction index m is the *slow* index of both operands in memory (NHWC rows), i.e. a "TN" GEMM.
// Tiles are DMA'd to LDS as [m][channels] rows (256 B) and the MFMA fragments (8 consecutive m per
// lane) are produced by ds_read_b64_tr_b16 transposing reads -- no transposed copies in HBM.
// The LDS image is XOR-swizzled (on the DMA source side) so that every transposing read is
// bank-conflict free.  Split-K over m with fp32 atomics (atomic == 1), per-split slabs written with plain stores
// (atomic == 2: the deterministic form, summed by fmri_unpack_grad) or plain stores when splits == 1.
#include "kernels.h"

namespace fmri {

template <int BA, int WA, int WB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    if (a.gate && *a.gate == 0) return;       // the sub-network is not trained in this step (fmri_wgrad_if)
    constexpr int TILE = 64 * 256;   // 64 m-rows x 128 halfs
    constexpr int STAGE = 2 * TILE;
    constexpr int WAVE_A = BA / WA;
    constexpr int WAVE_B = 128 / WB;
    constexpr int TA = WAVE_A / 16;
    constexpr int TB = WAVE_B / 16;
    static_assert(WA * WB == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a0 = blockIdx.y * BA;
    const int j0 = blockIdx.x * 16;   // first 8-wide column chunk of this tile
    const int split = blockIdx.z;

    const int trow = tid >> 4;
    const int cphys = tid & 15;
    const int fsw = (((trow & 3) | (((trow >> 3) & 1) << 2)) << 1);
    const int clog = cphys ^ fsw;

    // ---- P source (linear rows)
    const bool p_on = (clog * 8 < BA) && (a0 + clog * 8 < a.A);
    const half_t* pbase = a.P + a0 + clog * 8;

    // ---- Q source: this thread always loads the same (tap, 8 channels) column chunk
    const int j = j0 + clog;
    const bool q_on = j < a.ncol_chunks;
    const int tap = (int)fd_div((uint32_t)(q_on ? j : 0), a.fdBc8);
    const int b0 = ((q_on ? j : 0) - tap * (a.Bc >> 3)) * 8;
    const int ty = (int)fd_div((uint32_t)tap, a.fdTW);
    const int tx = tap - ty * a.TW;
    const int dy = a.dy0 + ty * a.dstep;
    const int dx = a.dx0 + tx * a.dstep;
    const int YX = a.Yc * a.Xc;

    const int mb = split * a.steps_per_split * 64;
    int me = mb + a.steps_per_split * 64;
    if (me > a.M) me = a.M;
    const int nsteps = (me - mb + 63) / 64;

    auto stage_load = [&](int buf, int step) {
        char* dstP = smem + buf * STAGE + wave * (4 * 256);
        char* dstQ = dstP + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + step * 64 + trow + 16 * i;
            const bool mv = m < me;
            const half_t* ps = (mv && p_on) ? pbase + (int64_t)m * a.A : a.zero;
            glds16(ps, dstP + i * (16 * 256));
            const uint32_t mm = mv ? (uint32_t)m : 0u;
            const uint32_t n = fd_div(mm, a.fdYX);
            const uint32_t rem = mm - n * (uint32_t)YX;
            const uint32_t y = fd_div(rem, a.fdX);
            const uint32_t x = rem - y * (uint32_t)a.Xc;
            const int iy = (int)y * a.s + dy;
            const int ix = (int)x * a.s + dx;
            const bool ok = mv && q_on && (unsigned)iy < (unsigned)a.Hq && (unsigned)ix < (unsigned)a.Wq;
            const half_t* qs = ok ? a.Q + (((int64_t)n * a.Hq + iy) * a.Wq + ix) * a.Bc + b0 : a.zero;
            glds16(qs, dstQ + i * (16 * 256));
        }
    };

    f4 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int k = 0; k < TB; ++k) acc[i][k] = (f4){0.f, 0.f, 0.f, 0.f};

    const int wa = wave / WB, wb = wave % WB;
    // transposing-read lane roles: group g = lane>>4 covers m rows 8g..8g+7 of a 32-row sub-step;
    // lane 4q+p of the group addresses row q, columns 4p..4p+3 of the 16-column block.
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int fr = ((q | ((g & 1) << 2)) << 1);          // swizzle key of rows 8g+q and 8g+4+q
    const int rowoff = (8 * g + q) * 256 + (p & 1) * 8;

    auto compute = [&](int buf) {
        const char* Ps = smem + buf * STAGE;
        const char* Qs = Ps + TILE;
        // second-half transposing reads are interleaved with the first half's MFMAs (see igemm.hip)
        h8 af[2][TA], bf[2][TB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                const int blk = (wa * WAVE_A + ta * 16) >> 4;
                const int ch = (2 * blk + (p >> 1)) ^ fr;
                const char* ad = Ps + ks * (32 * 256) + rowoff + ch * 16;
                union { s4v s[2]; h8 h; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 256));
                af[ks][ta] = u.h;
            }
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int blk = (wb * WAVE_B + tb * 16) >> 4;
                const int ch = (2 * blk + (p >> 1)) ^ fr;
                const char* ad = Qs + ks * (32 * 256) + rowoff + ch * 16;
                union { s4v s[2]; h8 h; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 256));
                bf[ks][tb] = u.h;
            }
            if (ks == 0) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][ta], bf[ks][tb], acc[ta][tb], 0, 0, 0);
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < TA + TB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS (transposing) reads
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TA * TB - (TA + TB), 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (nsteps > 0) stage_load(0, 0);
    for (int it = 0; it < nsteps; ++it) {
        const int cur = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < nsteps) stage_load(cur ^ 1, it + 1);
        compute(cur);
    }
    if (nsteps <= 0 && a.atomic == 1) return;

    // D[i = a][j = column]: lane owns column (lane&15), rows (lane>>4)*4 .. +3
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int col = blockIdx.x * 128 + wb * WAVE_B + tb * 16 + (lane & 15);
            const int arow = a0 + wa * WAVE_A + ta * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // atomic == 2: every K split stores its own slab (plain stores; fmri_unpack_grad sums them in order)
                float* o = a.out + (a.atomic == 2 ? (int64_t)blockIdx.z * a.slab_stride : 0) +
                           (int64_t)(arow + r) * a.ldo + col;
                if (a.atomic == 1) atomicAdd(o, acc[ta][tb][r]);
                else *o = acc[ta][tb][r];
            }
        }
}

int wgrad_launch(const WgradArgs& a, int apad, int ba_tile, hipStream_t st) {
    dim3 grid(a.ldo / 128, apad / ba_tile, a.splits);
    const int lds = 2 * 2 * 64 * 256;
    switch (ba_tile) {
        case 128: hipLaunchKernelGGL((wgrad_kernel<128, 2, 2>), grid, dim3(256), lds, st, a); break;
        case 64: hipLaunchKernelGGL((wgrad_kernel<64, 1, 4>), grid, dim3(256), lds, st, a); break;
        case 32: hipLaunchKernelGGL((wgrad_kernel<32, 1, 4>), grid, dim3(256), lds, st, a); break;
        default: return E_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
