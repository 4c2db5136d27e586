// Weight gradient of the 5x5 stride-1 convolutions between a 32-channel and a 3(8)-channel map (gfx950):
//
//   dW[a][tap*8 + b] += sum_{m=(n,y,x)} P[m][a] * Q[n, y + s(ty), x + s(tx), b],   a < 32, b < 8, tap = ty*5 + tx,
//   s(t) = t - 2 (flip = 0: discriminator.conv.0, P = d pre-activation, Q = image) or 2 - t (flip = 1:
//   decoder.conv.3 with the roles exchanged, P = input activation, Q = d output)
//
// (models/vae_gan.py:118-121, 145-147; wgrad.hip's contract for A = 32, Bc = 8, k = 5.)  The output is 32 x 200
// numbers but the reduction runs over millions of pixels: wgrad.hip gathers Q once per tap (25x) through L2.  Here
// every WAVE works alone: it DMAs its half of the P rows of an 8x8-pixel tile (2 KB) and the (8+4)^2-pixel window of Q (2.3 KB)
// into its private LDS slice (double buffered -- no block barrier anywhere, only the wave's own vmcnt), reads both
// with transposing LDS reads (the MFMA K index is the pixel) and accumulates its 16 x 208 half of the result in registers
// (13 MFMA tiles; one N tile = two taps x 8 channels, lanes of the second tap read at their own window
// offset; the unused 26th tap multiplies by ones: output column 200 = sum_m P[m][a]).  P and Q are each read from HBM once.  Waves add their result into the fp32 output at the end.
#include "kernels.h"

namespace fmri {

// Round 3: each wave owns HALF of the 32 rows a (one 16-row MFMA tile) of its tile stream; the two waves of a pair walk
// the same tiles.  Round 2's wave held all 32 x 208 sums (104 accumulator + 200 other registers: ONE wave per SIMD) and
// sat 56 % of its cycles in LDS-issue stalls -- a single wave per SIMD cannot cover the latency of its transposing
// reads -- at 2.1 TB/s.  With 13 accumulator tiles per wave the kernel fits three waves per SIMD (12 per CU, each with
// its private 2-stage DMA ring: 1.5 x the bytes in flight), the other waves' MFMAs and DMA issue cover a wave's LDS
// latency, and P is still read from HBM once (a wave DMAs only its 32-byte half of every P row; the 2.3 KB window is
// fetched by both waves of a pair, the second time from L2).
__global__ __launch_bounds__(256, 3) void wgrad_narrow_kernel(const WgradNarrowArgs a) {
    if (a.gate && *a.gate == 0) return;       // the sub-network is not trained in this step (fmri_wgrad_if)
    constexpr int WW = 12, WPIX = WW * WW;          // Q window of an 8x8 tile (5x5 taps)
    constexpr int P_BYTES = 64 * 32;                // 64 pixels x 16 channels (this wave's half of the 32)
    constexpr int W_BYTES = 4 * 1024;               // 144 pixels x 16 B; bytes 2304.. hold ones (the 26th tap reads there)
    constexpr int SLICE = P_BYTES + W_BYTES;        // per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const mine = smem + wave * (2 * SLICE);
    const int gw = blockIdx.x * 4 + wave;           // global wave id
    const int half = gw & 1;                        // rows a = 16 half .. 16 half + 15
    const int pair = gw >> 1, npairs = gridDim.x * 2;

    // the tail of both window buffers holds ONES (fp16 1.0): the lanes of the non-existent 26th tap read there, so
    // columns 200 .. 207 of the result are the column sums of P (flip = 0: the bias gradient of the layer)
    for (int b = 0; b < 2; ++b)
        for (int o = WPIX * 16 + lane * 4; o < W_BYTES; o += 256) *(int*)(mine + b * SLICE + P_BYTES + o) = 0x3C003C00;

    const int tpi = a.tiles_y * a.tiles_x;
    auto stage_load = [&](int buf, int t) __attribute__((always_inline)) {
        const int n = t / tpi;
        const int r = t - n * tpi;
        const int tyi = r / a.tiles_x, txi = r - tyi * a.tiles_x;
        const int y0 = tyi * 8, x0 = txi * 8;
        char* dstP = mine + buf * SLICE;
        // P: DMA instruction i covers tile pixels 32 i .. 32 i + 31 (lane = pixel * 2 + 16-B chunk of the 32-B half row)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = 32 * i + (lane >> 1);
            const int y = y0 + (pix >> 3), x = x0 + (pix & 7);
            const bool ok = y < a.H && x < a.W;
            const half_t* ps = ok ? a.P + ((int64_t)(n * a.H + y) * a.W + x) * 32 + half * 16 + (lane & 1) * 8 : a.zero;
            glds16_raw(ps, dstP + i * 1024);
        }
        // Q window: unit u = 64 i + lane -> window pixel u (16 B = 8 channels)
        char* dstW = dstP + P_BYTES;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = 64 * i + lane;
            const int j = u / WW, c = u - j * WW;
            const int iy = y0 - 2 + j, ix = x0 - 2 + c;
            const bool ok = u < WPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const half_t* qs = ok ? a.Q + ((int64_t)(n * a.H + iy) * a.W + ix) * 8 : a.zero;
            if (i < 2 || lane < WPIX - 128) glds16_raw(qs, dstW + i * 1024);
        }
    };

    // transposing-read lane roles (wgrad.hip): group g = lane>>4 covers K rows 8g..8g+7 of a 32-row half; lane 4q+p of
    // the group addresses row q (and q+4), columns 4p..4p+3 of a 16-column block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int poff = (8 * g + q) * 32 + p * 8;                       // P: 32-B pixel rows (16 channels)
    // window offset of this lane for N tile j: its tap is 2j + (p >> 1)
    int woff[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        const int tap = 2 * j + (p >> 1);
        const int ty = tap / 5, tx = tap - ty * 5;
        const int sy = a.flip ? 4 - ty : ty, sx = a.flip ? 4 - tx : tx;    // window row/col offset of the tap
        woff[j] = tap < 25 ? ((g + sy) * WW + q + sx) * 16 + (p & 1) * 8 : WPIX * 16;
    }

    f4 acc[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};

    // wave-private 2-stage ring: 2 + 3 (lanes < 16 of the third window instruction only) DMA instructions per stage
    int t = pair;
    if (t < a.ntiles) stage_load(0, t);
    int cur = 0;
#pragma unroll 1
    for (; t < a.ntiles; t += npairs) {
        const int tn = t + npairs;
        if (tn < a.ntiles) {
            stage_load(cur ^ 1, tn);
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");        // everything but the stage just issued
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* Ps = mine + cur * SLICE + poff;
        const char* Ws = mine + cur * SLICE + P_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 af;
            {
                const char* ad = Ps + ks * (32 * 32);
                union { s4v s[2]; h8 h; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 32));
                af = u.h;
            }
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                // K row 32 ks + 8 g + q (+4) is tile pixel (4 ks + g, q (+4)): + 4 ks window rows, + 4 pixels
                const char* ad = Ws + woff[j] + ks * (4 * WW * 16);
                union { s4v s[2]; h8 h; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ad + 4 * 16));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, u.h, acc[j], 0, 0, 0);
            }
        }
        // the LDS reads of this stage are complete (their values fed the MFMAs) before the next iteration's DMA
        // overwrites it: the DMA is issued after these instructions in program order and LDS ops of a wave retire
        // in order, but make the dependency explicit
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur ^= 1;
    }

    // ---- block reduction: waves 2, 3 hand their sums to waves 0, 1 (same half of the rows a: half = wave & 1) through
    // LDS with plain stores, and those add the pair to the fp32 output with global atomics straight from registers.
    // (Rounds 1-2 reduced the block with ds_add_f32 into a [32][201] LDS matrix first: LDS float atomics retire about one
    // LANE per three cycles -- 52 wave-instructions per wave took 48 us of the kernel's 120, timed with the loop
    // emptied: DESIGN section 6.)  D[i = a][j = column of the N tile]: lane owns column (lane & 15) = tap
    // parity * 8 + b and rows (lane >> 4) * 4 .. + 3 of its half: the 16 lanes of a row are 64 contiguous bytes.
    __syncthreads();                                 // every wave is done with its staging slices
    f4* xch = (f4*)smem;                             // [2][13][64] f4 = 26 KB
    if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < 13; ++j) xch[((wave - 2) * 13 + j) * 64 + lane] = acc[j];
    }
    __syncthreads();
    if (wave < 2) {
        const int col16 = lane & 15;
        float* slab = a.out + (int64_t)(blockIdx.x % a.nslabs) * a.slab_stride;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const f4 o = xch[(wave * 13 + j) * 64 + lane];
            const int tap = 2 * j + (col16 >> 3);
            if (tap > 25 || (tap == 25 && (col16 & 7))) continue;
            const int col = tap * 8 + (col16 & 7);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(slab + (int64_t)(half * 16 + (lane >> 4) * 4 + r) * a.ldo + col, acc[j][r] + o[r]);
        }
    }
}

int wgrad_narrow_launch(const WgradNarrowArgs& a, int nblocks, hipStream_t st) {
    const int lds = 4 * 2 * (64 * 32 + 4 * 1024);       // 48 KB: three blocks per CU
    hipLaunchKernelGGL(wgrad_narrow_kernel, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? OK : E_LAUNCH;
}

}  // namespace fmri
