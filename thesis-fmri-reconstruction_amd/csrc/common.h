// Shared device/host definitions for the gfx950 (CDNA4) kernels of the fMRI VAE/GAN engine.
// Everything here is written for MI355X only: 64-wide wavefronts, MFMA 16x16x32 f16,
// LDS-DMA (global_load_lds) staging, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmri {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

// gfx950 store-data hazard.  A VMEM store of more than 64 bits reads its data VGPRs AFTER it has issued; a VALU write of
// one of them in the next two issue slots can land first, and the store then carries the new value.  hipcc (ROCm 7.2) pads
// such pairs with wait states -- except when the store has an SGPR offset, which is how the loader waves of igemm_c5w /
// igemm_tc5w address a tile's rows.  Round 3 met the back-to-back packed-fp32 form (wrong second dword, every launch);
// round 4's deterministic mode exposed the single-register form: `buffer_store_dwordx4 v[24:27], v28, s[20:23], s15 offen`
// directly followed by `v_cndmask_b32 v24, ...` (the next store's offset computed into the first data register) stored a
// wrong FIRST dword in about one launch in a few hundred, and only while a kernel on another stream kept the memory system
// busy (tools/probes/store_hazard_stress.py).  Every such store is therefore followed by two wait states that the
// scheduler may not move: FMRI_STORE_FENCE().  tools/scan_store_hazard.py (a build step, fmri_hip/build.py) counts the wait
// states behind every wide VMEM store of the library and fails the build on fewer than two.
// (-DFMRI_NO_STORE_FENCE builds the library without it: tools/probes/build_variant.sh, to show that the tests catch it.)
#ifndef FMRI_NO_STORE_FENCE
#define FMRI_STORE_FENCE()                        \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_nop 1" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#else
#define FMRI_STORE_FENCE() do { } while (0)
#endif

// error codes of the C ABI (include/fmri_hip.h)
enum Err { OK = 0, E_BADARG = -1, E_UNSUPPORTED = -2, E_LAUNCH = -3, E_WORKSPACE = -4 };

// ---------------------------------------------------------------------------------------------
// exact unsigned division by a runtime constant (Granlund-Montgomery round-up method):
//   q = (t + ((n - t) >> 1)) >> sh,  t = umulhi(n, magic)      for d >= 2
// valid for all 32-bit n.  d == 1 is encoded as magic = 0, sh = 0 with the identity path.
// ---------------------------------------------------------------------------------------------
struct FastDiv {
    uint32_t magic;
    uint32_t sh;   // shift - 1 ; 0xffffffff marks d == 1
    uint32_t d;
    uint32_t pad;
};

inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.pad = 0;
    if (d <= 1) { f.magic = 0; f.sh = 0xffffffffu; f.d = 1; return f; }
    uint32_t s = 0;
    while ((1ull << s) < (uint64_t)d) ++s;            // s = ceil(log2 d), 1..32
    uint64_t m = (((1ull << s) - d) << 32) / d + 1;   // < 2^32
    f.magic = (uint32_t)m;
    f.sh = s - 1;
    return f;
}

__host__ __device__ inline uint32_t fd_div(uint32_t n, const FastDiv& f) {
    if (f.sh == 0xffffffffu) return n;
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t t = __umulhi(n, f.magic);
#else
    uint32_t t = (uint32_t)(((uint64_t)n * f.magic) >> 32);
#endif
    return (t + ((n - t) >> 1)) >> f.sh;
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM geometry (see igemm.hip).  One "class" = one dense sub-problem:
//   out[n, y*os+oy0, x*os+ox0, co] = sum_{tap<T, ci} in[n, y*s+dy(tap), x*s+dx(tap), ci] * w[co][tap*Ci+ci]
// with tap = ty*TW + tx, dy = dy0 + ty*dstep, dx = dx0 + tx*dstep.  A plain convolution has one
// class; a stride-2 transposed convolution has four (output parity classes).
// ---------------------------------------------------------------------------------------------
// BatchNorm statistics emitted by a contraction's epilogue: per output channel sum x and sum x^2 of the STORED (fp16)
// values over the valid output pixels of one block, written (plain stores: deterministic) to the block's own row of
//   part[group][rows_cap][2][C]
// group = image / group_n (decoder: several BatchNorm batches in one launch; group_n = 0: one group).  A group's rows
// are dense: [0, P) with P = sum over classes of tpg[class]; fmri_bn_fold_finalize folds them.
struct StatEpi {
    float* part;           // null: no statistics
    int32_t rows_cap;      // rows allocated per group
    int32_t C;             // channels per row half (= CoStore of the launch)
    int32_t group_n;       // images per statistics group, 0 = all
    int32_t tpg[4];        // row tiles per group of class c (igemm_tc5: [0] = tiles per group, all classes in one tile)
};

// BatchNorm + ReLU BACKWARD statistics in the epilogue of the data-gradient contraction that produces the cotangent dy
// of a BatchNorm output (dgrad of the next layer):  with the saved forward tensor x (same geometry as the contraction's
// output), xhat = (x - mean)*rstd and on = !relu || xhat*gamma + beta > 0, the epilogue stores g = on ? dy : 0 and
// emits the StatEpi rows with (sum g, sum g*xhat) instead of (sum x, sum x^2).  Statistics group i (image / group_n:
// cotangent streams or decoder calls stacked along the batch) reads x from image x_img0[i] on and uses the batch
// statistics mean[i] / rstd[i] of the forward call it belongs to.
// fmri_epilogue.aff_*: out = relu?(acc * scale[co] + shift[co]) (eval-mode BatchNorm folded into the producer); null scale: off
struct AffEpi {
    const float* scale;
    const float* shift;
    int32_t relu, pad0;
};

struct BnBwdEpi {
    const half_t* x;       // null: forward statistics (StatEpi only)
    const float* gamma;
    const float* beta;
    const float* mean[4];
    const float* rstd[4];
    int32_t x_img0[4];
    int32_t relu, pad0;
};

struct IgemmClass {
    int32_t Yc, Xc;        // rows / cols of the output sub-grid per image
    int32_t oy0, ox0;      // output offset of the class
    int32_t T, TW;         // number of taps, taps per tap-row
    int32_t dy0, dx0, dstep;
    int32_t M;             // N * Yc * Xc
    int32_t Kpad;          // padded reduction length (multiple of 64)
    int32_t ksteps;        // Kpad / 64
    int64_t w_off;         // element offset of this class's packed weight matrix
    FastDiv fdX, fdYX, fdTW;
};

struct IgemmArgs {
    const half_t* in;
    const half_t* w;
    void* out;
    const float* bias;       // may be null
    const half_t* zero;      // >= 16 bytes of zeros (gather target for padding)
    int32_t N, Hi, Wi, Ci;
    int32_t Ho, Wo, CoStore, Co;
    int32_t s, os;
    int32_t act;
    int32_t ncls, splits;
    int64_t slab_stride;     // elements between split-K slabs (fp32 output only)
    FastDiv fdCi;
    FastDiv fdCpt;           // divide by Ci/64 (K-steps per tap) when Ci % 64 == 0
    StatEpi st;              // BatchNorm statistics epilogue (fp16 output only)
    BnBwdEpi bb;             // ... of a BatchNorm backward (with st)
    IgemmClass cls[4];
};

// all four parity classes of a k5 s2 p2 transposed convolution per block (igemm_tc5.hip)
struct Tc5Class {
    int32_t Yc, Xc;        // class output grid: class (cy, cx) writes output pixel (2y + cy, 2x + cx)
    int32_t Kpad, pad0;
    int64_t w_off;         // element offset of the class's packed weight matrix
};

struct Tc5Args {
    const half_t* in;      // [N][Hi][Wi][Ci]
    const half_t* w;
    half_t* out;           // [N][Ho][Wo][CoStore]
    const float* bias;
    int32_t N, Hi, Wi, Ci;
    int32_t Ho, Wo, CoStore, Co;
    int32_t act, nchunks;                        // nchunks = Ci / 64 (even)
    int32_t pw_log2, ph_log2, PH, IPB;           // tile of 128 class-grid positions: PW x PH x IPB images
    int32_t tiles_x, tiles_y, ntiles;
    int32_t IH, IW, nslice;                      // union window (PH + 2) x (PW + 2); 4 KB DMA slices per 64-channel chunk
    uint32_t in_bytes, w_bytes;                  // buffer descriptor ranges (in_bytes < 2^31)
    FastDiv fdTPI, fdTX, fdIHW, fdIW;
    StatEpi st;
    BnBwdEpi bb;
    AffEpi aff;                                  // igemm_tc5w only
    int32_t solo, pad_solo;                      // igemm_tc5w: one parity class per block (grid.z = 4) for launches of few tiles
    Tc5Class cls[4];
};

// window-resident stride-2 convolution k5 p2 (igemm_c5.hip): 128 output pixels x 128 channels per block
struct C5Args {
    const half_t* in;      // [N][Hi][Wi][Ci], Ci % 32 == 0
    const half_t* w;       // [copad][Kpad], column = (ky*5 + kx)*Ci + ci
    half_t* out;           // [N][Ho][Wo][CoStore]
    int32_t N, Hi, Wi, Ci;
    int32_t Ho, Wo, CoStore, Co;
    int32_t Kpad, nsub;                          // nsub = Ci / 32
    int32_t pw16;                                // 1: tiles of 8 x 16 pixels of one image, 0: 8 x 8 pixels of two images
    int32_t tiles_x, tiles_y, ntiles;
    int32_t tpb;                                 // consecutive tiles per block (grid.x = ceil(ntiles / tpb))
    uint32_t in_bytes, w_bytes;                  // buffer descriptor ranges (in_bytes < 2^31)
    FastDiv fdTPI, fdTX;
    StatEpi st;                                  // one row per BLOCK; tpg[0] = blocks per statistics group
    BnBwdEpi bb;
    AffEpi aff;                                  // igemm_c5w only
};

// fused latent-discriminator MLP (mlp.hip): z -> H -> H -> H -> H -> 1, ReLU between the layers
struct MlpFwdArgs {
    const half_t* z;           // [M][Zp]
    int32_t M, Zp, H, pad0;
    const half_t* w[5];        // forward-orientation packed weights of layers 0..4: [rows_pad][kp], row = output feature
    int32_t kp[5], pad1;
    const float* bias[5];      // may be null
    half_t* hs[4];             // hidden activations h1..h4 [M][H] (saved for the backward pass)
    float* logit;              // [M]
};

struct MlpBwdArgs {
    const half_t* dlogit;      // [M][ldl], column 0
    int32_t ldl, M, Zp, H, Z, pad0;
    const half_t* hs[4];
    const half_t* w4;          // row 0 of the output layer's forward-orientation matrix: [H]
    const half_t* wd[4];       // data-gradient orientation of layers 0..3: [rows_pad][kpd], row = input feature
    int32_t kpd[4];
    half_t* delta[4];          // cotangents of the pre-activations of h1..h4: [M][H]
    float* dbias[5];           // += inv_scale * column sums (null: skip)
    float* dz;                 // [M][Z] fp32, null: skip
    float inv_scale;
    int32_t pad1;
};

// weight-gradient implicit GEMM (wgrad.hip):
//   dW[a][tap*Bc + b] (+)= sum_m P[m][a] * Q[n, y*s+dy(tap), x*s+dx(tap), b],  m = (n, y, x)
struct WgradArgs {
    const half_t* P;       // [M][A]      (A multiple of 8)
    const half_t* Q;       // [N][Hq][Wq][Bc]
    float* out;            // [Apad][ldo] fp32, ldo = padded T*Bc
    const half_t* zero;
    int32_t N, Yc, Xc, A;
    int32_t Hq, Wq, Bc;
    int32_t s, T, TW, dy0, dx0, dstep;
    int32_t M, ldo;
    int32_t splits, steps_per_split, atomic;
    int32_t ncol_chunks;   // T*Bc/8
    int64_t slab_stride;   // atomic == 2: floats between the per-split output slabs
    const int* gate;       // device flag or null: *gate == 0 -> the launch does nothing (fmri_wgrad_if)
    FastDiv fdX, fdYX, fdTW, fdBc8;
};

// narrow-channel 5x5 stride-1 convolution (igemm_narrow.hip)
struct NarrowArgs {
    const half_t* in;      // [N][H][W][Ci]
    const half_t* w;       // packed [rows_pad][Kpad], row = co, k = tap*Ci + ci
    half_t* out;           // [N][H][W][CoStore]
    const float* bias;
    int32_t N, H, W;
    int32_t CoStore, Co, Kpad, act;
    int32_t tiles_y, tiles_x, ntiles;
};

// stride-2 transposed convolution (k5 p2) 128 -> <= 32 channels (igemm_tc32.hip)
struct Tc32Class {
    int32_t Yc, Xc;                    // class output grid (class (cy, cx) writes output pixel (2y + cy, 2x + cx))
    int32_t Kpad, pad0;
    int64_t w_off;
};

struct Tc32Args {
    const half_t* in;      // [N][Hi][Wi][128]
    const half_t* w;
    half_t* out;           // [N][Ho][Wo][CoStore]
    const float* bias;
    int32_t N, Hi, Wi, Ho, Wo;
    int32_t CoStore, Co, act;
    int32_t tiles_y, tiles_x, ntiles;  // 8 x 16 tiles of class-grid positions (all four classes per tile)
    const half_t* relu_y;  // null, or [N][Ho][Wo][CoStore]: out = (relu_y > 0) ? result : 0 (ReLU backward of the layer below)
    Tc32Class cls[4];
};

// window-resident weight gradient for stride-2 sampling (wgrad_win.hip): the gathered operand Q is split into its
// 4 parity planes; per plane the taps are unit shifts of one LDS-resident window
struct WgradWinArgs {
    const half_t* P;       // [M][A]
    const half_t* Q;       // [N][Hq][Wq][Bc]
    float* out;            // [Apad][ldo] fp32, column = tap*Bc + b
    const half_t* zero;
    int32_t N, Yc, Xc, A;
    int32_t Hq, Wq, Bc;
    int32_t pad, TW;                 // conv padding, taps per tap row (k)
    int32_t tiles_y, tiles_x;        // 8x8 output-pixel tiles per image
    int32_t ntiles, splits;          // grid.z = splits = max over planes
    int32_t plane_tps[4];            // tiles per split of plane (py*2 + px)
    int32_t plane_pieces[4];         // K pieces (blocks per row / column block) of the plane; splits = their maximum
    int32_t ldo, a_tiles;            // a_tiles = 128-row blocks
    int64_t slab_stride;             // elements between the per-split output slabs
    int32_t nsy[2], nsx[2];          // shifts per parity (2 or 3)
    int32_t tmin[2];                 // first plane shift per parity (same for rows and columns)
    const int* gate;                 // device flag or null: *gate == 0 -> the launch does nothing (fmri_wgrad_if)
    FastDiv fdTPI, fdTX;
};

// weight gradient of the 5x5 stride-1 convolutions between 32 and 3(8) channels (wgrad_narrow.hip)
struct WgradNarrowArgs {
    const half_t* P;       // [N][H][W][32]
    const half_t* Q;       // [N][H][W][8]
    float* out;            // [32..][ldo] fp32, pre-zeroed (atomic accumulation), column = tap*8 + b
    const half_t* zero;
    int32_t N, H, W;
    int32_t ldo, flip;
    int32_t tiles_y, tiles_x, ntiles;
    int32_t nslabs, pad0;          // out holds nslabs pre-zeroed partial matrices, slab_stride elements apart
    int64_t slab_stride;
    const int* gate;               // device flag or null: *gate == 0 -> the launch does nothing (fmri_wgrad_if)
};

// Pin a wave-uniform kernel-argument value in an SGPR.  Fields of the by-value argument struct live in the kernarg
// segment; under register pressure the compiler re-loads them (s_load + s_waitcnt lgkmcnt(0)) wherever they are used --
// inside a K loop that is a chain of scalar-cache round trips per step that also drains the LDS counter.  After this
// the value is opaque to the compiler: it must keep (or spill) the register instead of re-loading.
#define FMRI_KEEP(x) asm volatile("" : "+s"(x))

// 16-byte global -> LDS DMA.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Same DMA issued from inline asm.  The compiler does not see an LDS write, so it does not put `s_waitcnt vmcnt(0)`
// in front of every later ds_read that might alias it (which silently turns a counted-vmcnt pipeline into a drained
// one); the caller owns the vmcnt / barrier protocol for the DMA'd bytes.  M0 is reserved (set before each use).
__device__ __forceinline__ void glds16_raw(const void* gsrc, void* lds_dst) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "memory");
}

// XCD-aware block -> tile map for a (tiles_x, tiles_y) grid whose blocks with the same x share an operand tile.
// Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each has its own L2), so neighbouring ids never share
// an L2.  Logical ids are laid out so that a run of nb/8 of them lives on one XCD, with y fastest: the blocks that
// re-read the same A tile (and the spatially adjacent tiles that share its halo) then hit in that XCD's L2.
// Speed only -- any bijection is correct; falls back to the identity when nb is not a multiple of 8.
__device__ __forceinline__ void xcd_tile(int& tx, int& ty) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nb = gx * gy;
    int l = blockIdx.x + gx * blockIdx.y;
    if ((nb & 7) == 0) l = (l & 7) * (nb >> 3) + (l >> 3);
    ty = l % gy;
    tx = l / gy;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

// BatchNorm statistics out of MFMA accumulators laid out D[i = channel][j = pixel] (16x16x32, weights as the A operand):
// lane (fq = lane >> 4, frow = lane & 15) holds, per 16-channel tile tn, the per-lane sums over its pixels of channels
// tn*16 + fq*4 + 0..3.  The 16 lanes of a DPP row hold the same channels of 16 different pixels: one row sum each, then
// lane (fq, frow) owns channel (frow >> 2)*16 + fq*4 + (frow & 3) of the wave's 16*TN channels and adds the two sums
// of that channel to (vs, vq).
template <int TN>
__device__ __forceinline__ void stat_lane_add(const f4 (&ssum)[TN], const f4 (&ssq)[TN], int lane, float& vs,
                                              float& vq) {
    const int frow = lane & 15;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float s = row16_sum(ssum[tn][rg]);
            const float q = row16_sum(ssq[tn][rg]);
            if (frow == tn * 4 + rg) { vs += s; vq += q; }
        }
}

// The per-lane sums of stat_lane_add -> the block's StatEpi row.  The WM waves that cover the same channels (different
// pixel rows of the block tile) meet in `scratch` (LDS, (WM-1)*WN*128 floats, free after the K loop) in a fixed order,
// and the wm == 0 waves store row[co] = sum, row[C + co] = second sum.  Called by every thread of the block.
template <int TN, int WM, int WN>
__device__ __forceinline__ void stat_store(float vs, float vq, int lane, int wm, int wn, int co_block, float* scratch,
                                           float* row, int C) {
    const int frow = lane & 15, fq = lane >> 4;
    const int ch = (frow >> 2) * 16 + fq * 4 + (frow & 3);      // channel within the wave's 16*TN
    __syncthreads();                                             // everyone is done with the operand tiles in LDS
    if (WM > 1 && wm > 0) {
        float* dst = scratch + ((wm - 1) * WN + wn) * 128;
        dst[ch] = vs;
        dst[64 + ch] = vq;
    }
    if (WM > 1) __syncthreads();
    if (wm == 0 && (frow >> 2) < TN) {
#pragma unroll
        for (int w = 1; w < WM; ++w) {
            const float* src = scratch + ((w - 1) * WN + wn) * 128;
            vs += src[ch];
            vq += src[64 + ch];
        }
        const int co = co_block + wn * (16 * TN) + ch;
        if (co < C) {
            row[co] = vs;
            row[C + co] = vq;
        }
    }
}

// BnBwdEpi fields of statistics group g (wave-uniform), selected without indexing the kernel-argument arrays with a
// run-time value (which makes the compiler copy them to scratch)
__device__ __forceinline__ void bn_bwd_group(const BnBwdEpi& b, int g, const float*& mean, const float*& rstd,
                                             int& x_img0) {
    mean = b.mean[0]; rstd = b.rstd[0]; x_img0 = b.x_img0[0];
    if (g == 1) { mean = b.mean[1]; rstd = b.rstd[1]; x_img0 = b.x_img0[1]; }
    if (g == 2) { mean = b.mean[2]; rstd = b.rstd[2]; x_img0 = b.x_img0[2]; }
    if (g == 3) { mean = b.mean[3]; rstd = b.rstd[3]; x_img0 = b.x_img0[3]; }
}

// BnBwdEpi transform of 4 consecutive channels of one pixel: v = dy (fp32 accumulators), xr = saved forward values;
// returns the stored g and adds (g, g*xhat) to the running sums.
__device__ __forceinline__ h4 bn_bwd_mask4(const f4& v, const h4& xr, const f4& mu, const f4& rs, const f4& ga,
                                           const f4& be, int relu, f4& s0, f4& s1) {
    h4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xh = ((float)xr[r] - mu[r]) * rs[r];
        const bool on = !relu || (xh * ga[r] + be[r] > 0.f);
        hv[r] = (half_t)(on ? v[r] : 0.f);
        const float g = (float)hv[r];
        s0[r] += g;
        s1[r] += g * xh;
    }
    return hv;
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

}  // namespace fmri
