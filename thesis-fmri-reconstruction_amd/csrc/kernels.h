// Internal launcher declarations shared between the kernel translation units and api.hip.
#pragma once
#include "common.h"

namespace fmri {

struct PackArgs {
    const float* src;
    half_t* dst;
    int64_t sa, sta, sb, stb;
    int32_t A, TA, B, Bp;          // Bp = B padded to a multiple of 8
    int32_t KW, py, px, step, TH, TW;
    int32_t rows_pad, kpad;        // destination matrix [rows_pad][kpad]
};

struct PackEntry {      // one row of the device-resident table of fmri_pack_weight_batch
    PackArgs p;
    int32_t run;
    int32_t tile_begin;
};

struct UnpackArgs {
    const float* src;      // packed [rows][ld]
    float* dst;            // reference layout
    int64_t sa, sta, sb, stb;
    int32_t A, TA, B, Bp;
    int32_t KW, py, px, step, TH, TW;
    int32_t ld;
    float scale;
    int32_t accumulate;
    int32_t nslabs;        // src holds nslabs partial matrices, slab_stride elements apart; they are summed
    int64_t slab_stride;
};

// One row of the device-resident table of fmri_apply_batch (round 4): everything that happens to ONE parameter tensor
// between its weight-gradient GEMM and the next forward pass -- slab sum, scaling, the map from the GEMM layout to the
// reference layout, the optimizer update, the fp16 GEMM copy in the gradient's own orientation -- in one pass.
struct ApplyEntry {
    const float* gsrc;     // kinds 0 / 1: packed fp32 gradient [nslabs][rows][ld] (a weight-gradient GEMM's output)
    float* w;              // master parameters, reference layout: tensor base (kinds 0 / 1) or segment base (kind 2)
    float* sq;             // RMSprop state at the same offsets
    float* grad;           // reference-layout gradient: kind 2 reads it (mode 1) / clears it (mode 2); kinds 0 / 1 write it in mode 0
    half_t* pk;            // fp16 GEMM copy [rows_pad][kpad] in the orientation of gsrc, or null
    int64_t sa, sta, sb;   // reference-layout strides of a, ta, b (taps are contiguous)
    int64_t slab_stride;
    int64_t n;             // kind 2: elements of the segment
    int32_t A, TA, B, Bp;
    int32_t run;           // taps (kind 0)
    int32_t ld, kpad, nslabs;
    int32_t clear;         // write zeros back over gsrc (buffers a weight-gradient kernel ADDS into)
    int32_t kind;          // 0 tap-transposing tile, 1 row-contiguous, 2 flat segment of the parameter buffer
    int32_t bt;            // kind 0: b per tile (32 or 64)
    int32_t tile_begin;
    float scale;
    int32_t pad_;
};
struct ApplyOpt {
    const float* lr_dev; const float* gdev; const int* flag;
    float alpha, eps, gscale, clamp;
    int32_t gated;         // the weight gradients were launched under the same flag (fmri_wgrad_if): when it is 0 they
                           // added nothing, so there is nothing to clear either
    int32_t mode;          // 0: gradients only (reference layout, first-writer stores), 1: RMSprop update + fp16 copy,
                           // 2: clear the flat segments' gradients, 3: as 1 with the gradient read from the reference
                           //    layout (`grad`, e.g. after an all-reduce) instead of gsrc
};
constexpr int APPLY_CHUNK = 1024;       // elements per block of kinds 1 / 2
int apply_entry_tiles(ApplyEntry& e, int TH, int TW, int KW, int py, int px, int step, int64_t stb);
int apply_batch_launch(const ApplyEntry* tab, int n, int total_tiles, const ApplyOpt& o, hipStream_t st);

int igemm_launch(const IgemmArgs& a, int maxM, int bn_tile, int copad, bool out_f32, hipStream_t st);
int igemm_bm(const IgemmArgs& a, int maxM, int bn_tile, int copad, bool out_f32);   // row tile igemm_launch will use
int igemm_tc32_launch(const Tc32Args& a, int nblocks, hipStream_t st);
int igemm_tc5_launch(const Tc5Args& a, int bn_tile, int copad, hipStream_t st);
int igemm_c5_launch(const C5Args& a, int copad, hipStream_t st);
int igemm_c5w_launch(const C5Args& a, int copad, hipStream_t st);
int igemm_tc5w_launch(const Tc5Args& a, int copad, hipStream_t st);
int mlp_fwd_launch(const MlpFwdArgs& a, hipStream_t st);
int mlp_bwd_launch(const MlpBwdArgs& a, hipStream_t st);
int igemm_narrow_launch(const NarrowArgs& a, int ci, int co_tiles, bool flip, hipStream_t st);
int wgrad_launch(const WgradArgs& a, int apad, int ba_tile, hipStream_t st);
int wgrad_win_launch(const WgradWinArgs& a, int apad, hipStream_t st);
int wgrad_narrow_launch(const WgradNarrowArgs& a, int nblocks, hipStream_t st);

int pack_weight_launch(const PackArgs& p, hipStream_t st);
int unpack_grad_launch(const UnpackArgs& p, hipStream_t st);
int pack_tile_count(const PackArgs& p, int* run_out);
int pack_batch_launch(const PackEntry* tab, int n, int total_tiles, hipStream_t st);
struct TransposeEntry {    // one row of the device-resident table of fmri_transpose_f16_batch
    const half_t* src;     // first element of the [R][C] source slice (row stride lds)
    half_t* dst;           // first element of the [C][R] destination slice (row stride ldd)
    int32_t R, C, Rbuf;    // Rbuf >= R rows exist behind src (the rows from R on are zero)
    int32_t width;         // source columns readable from src on (>= C; whole 16-byte items are read)
    int32_t lds, ldd;
    int32_t tile_begin, pad_;
};
int transpose_batch_launch(const TransposeEntry* tab, int n, int total_tiles, hipStream_t st);
int transpose_f16_launch(const half_t* src, half_t* dst, int R, int C, int Rbuf, int lds_, int ldd, hipStream_t st);
int nchw_to_nhwc_launch(const float* s, half_t* d, int N, int C, int HW, int Cp, hipStream_t st);
int nhwc_to_nchw_launch(const half_t* s, float* d, int N, int C, int HW, int Cp, float scale, hipStream_t st);
int rows_f32_to_f16_launch(const float* s, half_t* d, int M, int C, int Cp, float scale, hipStream_t st);
int rows_f16_to_f32_launch(const half_t* s, float* d, int M, int C, int Cp, float scale, hipStream_t st);
int reduce_slabs_launch(const float* slabs, int nslabs, int64_t slab_stride, int M, int C, int ld, const float* bias,
                        int act, float* out32, int ld32, half_t* out16, int ld16, hipStream_t st);
int permute_chw_launch(const float* s, float* d, int C, int HW, int to_engine, float scale, int accumulate,
                       hipStream_t st);

int64_t bn_ws_floats(int M, int C);
int bn_stats_launch(const half_t* x, int M, int C, float* sums, float* ws, int64_t ws_floats, hipStream_t st);
int bn_bwd_reduce2_launch(const half_t* x, const half_t* dy, int M, int C, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int relu, float* sums4C, float* ws, int64_t ws_floats,
                          float* dbeta, float* dgamma, float gscale, int param_stream, hipStream_t st);
int bn_bwd_apply2_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, float count, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int relu, const float* sums4C,
                         hipStream_t st);
int bn_stats_finalize_launch(const half_t* x, int M, int C, float* sums, float* ws, int64_t ws_floats, float count,
                             const float* gamma, const float* beta, float eps, float momentum, int updates, float* rm,
                             float* rv, float* mean, float* rstd, float* scale, float* shift, long long* nbt,
                             hipStream_t st);
int bn_bwd_reduce_launch(const half_t* x, const half_t* dy, int M, int C, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int relu, float* sums, float* ws, int64_t ws_floats,
                         float* dbeta, float* dgamma, float gscale, hipStream_t st);
constexpr int FOLD_STAGE_ROWS = 32;     // rows of the intermediate buffer of a two-stage statistics fold
int bn_fold_finalize_launch(const float* part, int rows, int C, float* scratch, float* sums, float count,
                            const float* gamma, const float* beta, float eps, float momentum, int updates, float* rm,
                            float* rv, float* mean, float* rstd, float* scale, float* shift, long long* nbt,
                            hipStream_t st);
int bn_fold_launch(const float* part, int rows, int n, float* scratch, float* sums, hipStream_t st);
int bn_bwd_fold_launch(const float* part, int rows, int rows_cap, int C, int G, float* scratch, float* sums,
                       float* dbeta, float* dgamma, float gscale, int pgroup, hipStream_t st);
int bn_finalize_launch(const float* sums, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* rm, float* rv, float* mean, float* rstd, float* scale,
                       float* shift, long long* nbt, const float* in_scale, hipStream_t st);
int bn_cols_fwd_launch(const half_t* x, half_t* y, int M, int C, float count, const float* gamma, const float* beta,
                       float eps, float momentum, int updates, float* rm, float* rv, float* mean, float* rstd,
                       float* scale, float* shift, float* sums2C, long long* nbt, int relu, const float* in_scale,
                       hipStream_t st);
int bn_cols_bwd_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, int nstreams, float count,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, float* sums,
                       float* dbeta, float* dgamma, float gscale, int pstream, hipStream_t st);
int bn_apply_launch(const half_t* x, half_t* y, int M, int C, const float* scale, const float* shift, int relu,
                    hipStream_t st);
int bn_bwd_apply_launch(const half_t* x, const half_t* dy, half_t* dx, int M, int C, float count, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int relu, const float* sums,
                        hipStream_t st);
int act_bwd_launch(const half_t* y, const half_t* dy, half_t* dpre, int M, int C, int act, float* colsum, float* ws,
                   int64_t ws_floats, float* dbias, int dbias_n, float gscale, hipStream_t st);
int colsum_rows_launch(const half_t* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float* dbias,
                       int dbias_n, float gscale, hipStream_t st);
int colsum_acc_launch(const void* src, int is_f16, int M, int C, int64_t ld_row, int64_t ld_col, float scale, float* dst,
                      hipStream_t st);

// api.hip: routing probe (fmri_igemm_route).  Every igemm launch function calls route_probe() with the name of the kernel
// instantiation it is about to launch, after its own support checks and before any HIP call; while a probe is active on
// the calling thread the name is recorded and `true` returned: the launcher then returns OK without touching the GPU.
bool route_probe(const char* fmt, ...);

// loss.hip: process-wide deterministic-reduction switch (fmri_set_deterministic)
extern int g_deterministic;

int latent_fwd_launch(const float* head, const float* eps, int B, int Z, int zp, half_t* z16, float* kl_rows,
                      float* kl_total, int sample, hipStream_t st);
int latent_ranged_launch(const float* head, const float* eps, int B, int Z, int zp, half_t* z16, float* kl_rows,
                         float* kl_total, int sample, float* z32, float* zmax, float* zscale, float cap, int phase,
                         hipStream_t st);
int rows_absmax_launch(const float* x, int64_t n, float* zmax, hipStream_t st);
float latent_range_scale_host(float zmax, float cap);
int latent_bwd_launch(const float* head, const float* eps, const float* dz, int ldz, float dz_unscale, float kl_w,
                      const float* kl_dev, int B, int Z, float out_scale, half_t* dhead16, float* dhead32, int sample,
                      hipStream_t st);
int feat_mse_launch(const half_t* feat, int B, int F, float* mse_rows, float* mse_total, hipStream_t st);
int feat_mse_bwd_launch(const half_t* feat, int B, int F, half_t* dfeat, float gscale, const float* norm,
                        hipStream_t st);
int pixel_sq_launch(const half_t* x, const half_t* xt, int64_t npix, int C, int Cp, float* total, half_t* dxt,
                    float gscale, hipStream_t st);
int gan_head_launch(const float* logit, int ldl, int B, float* prob, float* scal, int parts, hipStream_t st);
int gan_head_bwd_launch(const float* logit, int ldl, int B, half_t* dlogit, int ldg, float gscale, const float* norm,
                        int parts, hipStream_t st);
int wae_logloss_launch(const float* logit, int ldl, int n, int one_minus, float w, float* total, float* prob,
                       half_t* dlogit, int ldg, float gscale, hipStream_t st);
int compose_gate_launch(float* scal, int* flags, float batch, float nfeat, float npix, float lambda_mse,
                        float equilibrium, float margin, float beta, const float* hp_dev, int mode, int gate_on,
                        int force_dis, int force_dec, hipStream_t st);
int counter_inc_launch(int* t, hipStream_t st);
int axpby_f16_launch(const half_t* x, const half_t* y, half_t* out, int64_t n, float a, float b, const float* pa,
                     const float* pb, hipStream_t st);
int sumsq_launch(const float* x, int64_t n, float* acc, hipStream_t st);
int renorm_launch(const float* x, half_t* out, int64_t n, float scale, const float* sumsq, float count,
                  const float* factor_in, float* factor_out, hipStream_t st);
int sumsq64_launch(const float* x, int64_t n, double* acc, int zero_first, hipStream_t st);
int renorm64_launch(const float* x, half_t* out, int64_t n, float scale, const double* sumsq, float count,
                    const float* factor_in, float* factor_out, hipStream_t st);
int rmsprop_launch(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float gscale,
                   const float* gdev, float clamp, const int* flag, const float* lr_dev, hipStream_t st);
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                float bc1, float bc2_sqrt, float gscale, const float* gdev, float clamp, const int* flag,
                const float* lr_dev, const int* t_dev, hipStream_t st);

int ingest_u8_launch(const uint8_t* src, int N, int H, int W, int C, const int* flip, const int* shift,
                     const float* mean3, const float* std3, half_t* dst16, float* dst32, hipStream_t st);
int resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize_cap);
int crop_resize_u8_launch(const uint8_t* pool, const int64_t* offsets, const int32_t* dims, int N, int crop, int S,
                          const int32_t* hb, const int32_t* hk, int hks, const int32_t* vb, const int32_t* vk, int vks,
                          int vcount_max, uint8_t* out, hipStream_t st);
int pcc_launch(const float* x, const float* y, int64_t n, double* sums5, float* out, hipStream_t st);
int ssim_launch(const float* a, const float* b, int planes, int H, int W, double* acc2, float* ssim, float* contrast,
                hipStream_t st);

}  // namespace fmri
