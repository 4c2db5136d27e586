/*
 * fmri_hip.h -- C ABI of libfmri_hip.so, the MI355X (gfx950) kernel library behind the drop-in
 * `models.vae_gan` engine.
 *
 * The reference (MariaPdg/thesis-fmri-reconstruction) is pure Python on PyTorch: it has no FFI of its
 * own.  The boundary this library sits behind is the ATen operator set that `models/vae_gan.py` calls
 * (SURVEY.md 8a row a15); each entry point below names the reference call site(s) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer (incl. workspaces) is owned by the caller
 *     (the PyTorch caching allocator in our host code); the library allocates nothing and keeps no
 *     mutable global state => re-entrant, callable from autograd worker threads.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no host sync.
 *   - return 0 on success, a negative fmri_err otherwise; never throws, never exits.
 *   - activations: fp16 NHWC rows [N][H][W][C] with C a multiple of 8 (images are channel-padded 3->8);
 *     weights: fp16 "packed" matrices [rows_pad][kpad] produced by fmri_pack_weight;
 *     reductions / statistics / losses / master weights: fp32.
 */
#ifndef FMRI_HIP_H
#define FMRI_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum fmri_err { FMRI_OK = 0, FMRI_E_BADARG = -1, FMRI_E_UNSUPPORTED = -2, FMRI_E_LAUNCH = -3, FMRI_E_WORKSPACE = -4 };
enum fmri_act { FMRI_ACT_NONE = 0, FMRI_ACT_RELU = 1, FMRI_ACT_TANH = 2, FMRI_ACT_SIGMOID = 3 };
/* contraction modes of fmri_igemm */
enum fmri_mode {
    FMRI_CONV = 0,        /* correlation: in pixel = out*stride + k - pad          (nn.Conv2d fwd, deconv dgrad) */
    FMRI_TCONV2 = 1,      /* stride-2 transposed conv as 4 parity classes          (nn.ConvTranspose2d fwd, conv-s2 dgrad) */
    FMRI_CONV_FLIP = 2    /* stride-1 correlation with flipped taps: in = out + pad - k   (conv-s1 dgrad) */
};

int fmri_version(void);
const char* fmri_last_error_string(int code);

/* exact division helper exported for the host-side tests of the kernels' index arithmetic */
uint32_t fmri_test_fastdiv(uint32_t n, uint32_t d);

/* Geometry of parity class (cy,cx) of a k x k, stride-2, pad-p transposed convolution: first tap
 * (py,px), tap grid TH x TW, padded K and element offset of the class's packed weight block. */
int fmri_tconv_class(int k, int pad, int cy, int cx, int ci, int rows_pad, int* py, int* px, int* th, int* tw,
                     int* kpad, int64_t* w_off);
/* padded K (multiple of 64) of a T-tap, ci-channel reduction */
int fmri_kpad(int taps, int ci);

/* ---- weights ------------------------------------------------------------------------------------
 * dst[(ta*A + a)][(tb*Bp + b)] = (fp16) src[a*sa + ta*sta + b*sb + t(tb)*stb],  t(tb) = (py+step*ty)*KW + (px+step*tx),
 * zero padded to [rows_pad][kpad].  Replaces the implicit weight layout handling of ATen conv/linear
 * (models/vae_gan.py:18,46,79,107,119,146,156) incl. the (C,H,W) flatten order at :89,:127,:181. */
int fmri_pack_weight(const float* src, void* dst, int64_t sa, int64_t sta, int64_t sb, int64_t stb, int A, int TA,
                     int B, int KW, int py, int px, int step, int TH, int TW, int rows_pad, int kpad, void* stream);
/* Batched fmri_pack_weight: a device-resident table of rows (filled on the host with fmri_pack_entry_fill, which
 * returns the number of blocks of the row, 0 = not eligible, use fmri_pack_weight) repacked by ONE launch --
 * what a sub-network needs after its optimizer step. */
int fmri_pack_entry_bytes(void);
int fmri_pack_entry_fill(void* host_entry, const float* src, void* dst, int64_t sa, int64_t sta, int64_t sb,
                         int64_t stb, int A, int TA, int B, int KW, int py, int px, int step, int TH, int TW,
                         int rows_pad, int kpad, int tile_begin);
int fmri_pack_weight_batch(const void* table_dev, int n, int total_tiles, void* stream);
/* inverse map for fp32 weight gradients: dst[...] (+)= scale * sum_z src[z*slab_stride + (ta*A+a)*ld + tb*Bp + b],
 * z < nslabs (the per-split partial results of fmri_wgrad, mode 2) */
int fmri_unpack_grad(const float* src, float* dst, int64_t sa, int64_t sta, int64_t sb, int64_t stb, int A, int TA,
                     int B, int KW, int py, int px, int step, int TH, int TW, int ld, float scale, int accumulate,
                     int nslabs, int64_t slab_stride, void* stream);

/* ---- contractions (MFMA) ------------------------------------------------------------------------
 * out = act(bias + contraction(in, w)); see csrc/igemm.hip.  Replaces F.conv2d / F.conv_transpose2d /
 * F.linear forward and data-gradient (models/vae_gan.py:26,32,57,90-92,126,128,175,182).
 * out_f32 != 0: raw fp32 result written to `splits` slabs (slab_stride elements apart), no bias/act. */
int fmri_igemm(const void* in, const void* w, void* out, const float* bias, const void* zero16, int N, int Hi, int Wi,
               int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad, int mode, int act,
               int out_f32, int splits, int64_t slab_stride, int bn_tile, void* stream);
/* The same contraction with an epilogue that a following BatchNorm needs (models/vae_gan.py:26-34,57-59: conv -> bn):
 * every block writes, per output channel, sum x and sum x^2 of the values it STORED (valid output pixels only) to its
 * own row of stat_part[group][stat_rows_cap][2][CoStore] (plain stores, deterministic), group = image / stat_group_n
 * (stat_group_n = 0: one group; a row tile never mixes groups).  *ep_done = rows written per group (they are the
 * first rows of each group; fold them with fmri_bn_fold_finalize), or 0 when the kernel behind this geometry has no
 * such epilogue, a tile would straddle two groups or stat_rows_cap is too small -- then use fmri_bn_stats on the
 * output.  N*Ho*Wo/128 + 8 rows per group always suffice.  w_elems: elements of the packed weight buffer `w` (the
 * descriptor-addressed kernels bound their weight DMA with it; 0 = unknown, those kernels are skipped). */
typedef struct fmri_epilogue {
    float* stat_part;        /* NULL: no statistics */
    int32_t stat_rows_cap;   /* rows allocated per group */
    int32_t stat_group_n;    /* images per statistics group, 0 = all images */
    /* BatchNorm (+ReLU) BACKWARD statistics, for the data gradient that produces the cotangent dy of a BatchNorm output
     * (autograd of models/vae_gan.py:28-29,58-59): with the saved forward input bn_x of that BatchNorm (geometry of
     * `out`), xhat = (x - mean)*rstd and on = !relu || xhat*gamma + beta > 0, `out` receives g = on ? dy : 0 and the rows
     * hold (sum g, sum g*xhat).  Group i (<= 4: cotangent streams / decoder calls stacked along the batch) reads x from
     * image bn_x_img0[i] on, with the batch statistics of its forward call.  NULL bn_x: forward statistics.  Requires
     * stat_part, no bias / activation; if *ep_done comes back 0 `out` holds the plain dy (use fmri_bn_bwd_reduce). */
    const void* bn_x;
    const float* bn_gamma;
    const float* bn_beta;
    const float* bn_mean[4];
    const float* bn_rstd[4];
    int32_t bn_x_img0[4];
    int32_t bn_relu;
    int32_t reserved;
    /* ReLU backward of the layer BELOW, for a data gradient whose output is the cotangent of that layer's ReLU output
     * (autograd of nn.ReLU after discriminator.conv.0, models/vae_gan.py:145-147): act_y = the saved ReLU output
     * (geometry of `out`), `out` receives (act_y > 0) ? dy : 0.  Independent of the statistics fields.  Kernels that
     * apply it set FMRI_EP_ACT_APPLIED in *ep_done; if the bit comes back clear `out` holds the plain dy (use
     * fmri_act_bwd). */
    const void* act_y;
    /* Per-channel affine map (+ ReLU) on the fp32 accumulators in front of the fp16 store: out = relu?(acc * aff_scale[co] +
     * aff_shift[co]) -- an eval-mode BatchNorm (running statistics, models/vae_gan.py:288-297 under model.eval()) folded
     * into the convolution that feeds it: no pass over the raw output, one rounding less.  Vectors of CoStore floats.  Only
     * without statistics, bias and activation.  Kernels that apply it set FMRI_EP_AFFINE_APPLIED in *ep_done; if the bit
     * comes back clear `out` holds the plain contraction (use fmri_bn_apply). */
    const float* aff_scale;
    const float* aff_shift;
    int32_t aff_relu;
    int32_t reserved2;
} fmri_epilogue;
#define FMRI_EP_ACT_APPLIED 0x40000000
#define FMRI_EP_AFFINE_APPLIED 0x20000000
int fmri_igemm_ep(const void* in, const void* w, void* out, const float* bias, const void* zero16, int N, int Hi,
                  int Wi, int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad, int mode, int act,
                  int out_f32, int splits, int64_t slab_stride, int bn_tile, int64_t w_elems, const fmri_epilogue* ep,
                  int* ep_done, void* stream);
/* Name of the kernel instantiation fmri_igemm_ep routes a call with these arguments to, e.g.
 * "fmri::igemm_c5w_kernel<16,1>" ("none": nothing to launch).  The device pointers of fmri_igemm_ep are replaced by
 * flags (has_bias; stat_rows_cap > 0: statistics epilogue asked with that row capacity / stat_group_n; want_bn_bwd,
 * want_act_y, want_affine: the other fmri_epilogue fields set).  Pure host code -- no GPU is touched, it runs on a machine
 * without one -- and the ONLY statement of the routing: profilers, benches and tests ask it instead of mirroring the
 * rules.  name_out receives at most cap - 1 characters. */
int fmri_igemm_route(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoStore, int Co, int k, int stride, int pad,
                     int mode, int act, int out_f32, int splits, int bn_tile, int64_t w_elems, int has_bias,
                     int stat_rows_cap, int stat_group_n, int want_bn_bwd, int want_act_y, int want_affine, char* name_out,
                     int cap);
/* dW[a][tap*Bc+b] (+)= sum_m P[m][a] * Q[gather(m,tap)][b]; see csrc/wgrad.hip.  Replaces the weight
 * gradients autograd computes for the same modules (train/train_vgan_stage1.py:412,422,430). */
int fmri_wgrad(const void* P, const void* Q, float* out, const void* zero16, int N, int Yc, int Xc, int A, int Hq,
               int Wq, int Bc, int k, int stride, int pad, int flip, int apad, int ba_tile, int ldo, int splits,
               int atomic, void* stream);
/* The same launch under a device-side condition: with gate != NULL and *gate == 0 when the kernel starts, it does
 * nothing (`out` is left as it is).  The reference runs `loss_decoder.backward()` / `loss_discriminator.backward()` only
 * `if train_dec:` / `if train_dis:` (train_vgan_stage1.py:420-431: the equilibrium gate of :395-403); the engine's
 * gate is evaluated on the device (fmri_compose_gate_dev), so the host always issues the launches and the kernels of a
 * sub-network that is not trained in this step retire at once. */
int fmri_wgrad_if(const int* gate, const void* P, const void* Q, float* out, const void* zero16, int N, int Yc, int Xc,
                  int A, int Hq, int Wq, int Bc, int k, int stride, int pad, int flip, int apad, int ba_tile, int ldo,
                  int splits, int atomic, void* stream);
/* atomic = 0: one split, plain stores.  1: atomic adds into a pre-zeroed out.  2: split z stores its partial result
 * to out + z*apad*ldo (stride-2, 128-row, 32-channel-block geometries only -- csrc/wgrad_win.hip -- else
 * FMRI_E_UNSUPPORTED).  For that kernel `splits` is the block budget per (128-row, 32-channel) group over the 4
 * parity planes; the number of slabs it writes is fmri_wgrad_slabs().
 * atomic = 3: 5x5 stride-1 layers with A = 32, Bc = 8 only (csrc/wgrad_narrow.hip, else FMRI_E_UNSUPPORTED): out
 * holds `splits` pre-zeroed slabs of apad*ldo floats that the blocks add into round-robin; fmri_unpack_grad sums
 * them (`splits` = the number of blocks the kernel is launched with, fmri_wgrad_narrow_blocks(): one slab per block,
 * every element added exactly once onto zero -- bit-reproducible).
 * atomic = 4: per-split slabs of the generic kernel (csrc/wgrad.hip): out holds `splits` ZERO-FILLED slabs of apad*ldo
 * floats, split z stores its partial result to slab z with plain stores (the kernel may use fewer splits than asked
 * for; the remaining slabs stay zero); fmri_unpack_grad sums them in slab order -- bit-reproducible.
 * flip = 0: Q pixel = m*stride + tap - pad.  flip = 1 (stride 1 only): Q pixel = m + pad - tap, i.e. the roles of
 * the two activations are exchanged so that the GATHERED operand is the one with fewer channels. */
int fmri_wgrad_slabs(int N, int Yc, int Xc, int k, int pad, int splits);
/* number of blocks fmri_wgrad(..., atomic = 3) launches for this geometry */
int fmri_wgrad_narrow_blocks(int N, int Yc, int Xc);

/* ---- deterministic-reduction mode (process-wide; initial value: environment FMRI_DETERMINISTIC=1).  When on, the
 * loss / norm kernels that end in an atomic add onto a device scalar (fmri_latent_fwd, fmri_feat_mse, fmri_pixel_sq,
 * fmri_gan_head*, fmri_wae_logloss, fmri_sumsq, fmri_pcc) are launched as ONE block each, i.e. every total is a
 * fixed-order sum.  Together with the slab forms of fmri_wgrad (atomic = 2, 3 with one slab per block, 4) and
 * dbias5 = NULL in fmri_mlp_bwd -- which the caller selects -- two runs of the same training step on the same inputs
 * then produce bit-identical parameters.  A verification mode (the one-block sums cost a few hundred microseconds at
 * batch 256); the default trades the fixed order for atomics.  Returns the previous value. */
int fmri_set_deterministic(int on);
int fmri_get_deterministic(void);

/* ---- batch ingest: uint8 [N][H][W][C = 1|3] (already cropped / resized) -> normalised fp16 NHWC8 (engine input)
 * and / or fp32 NCHW (module API input).  Per image: optional horizontal flip (flip_dev[n] != 0, may be NULL), then an
 * integer shift (shift_dev[2n] rows, [2n+1] columns, edge-replicated like scipy.ndimage.shift(order=0, 'nearest'),
 * may be NULL), u8/255, grey -> 3 channels, (v - mean) / std.  Replaces RandomHorizontalFlip / RandomShift / ToTensor
 * / GreyToColor / Normalize of train_vgan_stage1.py:162-170 and data_preprocessing/data_loader.py:186-217,374-401. */
int fmri_ingest_u8(const uint8_t* src, int N, int H, int W, int C, const int* flip_dev, const int* shift_dev,
                   float mean0, float mean1, float mean2, float std0, float std1, float std2, void* dst16,
                   float* dst32, void* stream);

/* ---- CenterCrop((crop, crop)) + Resize((S, S)) of a ragged batch of decoded uint8 images, the head of the COCO
 * pipeline (train/train_vgan_stage1.py:162-165: torchvision 0.5.0 transforms on PIL images).  Bit-exact with
 * torchvision.transforms.functional.center_crop / .resize over Pillow's 8-bit ImagingResample (BILINEAR: antialiased
 * triangle filter, 22-bit fixed-point coefficients, horizontal pass rounded to uint8, then vertical pass).
 * fmri_resize_coeffs (HOST function, no stream): coefficient tables of one pass in_size -> out_size: bounds[out][2] =
 *   (first input index, count), coef[out][ksize]; returns ksize > 0, 0 when in_size == out_size (identity pass: no tables
 *   needed), FMRI_E_WORKSPACE when ksize > ksize_cap (ksize = 2 * ceil(max(in / out, 1)) + 1).
 * fmri_crop_resize_u8: pool = the images back to back (HWC, C = 1 or 3 per image), offsets_dev[n] = byte offset of image
 *   n, dims_dev[n] = (H, W, C); both passes resample crop -> S with the tables of fmri_resize_coeffs(crop, S) uploaded by
 *   the caller (hks / vks = their ksize, 0 = identity), vcount_max = the largest count of the vertical table (LDS rows
 *   per block; 1 for the identity).  out [N][S][S][3] uint8, grey images replicated to three channels (GreyToColor,
 *   data_preprocessing/data_loader.py:374-401).  Pixels of the crop box outside the image are zeros (PIL crop). */
int fmri_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize_cap);
int fmri_crop_resize_u8(const uint8_t* pool, const int64_t* offsets_dev, const int32_t* dims_dev, int N, int crop, int S,
                        const int32_t* hb_dev, const int32_t* hk_dev, int hks, const int32_t* vb_dev, const int32_t* vk_dev,
                        int vks, int vcount_max, uint8_t* out, void* stream);

/* ---- evaluation metrics of the validation loop (train/train_utils.py) ------------------------------
 * fmri_pcc : PearsonCorrelation.forward (:276-292) over n fp32 elements (whole batch), *out = coefficient.
 * fmri_ssim: StructuralSimilarity.forward (:343-420, size_average=True): mean SSIM (and the mean contrast term of
 *            full=True) over planes = N*C images of H x W (>= 11), 11x11 Gaussian sigma 1.5, zero padding.
 * ws5 / ws2: 5 / 2 doubles of device scratch (zeroed by the call). */
int fmri_pcc(const float* pred, const float* truth, int64_t n, double* ws5, float* out, void* stream);
int fmri_ssim(const float* img1, const float* img2, int planes, int H, int W, double* ws2, float* ssim,
              float* contrast, void* stream);

/* ---- layout casts ------------------------------------------------------------------------------- */
int fmri_nchw_to_nhwc(const float* src, void* dst, int N, int C, int HW, int Cp, void* stream);
int fmri_nhwc_to_nchw(const void* src, float* dst, int N, int C, int HW, int Cp, float scale, void* stream);
int fmri_rows_f32_to_f16(const float* src, void* dst, int M, int C, int Cp, float scale, void* stream);
int fmri_rows_f16_to_f32(const void* src, float* dst, int M, int C, int Cp, float scale, void* stream);
int fmri_reduce_slabs(const float* slabs, int nslabs, int64_t slab_stride, int M, int C, int ld, const float* bias,
                      int act, float* out32, int ld32, void* out16, int ld16, void* stream);
int fmri_permute_chw(const float* src, float* dst, int C, int HW, int to_engine, float scale, int accumulate,
                     void* stream);

/* ---- batch norm (train mode, momentum 0.9; models/vae_gan.py:21,54,81,108,158) -------------------- */
/* reductions write per-block partials to a caller workspace of fmri_bn_ws_floats(M, C) floats (any smaller
 * size >= 2*C also works, with fewer blocks) and fold them into sums2C = [sum a | sum b], 2*C floats. */
int64_t fmri_bn_ws_floats(int M, int C);
int fmri_bn_stats(const void* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, void* stream);
int fmri_bn_finalize(const float* sums2C, int C, float count, const float* gamma, const float* beta, float eps,
                     float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, int64_t* num_batches_tracked /* += updates, may be NULL */,
                     void* stream);
/* fmri_bn_stats + fmri_bn_finalize in two launches instead of three (the fold of the partial sums finalizes): the
 * forward of nn.BatchNorm2d/1d in train mode when no statistics are exchanged between ranks (models/vae_gan.py:21). */
int fmri_bn_stats_finalize(const void* x, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float count,
                           const float* gamma, const float* beta, float eps, float momentum, int updates,
                           float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                           float* shift, int64_t* num_batches_tracked, void* stream);
/* BatchNorm over FEW rows in ONE launch per direction (the BatchNorm1d layers behind the dense layers,
 * models/vae_gan.py:81,108,158,200: M = batch rows): a block owns 32 channels for all rows -- statistics, finalize
 * (incl. the running statistics and num_batches_tracked, `updates` momentum updates), apply / the backward constants,
 * parameter gradients and dx.  fmri_bn_cols_fwd = fmri_bn_stats_finalize + fmri_bn_apply; fmri_bn_cols_bwd (nstreams 1
 * or 2 cotangent streams stacked along the rows, sums [nstreams][2][C], dbeta / dgamma += gscale * sums of stream
 * param_stream, may be NULL) = fmri_bn_bwd_reduce(2) + fmri_bn_bwd_apply(2).  Fixed summation order. */
int fmri_bn_cols_fwd(const void* x, void* y, int M, int C, float count, const float* gamma, const float* beta, float eps,
                     float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, float* sums2C, int64_t* num_batches_tracked, int relu, void* stream);
/* fmri_bn_cols_fwd / fmri_bn_finalize for rows stored range-scaled, x_stored = (*in_scale) * x with *in_scale a power of
 * two (the latent batch behind fmri_latent_fwd_ranged; in_scale NULL = 1): eps is scaled by s^2, so the output equals
 * BatchNorm(x); mean / rstd / scale / shift describe the STORED rows (what the apply and backward kernels read) and the
 * running statistics receive mean / s and var / s^2 -- nn.BatchNorm1d behind `Decoder.fc` (models/vae_gan.py:107-109). */
int fmri_bn_cols_fwd_s(const void* x, void* y, int M, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                       float* scale, float* shift, float* sums2C, int64_t* num_batches_tracked, int relu,
                       const float* in_scale, void* stream);
int fmri_bn_finalize_s(const float* sums2C, int C, float count, const float* gamma, const float* beta, float eps,
                       float momentum, int updates, float* running_mean, float* running_var, float* mean, float* rstd,
                       float* scale, float* shift, int64_t* num_batches_tracked, const float* in_scale, void* stream);
int fmri_bn_cols_bwd(const void* x, const void* dy, void* dx, int M, int C, int nstreams, float count, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int relu, float* sums, float* dbeta,
                     float* dgamma, float gscale, int param_stream, void* stream);
/* The statistics rows a contraction's epilogue wrote (fmri_igemm_ep: stat_part [rows][2][C] of ONE group) folded into
 * sums2C and finalized like fmri_bn_finalize; scratch: fmri_bn_fold_scratch_floats(C) floats (two-stage fold of long
 * row lists).  fmri_bn_fold only folds (data-parallel runs all-reduce sums2C before fmri_bn_finalize). */
int fmri_bn_fold_finalize(const float* stat_part, int rows, int C, float* scratch, float* sums2C, float count,
                          const float* gamma, const float* beta, float eps, float momentum, int updates,
                          float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                          int64_t* num_batches_tracked, void* stream);
int fmri_bn_fold(const float* stat_part, int rows, int C, float* scratch, float* sums2C, void* stream);
int fmri_bn_fold_scratch_floats(int C);
/* BatchNorm-backward rows (fmri_epilogue.bn_x) of `groups` cotangent groups, stat_part [groups][rows_cap][2][C], folded
 * into sums [groups][2][C] = per group [sum g | sum g*xhat] (the sums4C layout of fmri_bn_bwd_apply2 for two groups);
 * dbeta / dgamma (may be NULL) += gscale * the sums of group param_group.  scratch: groups *
 * fmri_bn_fold_scratch_floats(C) floats. */
int fmri_bn_bwd_fold(const float* stat_part, int rows, int rows_cap, int C, int groups, float* scratch, float* sums,
                     float* dbeta, float* dgamma, float gscale, int param_group, void* stream);
int fmri_bn_apply(const void* x, void* y, int M, int C, const float* scale, const float* shift, int relu,
                  void* stream);
int fmri_bn_bwd_reduce(const void* x, const void* dy, int M, int C, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int relu, float* sums2C, float* ws,
                       int64_t ws_floats, float* dbeta /* += gscale * sums[0..C), may be NULL */,
                       float* dgamma /* += gscale * sums[C..2C), may be NULL */, float gscale, void* stream);
int fmri_bn_bwd_apply(const void* x, const void* dy, void* dx, int M, int C, float count, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int relu, const float* sums2C,
                      void* stream);
/* The same for TWO cotangent streams through one saved forward (the discriminator's logit and feature streams,
 * train/train_vgan_stage1.py:369-372 back-propagates both): dy2 / dx2 = [stream A rows | stream B rows], M rows each;
 * x, xhat and the ReLU mask are read / computed once.  sums4C = [A: sum g | A: sum g*xhat | B: sum g | B: sum g*xhat];
 * dbeta / dgamma (may be NULL) accumulate gscale * the sums of stream param_stream (0 = A, 1 = B); the workspace needs
 * 2 * fmri_bn_ws_floats. */
int fmri_bn_bwd_reduce2(const void* x, const void* dy2, int M, int C, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int relu, float* sums4C, float* ws, int64_t ws_floats,
                        float* dbeta, float* dgamma, float gscale, int param_stream, void* stream);
int fmri_bn_bwd_apply2(const void* x, const void* dy2, void* dx2, int M, int C, float count, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, int relu, const float* sums4C,
                       void* stream);
/* dpre = dy * act'(y) (ReLU / tanh); if colsum2C != NULL its first C floats receive the column sums of dpre and, with
 * dbias != NULL, dbias[c] += gscale * colsum[c] for c < dbias_n <= C (the bias gradient of the layer in front of the activation: autograd of
 * `nn.Conv2d(..., bias=True)` + Tanh, models/vae_gan.py:118-121) */
int fmri_act_bwd(const void* y, const void* dy, void* dpre, int M, int C, int act, float* colsum2C, float* ws,
                 int64_t ws_floats, float* dbias, int dbias_n, float gscale, void* stream);
/* dst[c] += scale * sum_{m < M} src[m*ld_row + c*ld_col], c < C; src fp16 (is_f16) or fp32.  Bias gradients of the dense
 * layers (column sums of the cotangent rows, what autograd's Linear backward reduces) in one launch, fixed-order sums. */
int fmri_colsum_acc(const void* src, int is_f16, int M, int C, int64_t ld_row, int64_t ld_col, float scale, float* dst,
                    void* stream);
/* The many-row form for contiguous fp16 rows [M][C], C % 8 == 0 (the bias gradient of discriminator.conv.0 over
 * 3B x 64 x 64 pixels): per-block partial sums in `ws` (fmri_bn_ws_floats(M, C) floats) + fold; sums2C receives
 * [sum x | sum x^2], dbias[c] += gscale * sum x[c] for c < dbias_n (dbias may be NULL). */
int fmri_colsum_rows(const void* x16, int M, int C, float* sums2C, float* ws, int64_t ws_floats, float* dbias,
                     int dbias_n, float gscale, void* stream);

/* ---- latent / losses (models/vae_gan.py:266-269, :302-320; train_vgan_stage1.py:368-404) ----------- */
int fmri_latent_fwd(const float* head, const float* eps, int B, int Z, int zp, void* z16, float* kl_rows,
                    float* kl_total, int sample, void* stream);
/* Range-safe `VaeGan.reparameterize` (models/vae_gan.py:266-269) for fp16 consumers: sigma = exp(0.5 logvar) leaves
 * fp16's range at logvar > 22.2, where the reference's fp32 arithmetic is still finite (it overflows at logvar > 88.7).
 *   phase & 1: z = eps * exp(0.5 logvar) + mu (sample) or mu -> z32 [B][Z] fp32, KL as fmri_latent_fwd,
 *              *zmax = max(*zmax, max |z|)   (the caller zeroes *zmax; data-parallel SyncBN runs all-reduce it with MAX)
 *   phase & 2: *zscale = s = the largest power of two <= 1 with s * (*zmax) <= cap (1 when *zmax <= cap or non-finite),
 *              z16 [B][zp] = fp16(s * z32), padding columns zero.
 * Consumers: the GEMM of `Decoder.fc` on z16, fmri_bn_cols_fwd_s / fmri_bn_finalize_s with in_scale = zscale; the data
 * gradient w.r.t. z is s times the one w.r.t. the stored rows, the weight gradient needs no correction.
 * fmri_rows_absmax: *zmax = max(*zmax, max |x|) for a caller-provided fp32 latent (then phase 2 with z32 = x). */
int fmri_latent_fwd_ranged(const float* head, const float* eps, int B, int Z, int zp, void* z16, float* kl_rows,
                           float* kl_total, int sample, float* z32, float* zmax, float* zscale, float cap, int phase,
                           void* stream);
int fmri_rows_absmax(const float* x, int64_t n, float* zmax, void* stream);
/* host only: the scale phase 2 derives from a batch maximum (the same function the kernel evaluates) */
float fmri_latent_range_scale(float zmax, float cap);
int fmri_latent_bwd(const float* head, const float* eps, const float* dz, int ldz, float dz_unscale, float kl_w,
                    const float* kl_dev, int B, int Z, float out_scale, void* dhead16, float* dhead32, int sample,
                    void* stream);
int fmri_feat_mse(const void* feat, int B, int F, float* mse_rows, float* mse_total, void* stream);
int fmri_pixel_sq(const void* x, const void* xt, int64_t npix, int C, int Cp, float* total, void* dxt, float gscale,
                  void* stream);
/* scal: float block, slots [0..2] += bce sums (orig, pred, sampled), [9] += sum (d bce / d logit)^2 */
int fmri_gan_head(const float* logit, int ldl, int B, float* prob, float* scal, void* stream);
/* fmri_gan_head / fmri_gan_head_bwd for a discriminator loss made of a subset of the three terms (bit 0 orig, 1 pred,
 * 2 sampled; 'dcgan' / 'vae': 5, train_vgan_stage1.py:375,382): slot [9] and the cotangent only see those parts. */
int fmri_gan_head_parts(const float* logit, int ldl, int B, float* prob, float* scal, int parts, void* stream);
int fmri_gan_head_bwd_parts(const float* logit, int ldl, int B, void* dlogit, int ldg, float gscale, const float* norm,
                            int parts, void* stream);
int fmri_wae_logloss(const float* logit, int ldl, int n, int one_minus, float w, float* total, float* prob,
                     void* dlogit, int ldg, float gscale, void* stream);
/* ---- the WAE latent discriminator (models/vae_gan.py:499-529: Linear(z,H) ReLU [Linear(H,H) ReLU] x 3 Linear(H,1),
 * sigmoid left to fmri_wae_logloss) as one launch forward and one for the backward chain (H = 512, Zp a multiple of 64
 * <= 256; anything else: FMRI_E_UNSUPPORTED, use fmri_igemm layer by layer).
 * fmri_mlp_fwd: z16 [M][Zp] fp16; w5[i] = layer i's packed fp16 weights in the forward orientation ([rows_pad][kp5[i]],
 *   row = output feature: what fmri_pack_weight produces for fmri_igemm), bias5[i] fp32 (may be NULL); hs4[i] receives
 *   the hidden activations h_{i+1} [M][H] fp16 (the backward pass needs them), logit [M] fp32 the pre-sigmoid output.
 * fmri_mlp_bwd: dlogit16 [M][ldl] (column 0) -> delta4[i] = cotangent of the pre-activation of h_{i+1} ([M][H] fp16:
 *   the P operands of the layers' fmri_wgrad calls), dbias5[i] += inv_scale * column sums (NULL dbias5 or entry: skip),
 *   dz32 [M][Z] fp32 = inv_scale * (delta1 . W0) (NULL: skip).  w4row = row 0 of the output layer's forward-orientation
 *   matrix; wd4[i] = layer i's packed weights in the data-gradient orientation ([rows_pad][kpd4[i]], row = input
 *   feature); wd4[0] is read only when dz32 is given.  Replaces the autograd data path of
 *   train/train_wae_stage1.py:278-303.  Determinism: delta4 and dz32 are plain stores (bit-reproducible); the five
 *   dbias5 vectors are accumulated with fp32 atomics across the 32-row blocks, so their last bits depend on the
 *   arrival order of the blocks (run-to-run spread ~1e-7 relative; every other gradient of the engine is reduced in a
 *   fixed order). */
int fmri_mlp_fwd(const void* z16, int M, int Zp, int H, const void* const* w5, const int* kp5, const float* const* bias5,
                 void* const* hs4, float* logit, void* stream);
int fmri_mlp_bwd(const void* dlogit16, int ldl, int M, int Zp, int Z, int H, const void* const* hs4, const void* w4row,
                 const void* const* wd4, const int* kpd4, void* const* delta4, float* const* dbias5, float* dz32,
                 float inv_scale, void* stream);
/* scal slots: in [0..5] bce_o,bce_p,bce_s,kl,mse,nle and [9] dl2; out [6..8] loss_encoder/discriminator/decoder,
 * [10] nA = 1/rms(d bce/d logit), [11] nB = 1/rms(d mse/d feature), [12] nA/nB, [13] 1.0; flags = {train_dis, train_dec} */
int fmri_compose_gate(float* scal, int* flags, float batch, float nfeat, float lambda_mse, float equilibrium,
                      float margin, int gate_on, int force_dis, int force_dec, void* stream);
/* The same with the hyper-parameters on the DEVICE, hp4_dev = [lambda_mse, equilibrium, margin, beta] -- the values the
 * scripts decay every epoch (train_vgan_stage1.py:448-458); a step recorded into a HIP graph then follows the schedule --
 * and with the other loss compositions of train_vgan_stage1.py:359-388: mode 0 'vae-gan', 1 'beta-vae' (KL weight
 * beta / batch), 2 'dcgan' (pixel nle in place of the feature mse, discriminator loss bce_orig + bce_sampled), 3 'vae'
 * (dcgan's losses, decoder loss lambda * nle, train_dis starts False).  npix = reals per image (3*H*W).  Further
 * slots written: [16] nP = 1/rms(d nle/d x_tilde), [17] lambda*nA/nB, [18] 1 - lambda, [19] lambda*nA, [20] the factor
 * the decoder gradients carry (nP/lambda for 'vae', else nA), [21] the KL weight. */
int fmri_compose_gate_dev(float* scal, int* flags, float batch, float nfeat, float npix, const float* hp4_dev, int mode,
                          int gate_on, int force_dis, int force_dec, void* stream);
/* *counter_dev += 1 (the step count fmri_adam_dev derives its bias corrections from) */
int fmri_counter_inc(int* counter_dev, void* stream);
/* starting cotangents of the two back-propagated streams, fp16, multiplied by gscale * (*norm) (norm: device float) */
int fmri_gan_head_bwd(const float* logit, int ldl, int B, void* dlogit, int ldg, float gscale, const float* norm,
                      void* stream);
int fmri_feat_mse_bwd(const void* feat, int B, int F, void* dfeat, float gscale, const float* norm, void* stream);
/* out = a * (*a_dev) * x + b * y   (fp16 tensors, fp32 math; y and a_dev may be NULL) */
int fmri_axpby_f16(const void* x, const void* y, void* out, int64_t n, float a, float b, const float* a_dev,
                   void* stream);

/* out = a * (*a_dev) * x + b * (*b_dev) * y */
int fmri_axpby2_f16(const void* x, const void* y, void* out, int64_t n, float a, float b, const float* a_dev,
                    const float* b_dev, void* stream);

/* device-side unit-RMS re-normalisation of an fp32 cotangent before an fp16 backward pass:
 *   fmri_sumsq : *acc += sum x^2 (all-reduce acc across ranks if data parallel)
 *   fmri_renorm: f = 1/sqrt(*sumsq/count); out16 = x*f*scale; *factor_out = (*factor_in)*f */
int fmri_sumsq(const float* x, int64_t n, float* acc, void* stream);
int fmri_renorm(const float* x, void* out16, int64_t n, float scale, const float* sumsq, float count,
                const float* factor_in, float* factor_out, void* stream);
/* The same with the sum of squares in double precision (8-byte aligned device double; zero_first: cleared by the launch
 * itself): the encoder cotangent of `VaeGan.loss`'s KL term (models/vae_gan.py:310) holds 0.5 * (exp(logvar) - 1), whose
 * square leaves fp32 at logvar > 44 while the reference's fp32 step is finite up to logvar 88 -- what the steps use. */
int fmri_sumsq_f64(const float* x, int64_t n, double* acc, int zero_first, void* stream);
int fmri_renorm_f64(const float* x, void* out16, int64_t n, float scale, const double* sumsq, float count,
                    const float* factor_in, float* factor_out, void* stream);

/* ---- optimizers over flat fp32 buffers (train_vgan_stage1.py:275-283; train_wae_stage1.py:221-224) ---
 * g_true = g * gscale / (*gdev) (gdev: device float or NULL), clamped to +-clamp if clamp > 0; the whole
 * update is skipped when flag != NULL and *flag == 0 (device-side equilibrium gate). */
int fmri_rmsprop(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps, float gscale,
                 const float* gdev, float clamp, const int* flag, void* stream);
int fmri_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
              float bc1, float bc2_sqrt, float gscale, const float* gdev, float clamp, const int* flag, void* stream);

/* The same updates with the learning rate (and Adam's step count, incremented with fmri_counter_inc BEFORE the call)
 * read from device memory: lr schedules (ExponentialLR / StepLR, train_vgan_stage1.py:448-450) and Adam's bias
 * correction then need no re-recording of a captured step. */
int fmri_rmsprop_dev(float* p, const float* g, float* sq, int64_t n, const float* lr_dev, float alpha, float eps,
                     float gscale, const float* gdev, float clamp, const int* flag, void* stream);
int fmri_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float b1, float b2,
                  float eps, const int* t_dev, float gscale, const float* gdev, float clamp, const int* flag,
                  void* stream);

/* ---- one launch per sub-network between its weight gradients and its next forward pass (round 4) -----------------
 * What the reference does per parameter between `loss.backward()` and the next `model(x)` -- `.grad` accumulation,
 * `RMSprop.step()` (train_vgan_stage1.py:275-283, 425-432), and what the engine adds around it (slab sum and map of
 * a weight-gradient GEMM's output to the reference layout, the fp16 GEMM copy of the new weights) -- for EVERY
 * parameter of a sub-network in one launch, from a device-resident table:
 *   fmri_apply_entry_fill  fills one host-side row and returns the blocks it occupies (0: this tensor's layout map is
 *       not eligible -- keep fmri_unpack_grad / fmri_rmsprop_dev / fmri_pack_weight for the whole group; < 0: bad
 *       arguments).  flat_n > 0: a flat segment [w, w + flat_n) of 1-D parameters whose gradient `grad` the backward
 *       pass accumulated in place.  Otherwise: `gsrc` = packed gradient
 *       [nslabs][TA*A][ld] as fmri_wgrad wrote it, (sa, sta, sb, stb, A, TA, B, KW, py, px, step, TH, TW) the map of
 *       fmri_unpack_grad, `scale` its factor, `clear` != 0 to write zeros back over gsrc (outputs the weight-gradient
 *       kernel ADDS into), `pk` the fp16 [rows_pad][kpad] GEMM copy in gsrc's orientation or NULL.
 *   fmri_apply_batch  mode 1: w, sq <- RMSprop(w, sq, g * gscale / *gdev) exactly as fmri_rmsprop_dev (same operations,
 *       same order: bit-identical to the separate launches), skipped when *flag == 0; mode 0: gradients only, stored
 *       (not added) into `grad` in the reference layout; mode 2: zero the flat segments' `grad` (the start of a
 *       backward pass, instead of a memset of the whole gradient buffer); mode 3: as mode 1 with every gradient read
 *       from `grad` (reference layout, e.g. mode 0's result summed over the ranks by an all-reduce).
 *       gated != 0: the weight gradients behind gsrc were launched with fmri_wgrad_if under the same `flag` -- when it
 *       is 0 they added nothing and the `clear` pass is skipped as well. */
/* dst[c][r] = src[r][c] (fp16; r < R, c < C; leading dimensions multiples of 8, ld_dst >= R rounded up to 8; src holds
 * src_rows >= R rows, the rows from R on and the columns up to ld_src zero): the data-gradient orientation of a dense
 * layer's fp16 weight made from its forward orientation instead of a second pass over the fp32 master
 * (models/vae_gan.py:81,108,158,200: nn.Linear keeps ONE weight; the second GEMM layout is the engine's). */
int fmri_transpose_f16(const void* src, void* dst, int R, int C, int src_rows, int ld_src, int ld_dst, void* stream);
/* Batched form, one launch per sub-network from a device-resident table (fmri_transpose_entry_fill fills one host-side
 * row and returns the blocks it occupies, < 0 on bad arguments).  A row is one 2-D transpose: a dense weight's second
 * orientation, or ONE TAP of one output-parity class of a stride-2 transposed convolution's weight
 * (nn.Conv2d / nn.ConvTranspose2d keep one weight, models/vae_gan.py:18-20,46-53: the class blocks are the engine's):
 * `src` / `dst` point at the slices (16-byte aligned), `width` = source columns readable from `src` on to its row's end. */
int fmri_transpose_entry_bytes(void);
int fmri_transpose_entry_fill(void* host_entry, const void* src, void* dst, int R, int C, int src_rows, int width,
                              int ld_src, int ld_dst, int tile_begin);
int fmri_transpose_f16_batch(const void* table_dev, int n, int total_tiles, void* stream);
int fmri_apply_entry_bytes(void);
int fmri_apply_entry_fill(void* host_entry, const float* gsrc, float* w, float* sq, float* grad, void* pk, int64_t sa,
                          int64_t sta, int64_t sb, int64_t stb, int A, int TA, int B, int KW, int py, int px, int step,
                          int TH, int TW, int ld, int kpad, int nslabs, int64_t slab_stride, int clear, float scale,
                          int64_t flat_n, int tile_begin);
int fmri_apply_batch(const void* table_dev, int n, int total_tiles, int mode, const float* lr_dev, float alpha, float eps,
                     float gscale, const float* gdev, float clamp, const int* flag, int gated, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FMRI_HIP_H */
