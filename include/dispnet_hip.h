/*
 * dispnet_hip.h -- flat C ABI of libdispnet_hip.so, the MI355X (gfx950) implementation of the
 * zenithfang/supervised_dispnet training hot path (encoder-decoder forward/backward + per-pixel losses).
 *
 * This is the drop-in boundary.  The reference has no native code: every entry
 * point below replaces the ATen/cuDNN op(s) that the cited reference line dispatches.  Host code (Python,
 * supervised_dispnet_amd/) binds these with ctypes; a reference maintainer would bind them the same way
 * (INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP C++ types in signatures (dn_stream_t is a hipStream_t
 *     passed as void*; NULL = the null stream).
 *   - the library never allocates, frees or owns device memory; every buffer (workspace included) belongs to
 *     the caller and is borrowed for the duration of the enqueue.  No internal synchronisation.
 *   - every call returns 0 on success or a negative dn_status; dn_last_error() gives a thread-local message.
 *     Nothing throws across the ABI, nothing aborts.
 *   - re-entrant: no mutable global state (safe from PyTorch's autograd worker threads).
 *   - activations are NHWC fp32 (channels fastest).  Operands carry explicit element strides so NCHW user
 *     tensors (the 3-channel image) can be consumed without a transpose pass.
 *   - arithmetic is fp32 throughout: conv contractions run on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
 */
#ifndef DISPNET_HIP_H_
#define DISPNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dn_stream_t;

enum dn_status {
  DN_OK = 0,
  DN_ERR_BAD_ARG = -1,
  DN_ERR_UNSUPPORTED = -2,
  DN_ERR_WORKSPACE = -3,
  DN_ERR_LAUNCH = -4
};

int dn_version(void);                 /* ABI version, bumped on any signature/struct change */
const char* dn_last_error(void);      /* thread-local, valid until the next failing call on this thread */
int dn_device_arch_ok(void);          /* 1 if the current HIP device is gfx950, 0 otherwise, <0 on error */

/* ------------------------------------------------------------------------------------------------------------
 * Convolution family (implicit GEMM on MFMA).  One descriptor serves
 *   conv forward            nn.Conv2d            models/Disp_vgg_BN.py:40-50,66-70; torchvision vgg16_bn features
 *   conv input-gradient     autograd of the same  train.py:521 (loss.backward())
 *   conv-transpose forward  nn.ConvTranspose2d   models/Disp_vgg_BN.py:53-64; models/DispNetS.py:30-34
 *   conv-transpose input-gradient, and both weight gradients.
 * Fused on the way in : virtual channel-concat of up to 3 operands (torch.cat, Disp_vgg_BN.py:162-185), nearest x2
 *                       upsample of an operand (upsample_nn_nearest, Disp_vgg_BN.py:10-11), BatchNorm-apply + ReLU of
 *                       the producer layer (features[...] BatchNorm2d + ReLU).
 * Fused on the way out: bias, activation (ReLU / LeakyReLU(0.1) / ELU / alpha*sigmoid+beta of predict_disp,
 *                       Disp_vgg_BN.py:66-70,168), per-channel batch-statistic partial sums for the following
 *                       BatchNorm, channel-split of the result into up to 3 destination tensors (backward of cat),
 *                       accumulate-into-destination (skip connections with two consumers).
 * ------------------------------------------------------------------------------------------------------------ */
#define DN_MAX_OPERANDS 3

enum dn_activation {
  DN_ACT_NONE = 0,
  DN_ACT_RELU = 1,
  DN_ACT_LEAKY = 2,        /* p0 = negative slope */
  DN_ACT_ELU = 3,          /* alpha = 1 */
  DN_ACT_SIGMOID_AFFINE = 4 /* p0 * sigmoid(x) + p1 */
};

enum dn_conv_kind {
  DN_CONV_FWD = 0,         /* y = conv(x, w)                 in: x pieces,  out: y            */
  DN_CONV_DGRAD = 1,       /* dx = conv_dgrad(dy, w)         in: dy,        out: dx pieces    */
  DN_CONVT_FWD = 2,        /* y = conv_transpose(x, w)       in: x,         out: y            */
  DN_CONVT_DGRAD = 3       /* dx = conv_transpose_dgrad(dy)  in: dy,        out: dx           */
};

typedef struct dn_operand {        /* a read-only activation operand */
  const float* data;
  int32_t C;                       /* channels of this operand */
  int32_t up_shift;                /* 0, or 1 = stored at half resolution, nearest-x2 upsampled on the fly */
  int64_t stride_n, stride_h, stride_w, stride_c;   /* element strides of the STORED tensor */
  const float* scale;              /* optional [C]: value := max(0, value*scale[c] + shift[c]) on load */
  const float* shift;
} dn_operand;

typedef struct dn_result {         /* a destination tensor (channel stride 1) */
  float* data;
  int32_t C;
  int32_t accumulate;              /* 0: overwrite, 1: += */
  int64_t stride_n, stride_h, stride_w;
} dn_result;

typedef struct dn_conv_desc {
  int32_t kind;                    /* dn_conv_kind */
  int32_t N;
  int32_t IH, IW;                  /* spatial size seen by the taps (logical size of `in`, after up_shift) */
  int32_t OH, OW;                  /* spatial size of `out` (may crop a conv-transpose result, DispNetS.py:37-39) */
  int32_t R, S, stride, pad;       /* of the underlying nn.Conv2d / nn.ConvTranspose2d */
  int32_t n_in;
  dn_operand in[DN_MAX_OPERANDS];
  int32_t n_out;
  dn_result out[DN_MAX_OPERANDS];
  const float* w_packed;           /* from dn_conv_pack_weights for this kind */
  const float* bias;               /* optional [sum of out C] */
  int32_t act;                     /* dn_activation applied after bias */
  float act_p0, act_p1;
  float* bn_partial;               /* optional: [dn_conv_bn_partial_rows()][Cout][2] (sum, sum of squares) of the
                                      PRE-BIAS result per row tile; requires n_out == 1 */
} dn_conv_desc;

/* Elements of the packed weight buffer for desc->kind (depends on R,S,stride,pad, operand/result channels). */
int64_t dn_conv_packed_weight_elems(const dn_conv_desc* d);
/* Re-lay the framework weight tensor for desc->kind.  `w` is nn.Conv2d.weight [Cout][Cin][R][S] for DN_CONV_*,
 * nn.ConvTranspose2d.weight [Cin][Cout][R][S] for DN_CONVT_*.  Run once per optimizer step per kind. */
int dn_conv_pack_weights(const dn_conv_desc* d, const float* w, float* w_packed, dn_stream_t stream);
/* Number of row tiles (first dimension of bn_partial) the launch of this descriptor uses. */
int32_t dn_conv_bn_partial_rows(const dn_conv_desc* d);
/* Enqueue the convolution described by d. */
int dn_conv2d_fwd(const dn_conv_desc* d, dn_stream_t stream);      /* kind == DN_CONV_FWD */
int dn_conv2d_dgrad(const dn_conv_desc* d, dn_stream_t stream);    /* kind == DN_CONV_DGRAD */
int dn_convT2d_fwd(const dn_conv_desc* d, dn_stream_t stream);     /* kind == DN_CONVT_FWD */
int dn_convT2d_dgrad(const dn_conv_desc* d, dn_stream_t stream);   /* kind == DN_CONVT_DGRAD */

/* Weight gradient.  The descriptor is the FORWARD one (DN_CONV_FWD or DN_CONVT_FWD) with `in` = the forward
 * input pieces (same fused load transforms) and w_packed/bias/out ignored.  `dy` is the gradient w.r.t. the
 * pre-activation forward result, NHWC [N][OH][OW][Cout] contiguous.  Writes dw in the framework layout
 * ([Cout][Cin][R][S] / [Cin][Cout][R][S]).  Workspace: dn_conv_wgrad_workspace_bytes(). */
size_t dn_conv_wgrad_workspace_bytes(const dn_conv_desc* fwd);
int dn_conv2d_wgrad(const dn_conv_desc* fwd, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                    dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * BatchNorm (training statistics), ReLU, MaxPool 2x2, activations  --  all NHWC, HBM-bound.
 * Reference: torchvision vgg16_bn features used by models/Disp_vgg_BN.py:137-141.
 * ------------------------------------------------------------------------------------------------------------ */
/* partial (sum,sumsq) [rows][C][2] of the pre-bias conv result -> batch mean / biased var, folded affine
 * (scale = gamma*invstd, shift = beta - mean*scale), running-stat update (momentum, unbiased var), save mean/invstd.
 * count = N*H*W.  training != 0.  */
int dn_bn_finalize(const float* partial, int32_t rows, int32_t C, int64_t count, const float* conv_bias,
                   const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                   float eps, float* mean, float* invstd, float* scale, float* shift, dn_stream_t stream);
/* eval mode: scale/shift from running statistics. */
int dn_bn_eval_affine(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, dn_stream_t stream);
/* p = maxpool2x2(relu(y*scale+shift));  idx (uint8, one per output element): bits0-1 = argmax position in the
 * window (first max in row-major order, like ATen), bit2 = pooled value > 0. */
int dn_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, int32_t N, int32_t H, int32_t W,
                        int32_t C, float* pooled, uint8_t* idx, dn_stream_t stream);
/* dz (full resolution, pre-ReLU gradient; zero where not routed) from dpooled, plus per-channel partial sums
 * [blocks][C][2] of (dz, dz*xhat), xhat = (y-mean)*invstd.  Returns blocks used via dn_reduce_blocks(). */
int dn_bn_relu_pool_bwd(const float* dpooled, const uint8_t* idx, const float* y, const float* mean,
                        const float* invstd, int32_t N, int32_t H, int32_t W, int32_t C, float* dz, float* partial,
                        dn_stream_t stream);
/* dz = da * (y*scale+shift > 0) in place on `da`, plus the same partial sums. */
int dn_bn_relu_bwd_reduce(float* da_dz, const float* y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, int64_t rows, int32_t C, float* partial, dn_stream_t stream);
/* finalize partial sums -> dgamma, dbeta; then dy = gamma*invstd*(dz - dbeta/count - xhat*dgamma/count) in place. */
int dn_bn_bwd_apply(float* dz_dy, const float* y, const float* mean, const float* invstd, const float* gamma,
                    const float* partial, int32_t partial_rows, int64_t rows, int32_t C, float* dgamma, float* dbeta,
                    dn_stream_t stream);
int32_t dn_reduce_blocks(int64_t rows, int32_t C);   /* rows of `partial` the reduce kernels above write */

/* g_pre = g * act'(.) in place, using the stored POST-activation tensor y_post; also per-channel partial sums
 * [blocks][C] of g_pre (the conv bias gradient).  act: RELU / LEAKY(p0) / SIGMOID_AFFINE(p0,p1) / NONE. */
int dn_act_bwd_reduce(float* g, const float* y_post, int32_t act, float p0, float p1, int64_t rows, int32_t C,
                      float* partial, dn_stream_t stream);
/* out[c] = sum over rows of partial[row][c*stride + offset]  (finishes bias / gamma / beta gradients). */
int dn_colsum_finalize(const float* partial, int32_t rows, int32_t C, int32_t stride, int32_t offset, float* out,
                       dn_stream_t stream);
/* dlow[n,h,w] (+)= sum of the 2x2 block of dfull[n,2h..,2w..]   (backward of nearest x2 upsample, C == 1) */
int dn_upsample2x_nearest_bwd(const float* dfull, int32_t N, int32_t h, int32_t w, float* dlow, int32_t accumulate,
                              dn_stream_t stream);
/* bilinear x2, align_corners = False, 1 channel (models/DispNetS.py:120,126,132), cropped to (OH,OW). */
int dn_upsample2x_bilinear_fwd(const float* low, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* out,
                               dn_stream_t stream);
int dn_upsample2x_bilinear_bwd(const float* dout, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* dlow,
                               int32_t accumulate, dn_stream_t stream);
/* y = 1/x (train.py:445); dx = -dy * y*y */
int dn_reciprocal_fwd(const float* x, float* y, int64_t n, dn_stream_t stream);
int dn_reciprocal_bwd(const float* dy, const float* y, float* dx, int64_t n, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses and metrics (loss_functions.py)
 * ------------------------------------------------------------------------------------------------------------ */
enum dn_masked_loss_kind { DN_LOSS_L1 = 0, DN_LOSS_L2 = 1 };
/* l1_loss / l2_loss (loss_functions.py:77-129): per sample mean over valid = 0<gt<max_depth of
 * f(gt - clamp(pred,1e-3,max_depth)), then mean over the batch.  sample_stats: [B][2] (sum, count) scratch.
 * loss: 1 float.  Empty mask -> NaN like the reference. */
int dn_masked_loss_fwd(const float* gt, const float* pred, int32_t B, int64_t pixels, float max_depth, int32_t kind,
                       float* sample_stats, float* loss, dn_stream_t stream);
/* dpred = dloss * f'(.) / (count_b * B) inside the clamp range, 0 elsewhere.  dloss: device scalar. */
int dn_masked_loss_bwd(const float* gt, const float* pred, const float* sample_stats, const float* dloss, int32_t B,
                       int64_t pixels, float max_depth, int32_t kind, float* dpred, dn_stream_t stream);
/* smooth_loss for ONE map [B][H][W] (loss_functions.py:367-386): sum of the 4 second-difference |.|.mean() terms,
 * scaled by `weight`, accumulated into loss[0] (caller zeroes it).  partial: scratch [dn_smooth_blocks][4]. */
int32_t dn_smooth_blocks(int32_t B, int32_t H, int32_t W);
int dn_smooth2_fwd(const float* map, int32_t B, int32_t H, int32_t W, float weight, float* partial, float* loss,
                   dn_stream_t stream);
int dn_smooth2_bwd(const float* map, const float* dloss, int32_t B, int32_t H, int32_t W, float weight, float* dmap,
                   dn_stream_t stream);
/* compute_errors (loss_functions.py:401-448): out[8] = abs_diff, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3.
 * crop rows [y1,y2) cols [x1,x2) (pass 0,H,0,W for none).  scratch: [B][9] floats. */
int dn_compute_errors(const float* gt, const float* pred, int32_t B, int32_t H, int32_t W, float max_depth, int32_t y1,
                      int32_t y2, int32_t x1, int32_t x2, float* scratch, float* out8, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer (train.py:303-305,520-522: torch.optim.Adam, wd = 0) over a flat parameter arena.
 * g is multiplied by grad_scale first (1/world_size folded in).  step >= 1.
 * ------------------------------------------------------------------------------------------------------------ */
int dn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, int32_t step, double grad_scale, dn_stream_t stream);
int dn_fill(float* p, float value, int64_t n, dn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DISPNET_HIP_H_ */
