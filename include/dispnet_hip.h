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
 *   - results are fp32 throughout.  The DEFAULT arithmetic of the matrix-core kernels (dn_conv_desc.compute = 0) forms every fp32
 *     product on the bf16 matrix cores from three exact bf16 pieces per operand (DN_COMPUTE_F32X3: six partial products on
 *     v_mfma_f32_32x32x16_bf16 / _16x16x32_bf16, fp32 accumulation; error against fp64 <= 1.5x an fp32 FMA chain's,
 *     tests/test_gpu_f32x3_fp64.py); kernels without a three-piece form, and every kernel under DN_COMPUTE_F32, run exact fp32 FMA
 *     chains on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32.  The 3x3 / stride 1 / pad 1 layers with aligned channels use
 *     Winograd F(2x2,3x3): fp32 transforms (adds, halves) around those products, error at the level of the direct contraction
 *     (tests/test_gpu_kernels.py).
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

void dn_reload_knobs(void);          /* re-read the DN_* tuning / test switches (they are read once, at first use) */
int dn_version(void);                 /* ABI version, bumped on any signature/struct change */
const char* dn_last_error(void);      /* thread-local, valid until the next failing call on this thread */
const char* dn_last_kernel(void);     /* thread-local: name (as rocprofv3 prints it) of the main kernel the last conv-family call launched */
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
  float* bn_partial;               /* optional: [dn_conv_bn_partial_rows()][Cout][2] = (sum, sum of squared deviations from the
                                      tile mean) of the PRE-BIAS result per 128-row tile; requires n_out == 1 */
  int32_t pad_mode;                /* 0: zero padding; 1: reflection padding (nn.ReflectionPad2d(pad) in front of the conv,
                                      layers.py:124-136).  Reflection is honoured by DN_CONV_FWD and by the weight gradient;
                                      its input gradient = DN_CONV_DGRAD with pad 0 on the padded extent + dn_reflect_fold. */
  int32_t compute;                 /* Arithmetic of the matrix-core kernels that offer a choice (the Winograd, tiled and LDS-resident
                                      convolution kernels); tensors, statistics, transforms and all other kernels are fp32 in every mode.
                                      DN_COMPUTE_DEFAULT (0, what a zeroed descriptor gets): the library's default = DN_COMPUTE_F32X3.
                                      DN_COMPUTE_F32 (3): fp32 FMA chain on v_mfma_f32_32x32x2_f32 everywhere (0.74x the default's speed
                                        on the metric configuration); any unknown value is treated as this one.
                                      DN_COMPUTE_BF16: operands ROUNDED to bf16, fp32 accumulation (v_mfma_f32_32x32x16_bf16) -- the
                                        "mixed precision" mode of BASELINE configs[4]; ~4e-3 relative error per layer.
                                      DN_COMPUTE_F32X3: fp32 products on the bf16 matrix cores: each fp32 operand is split EXACTLY
                                        into three bf16 pieces (x = x0 + x1 + x2) and the six partial products of weight <= 2^-16
                                        (x0y0, x0y1, x1y0, x0y2, x1y1, x2y0) are accumulated in fp32; what is dropped is below
                                        2^-24 of |x||y|, the rounding an fp32 FMA chain commits itself (measured against fp64: the
                                        same error as DN_COMPUTE_F32 or less, tools/ubench/bf16x3.hip, tests/test_gpu_kernels.py). */
  int32_t dilation;                /* 0 / 1: none.  d > 1: nn.Conv2d(dilation = d) -- kernel tap (r, s) reads input offset
                                      (r*d - pad, s*d - pad) (models/ASPP.py:62-72,107-113: the dilated bottlenecks and the ASPP
                                      classifier).  Stride-1 DN_CONV_FWD / DN_CONV_DGRAD and their weight gradient, zero padding. */
  /* Optional, DN_CONV_DGRAD only (round 3): the column sums of the BatchNorm backward of the layer BELOW, taken in this call's epilogue.
   * The input gradient dx written here is dL/d(relu(bn(y))) of the producer of this layer's input; with bnb_y .. bnb_invstd set ([pixels][C]
   * dense like the result, [C] vectors) every 128-pixel tile also writes bnb_partial[tile][C][2] = (sum dz, sum dz * xhat),
   * dz = dx * [y * scale + shift > 0], xhat = (y - mean) * invstd -- exactly what dn_bn_relu_bwd_sums computes in a separate pass over dx
   * and y (one full read of dx and one launch saved per layer); rows = dn_conv_bn_partial_rows().  Honoured only where
   * dn_conv_dgrad_fuses_bn_sums() returns 1 (the Winograd input-gradient kernels, single non-accumulating result); otherwise ignored. */
  const float* bnb_y;
  const float* bnb_scale;
  const float* bnb_shift;
  const float* bnb_mean;
  const float* bnb_invstd;
  float* bnb_partial;
  /* Optional (round 3): workspace for the K split of small grids (Winograd input channels; chunks of the three-piece direct kernel).  A 4-image shard of the metric's batch leaves the
   * deep layers 32-104 blocks for 256 CUs, each walking all 32-48 chunks of the K axis alone; with a workspace of at least
   * dn_conv_splitk_workspace_bytes(desc) bytes the three-piece Winograd forward / input gradient splits K over 2-8 blocks per tile; the
   * partial output tiles meet here and the block that arrives last sums them in index order (deterministic) and runs the epilogue.
   * The first 4096 bytes are int counters: they must be ZERO before the first use and are left zero by every launch (self-resetting), so
   * one buffer serves every launch of ONE stream; launches on different streams need different buffers.  NULL: never split. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  /* Optional (round 4): second result of a one-channel disparity head.  The reference computes depth = 1 / disp in
   * the caller, right behind the network (train.py:445, `depth = [1/disp for disp in disparities]`); when dn_conv_fwd_fuses_reciprocal(desc)
   * returns 1 the head kernel that writes disp = alpha * sigmoid(conv) + beta also writes 1 / disp here (dense [N][OH][OW] floats, the
   * same IEEE division dn_reciprocal_fwd performs), so the caller's reciprocal needs no launch.  Otherwise ignored; NULL: not wanted. */
  float* recip_out;
  /* Optional (round 5), DN_CONV_FWD with bn_partial: finish the BatchNorm batch statistics INSIDE this launch.  The block that arrives
   * last for a 64-channel slice of the result merges the slice's bn_partial rows and writes what dn_bn_finalize would have written
   * (mean, invstd, the folded (scale, shift), the running statistics, the step counter) -- same arithmetic, same order, same bits
   * (dn_bn_finalize takes the same sliced order for up to 128 partial rows); the conv bias is this descriptor's `bias`.  Honoured only
   * where dn_conv_fwd_folds_bn_finalize(desc) returns 1 (three-piece Winograd kernels, <= 128 partial rows, a splitk_ws whose last 256
   * bytes serve as self-resetting counters); otherwise ignored and the caller runs dn_bn_finalize.  bnf_scale == NULL: not wanted. */
  const float* bnf_gamma;
  const float* bnf_beta;
  float* bnf_running_mean;         /* may be NULL together with bnf_running_var (untracked statistics) */
  float* bnf_running_var;
  int64_t* bnf_num_batches_tracked;   /* may be NULL */
  float bnf_momentum, bnf_eps;
  float* bnf_mean;
  float* bnf_invstd;
  float* bnf_scale;
  float* bnf_shift;
  /* Optional (round 5), DN_CONV_DGRAD with bnb_*: also FINISH the two BatchNorm-backward sums of the layer below inside this launch:
   * bnb_dbeta[c] = sum dz, bnb_dgamma[c] = sum dz * xhat -- the parameter gradients dn_bn_bwd_apply_relu derives from bnb_partial with a
   * launch of its own (pass partial = NULL there: the sums are final).  Same sliced order as that launch for <= 128 rows.  Honoured
   * only where dn_conv_dgrad_folds_bn_sums(desc) returns 1. */
  float* bnb_dgamma;
  float* bnb_dbeta;
} dn_conv_desc;

enum { DN_COMPUTE_DEFAULT = 0, DN_COMPUTE_BF16 = 1, DN_COMPUTE_F32X3 = 2, DN_COMPUTE_F32 = 3 };

/* Elements of the packed weight buffer for desc->kind (depends on R,S,stride,pad, operand/result channels). */
int64_t dn_conv_packed_weight_elems(const dn_conv_desc* d);
/* Re-lay the framework weight tensor for desc->kind.  `w` is nn.Conv2d.weight [Cout][Cin][R][S] for DN_CONV_*,
 * nn.ConvTranspose2d.weight [Cin][Cout][R][S] for DN_CONVT_*.  Run once per optimizer step per kind. */
int dn_conv_pack_weights(const dn_conv_desc* d, const float* w, float* w_packed, dn_stream_t stream);
/* Every weight re-lay of a training step in (at most) two launches.  The caller keeps a table of dn_pack_entry_bytes()-sized rows:
 * dn_pack_entry_fill() writes one row on the HOST for (descriptor, framework weights, packed destination) and returns 1 for a
 * Winograd-layout row, 2 for a bf16 Winograd row, 3 for a three-piece bf16 Winograd row, 0 for a direct-layout row (<0: error); the
 * caller orders the rows direct, Winograd, bf16, three-piece, copies the table to the device once, and calls dn_pack_many(device table,
 * #direct, #winograd, #bf16, #three-piece, stream) after every optimizer step (same arithmetic as
 * dn_conv_pack_weights per row; the pointers in the rows must still be valid). */
int64_t dn_pack_entry_bytes(void);
int dn_pack_entry_fill(const dn_conv_desc* d, const float* w, float* w_packed, void* entry_host);
int dn_pack_many(const void* entries_dev, int32_t n_direct, int32_t n_wino, int32_t n_wino_bf16, int32_t n_wino_x3, dn_stream_t stream);
/* Which packed layout dn_conv_pack_weights produces for this descriptor: 0 = implicit-GEMM [phase][Npad][K chunks], 2 = the Winograd
 * layout below rounded to bf16 (descriptor compute = DN_COMPUTE_BF16), 3 = the same as three exact bf16 pieces (DN_COMPUTE_F32X3),
 * 1 = Winograd
 * F(2x2,3x3) transformed weights in MFMA fragment order (3x3 / stride 1 / pad 1 layers with 16-aligned channels and even
 * extents that fill the kernel's tiles).  The layout depends on the geometry, not only on the weights: a caller that caches
 * packed weights must key the cache on it (the same layer at another resolution may pick the other algorithm). */
int32_t dn_conv_weight_layout(const dn_conv_desc* d);
/* Number of row tiles (first dimension of bn_partial) the launch of this descriptor uses. */
int32_t dn_conv_bn_partial_rows(const dn_conv_desc* d);
/* Bytes of dn_conv_desc.splitk_ws this call would use at most (0: the call does not split; < 0: bad descriptor).  Two kernel families
 * split: the three-piece Winograd forward / input gradient (input channels) and the three-piece direct kernel with one scheduled operand
 * (chunks of the K axis: the 4x13 / 8x26 transposed convolutions of the decoder). */
int64_t dn_conv_splitk_workspace_bytes(const dn_conv_desc* d);
/* 1 when dn_conv2d_fwd(desc) will finish the BatchNorm statistics itself (dn_conv_desc.bnf_*), 0 when the caller has to run
 * dn_bn_finalize; likewise for the input gradient and the BatchNorm-backward sums (dn_conv_desc.bnb_dgamma / bnb_dbeta). */
int32_t dn_conv_fwd_folds_bn_finalize(const dn_conv_desc* d);
int32_t dn_conv_dgrad_folds_bn_sums(const dn_conv_desc* d);
/* 1 if dn_conv2d_dgrad(d) will write bnb_partial (see dn_conv_desc), 0 if it ignores the bnb_* fields, < 0 on a bad descriptor. */
int32_t dn_conv_dgrad_fuses_bn_sums(const dn_conv_desc* d);
/* 1 if dn_conv2d_fwd(d) writes 1 / out[0] to d->recip_out (a one-channel head on the second-generation head kernel), else 0. */
int32_t dn_conv_fwd_fuses_reciprocal(const dn_conv_desc* d);
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
/* partial (sum, M2 about the tile mean) [rows][C][2] of the pre-bias conv result (128-row tiles, merged with the
 * parallel-variance update in fp64) -> batch mean / biased var, folded affine
 * (scale = gamma*invstd, shift = beta - mean*scale), running-stat update (momentum, unbiased var), save mean/invstd.
 * count = N*H*W.  training != 0.  num_batches_tracked (nullable): nn.BatchNorm2d's int64 step counter, incremented by one. */
int dn_bn_finalize(const float* partial, int32_t rows, int32_t C, int64_t count, const float* conv_bias,
                   const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                   float eps, float* mean, float* invstd, float* scale, float* shift, int64_t* num_batches_tracked,
                   dn_stream_t stream);
/* eval mode: scale/shift from running statistics. */
int dn_bn_eval_affine(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, dn_stream_t stream);
/* p = maxpool2x2(relu(y*scale+shift));  idx (uint8, one per output element): bits0-1 = argmax position in the
 * window (first max in row-major order, like ATen), bit2 = pooled value > 0.  scale == shift == NULL: y is a plain tensor
 * (already activated), p = maxpool2x2(y) -- the BatchNorm-free VGG encoders (reference models/Disp_vgg.py:79-100). */
int dn_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, int32_t N, int32_t H, int32_t W,
                        int32_t C, float* pooled, uint8_t* idx, dn_stream_t stream);
/* dz (full resolution, pre-ReLU gradient; zero where not routed) from dpooled, plus per-channel partial sums
 * [blocks][C][2] of (dz, dz*xhat), xhat = (y-mean)*invstd.  Returns blocks used via dn_reduce_blocks(). */
int dn_bn_relu_pool_bwd(const float* dpooled, const uint8_t* idx, const float* y, const float* mean,
                        const float* invstd, int32_t N, int32_t H, int32_t W, int32_t C, float* dz, float* partial,
                        dn_stream_t stream);
/* gradient of the plain 2x2 max-pool (dn_bn_relu_pool_fwd with NULL scale): dx[window arg-max] (+)= dpooled, 0 elsewhere. */
int dn_maxpool2_bwd(const float* dpooled, const uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, float* dx, int32_t accumulate,
                    dn_stream_t stream);
/* dz = da * (y*scale+shift > 0) in place on `da`, plus the same partial sums. */
int dn_bn_relu_bwd_reduce(float* da_dz, const float* y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, int64_t rows, int32_t C, float* partial, dn_stream_t stream);
/* finalize partial sums -> dgamma, dbeta; then dy = gamma*invstd*(dz - dbeta/count - xhat*dgamma/count) in place.
 * partial: [partial_rows][C][partial_stride] with (sum dz, sum dz*xhat) at offsets partial_offset, partial_offset+1
 * (stride 2 / offset 0 for the reduce kernels above; stride 4 / offset 0 or 2 for dn_bn_add_relu_bwd). */
int dn_bn_bwd_apply(float* dz_dy, const float* y, const float* mean, const float* invstd, const float* gamma,
                    const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset, int64_t rows,
                    int32_t C, float* dgamma, float* dbeta, dn_stream_t stream);
/* The two passes above without a materialised dz: the *_sums calls only produce the partial sums (no full-size write), and the
 * apply_* calls re-derive dz on the fly -- from da and the ReLU mask (y*scale+shift > 0; `da_dy` is overwritten with dy), or from
 * the pooled gradient and the arg-max codes (`dy` [N,H,W,C] is written; the sparse full-resolution dz of a max-pool never exists). */
int dn_bn_relu_bwd_sums(const float* da, const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                        int64_t rows, int32_t C, float* partial, dn_stream_t stream);
int dn_bn_relu_pool_bwd_sums(const float* dpooled, const uint8_t* idx, const float* y, const float* mean, const float* invstd, int32_t N,
                             int32_t H, int32_t W, int32_t C, float* partial, dn_stream_t stream);
int dn_bn_bwd_apply_relu(float* da_dy, const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* gamma, const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset,
                         int64_t rows, int32_t C, float* dgamma, float* dbeta, dn_stream_t stream);
int dn_bn_bwd_apply_pool(const float* dpooled, const uint8_t* idx, const float* y, const float* mean, const float* invstd, const float* gamma,
                         const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset, int32_t N, int32_t H,
                         int32_t W, int32_t C, float* dy, float* dgamma, float* dbeta, dn_stream_t stream);
int32_t dn_reduce_blocks(int64_t rows, int32_t C);   /* rows of `partial` the reduce kernels above write */

/* g_pre = g * act'(.) in place, using the stored POST-activation tensor y_post; also per-channel partial sums
 * [blocks][C] of g_pre (the conv bias gradient).  act: RELU / LEAKY(p0) / SIGMOID_AFFINE(p0,p1) / NONE. */
int dn_act_bwd_reduce_from(const float* g_in, float* g_out, const float* y_post, int32_t act, float p0, float p1, int64_t rows, int32_t C,
                           float* partial, dn_stream_t stream);   /* the same, out of place (g_in is not written) */
int dn_act_bwd_reduce(float* g, const float* y_post, int32_t act, float p0, float p1, int64_t rows, int32_t C,
                      float* partial, dn_stream_t stream);
/* out[c] = sum over rows of partial[row][c*stride + offset]  (finishes bias / gamma / beta gradients). */
int dn_colsum_finalize(const float* partial, int32_t rows, int32_t C, int32_t stride, int32_t offset, float* out,
                       dn_stream_t stream);
/* dlow[n,h,w] (+)= sum of the 2x2 block of dfull[n,2h..,2w..]   (backward of nearest x2 upsample, C == 1) */
int dn_upsample2x_nearest_bwd(const float* dfull, int32_t N, int32_t h, int32_t w, float* dlow, int32_t accumulate,
                              dn_stream_t stream);
/* the same for an NHWC tensor with C channels (layers.upsample, layers.py:193-196; C % 4 == 0 or C == 1) */
int dn_upsample2x_nearest_bwd_nhwc(const float* dfull, int32_t N, int32_t h, int32_t w, int32_t C, float* dlow, int32_t accumulate,
                                   dn_stream_t stream);
/* backward of ReflectionPad2d(pad): dxp is the gradient on the padded extent [N][H+2p][W+2p][C]; dx[N][H][W][C] (+)= the
 * sum over the padded positions that mirror onto each pixel. */
int dn_reflect_fold(const float* dxp, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, float* dx, int32_t accumulate,
                    dn_stream_t stream);
/* out = relu( (y*scale + shift) + (r_scale ? r*r_scale + r_shift : r) )  -- the tail of a ResNet bottleneck:
 * bn3(conv3) + identity / downsample-BN, then ReLU (models/Disp_res_50.py:229-247).  NHWC, C % 4 == 0.
 * r == NULL: no residual, i.e. a materialised relu(bn(y)) (a BatchNorm+ReLU output that leaves the engine as a tensor). */
int dn_bn_add_relu_fwd(const float* y, const float* scale, const float* shift, const float* r, const float* r_scale, const float* r_shift,
                       int64_t rows, int32_t C, float* out, dn_stream_t stream);
/* m = gout * (out > 0).  dz_y = m; partial [dn_reduce_blocks(rows,C)][C][4] = sums of (m, m*xhat_y, m, m*xhat_r).
 * Residual branch: with r_mean/r_invstd (a downsample BatchNorm) dr = m (overwrite); without, dr (+)= m (dr may be NULL). */
int dn_bn_add_relu_bwd(const float* gout, const float* out, const float* y, const float* mean, const float* invstd, const float* r,
                       const float* r_mean, const float* r_invstd, int64_t rows, int32_t C, float* dz_y, float* dr, int32_t dr_accumulate,
                       float* partial, dn_stream_t stream);
/* MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (models/Disp_res_50.py:73); idx (uint8 per output element) = window
 * position 0..8 of the first maximum in row-major order.  Backward gathers (deterministic), dx overwrite / accumulate. */
/* ceil_mode != 0: nn.MaxPool2d(..., ceil_mode=True) (models/ASPP.py:138): OH = ceil((H - 1) / 2) + 1 unless that window starts past the
 * input; the caller sizes out / idx with dn_maxpool3s2_out(H, ceil_mode). */
int32_t dn_maxpool3s2_out(int32_t H, int32_t ceil_mode);
int dn_maxpool3s2_fwd(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ceil_mode, float* out, uint8_t* idx, dn_stream_t stream);
int dn_maxpool3s2_bwd(const float* dout, const uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ceil_mode, float* dx,
                      int32_t accumulate, dn_stream_t stream);
/* out = (x - sub) / div  element-wise (the input normalisation of networks/vgg_encoder.py:80, resnet_encoder.py:89) */
int dn_sub_div(const float* x, int64_t n, float sub, float div, float* out, dn_stream_t stream);
/* bilinear x2, align_corners = False, 1 channel (models/DispNetS.py:120,126,132), cropped to (OH,OW). */
int dn_upsample2x_bilinear_fwd(const float* low, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* out,
                               dn_stream_t stream);
int dn_upsample2x_bilinear_bwd(const float* dout, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* dlow,
                               int32_t accumulate, dn_stream_t stream);
/* ---- pieces of the FCRN / ASPP nets (reference models/FCRN.py, models/ASPP.py, models/res_aspp.py) ----
 * Batch statistics of a MATERIALISED NHWC tensor in the layout the conv epilogues write and dn_bn_finalize merges:
 * partial[dn_bn_stats_rows(rows)][C][2] = (sum, sum of squared deviations from the tile mean) per 128-row tile.  FCRN's up-projection
 * normalises a map interleaved from four convolutions (models/FCRN.py:97-113), so no single conv epilogue sees its statistics. */
int32_t dn_bn_stats_rows(int64_t rows);
int dn_bn_stats_partial(const float* x, int64_t rows, int32_t C, float* partial, dn_stream_t stream);
/* out = y*scale[c] + shift[c]: a BatchNorm output WITHOUT ReLU that leaves as a tensor (FCRN bn2, models/FCRN.py:236-237). */
int dn_bn_apply_fwd(const float* y, const float* scale, const float* shift, int64_t rows, int32_t C, float* out, dn_stream_t stream);
/* FCRN's up-projection (models/FCRN.py:74-113) interleaves four convolutions -- 3x3, 2x3, 3x2, 2x2, each with one zero row above and one
 * zero column to the left -- into a 2H x 2W map: phase (a, b) = (row parity, column parity).  That is exactly a ConvTranspose2d(6x6,
 * stride 2, padding 2) whose weight is assembled from the four (wt[ci][co][a + 4 - 2 kr][b + 4 - 2 ks] = w_ab[co][ci][kr][ks], zero
 * elsewhere), so the engine runs DN_CONVT_* on that composite weight; what the transposed convolution cannot carry are the four
 * separate biases:  x[n][2y+a][2x+b][c] += bias4[a][b][c]  and, backwards,  out4[a][b][c] = sum of g over phase (a, b) (fixed-order sums;
 * workspace of dn_phase_colsum_workspace_bytes(C)). */
int dn_phase_bias_add(float* x, int32_t N, int32_t H2, int32_t W2, int32_t C, const float* bias4, dn_stream_t stream);
size_t dn_phase_colsum_workspace_bytes(int32_t C);
int dn_phase_colsum(const float* g, int32_t N, int32_t H, int32_t W, int32_t C, float* workspace, float* out4, dn_stream_t stream);
/* out = act(x) element-wise (the ASPP classifier sums four dilated convolutions BEFORE its sigmoid, models/ASPP.py:117-123). */
int dn_act_fwd(const float* x, int64_t n, int32_t act, float p0, float p1, float* out, dn_stream_t stream);
/* F.interpolate(x, size=(OH, OW), mode='bilinear', align_corners=...) of a one-channel map [N][IH][IW] (models/FCRN.py:253,
 * models/ASPP.py:192); source index and weights computed in fp32 exactly as ATen does.  Backward gathers (deterministic). */
int dn_resize_bilinear_fwd(const float* in, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t align_corners, float* out,
                           dn_stream_t stream);
int dn_resize_bilinear_bwd(const float* dout, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t align_corners, float* din,
                           int32_t accumulate, dn_stream_t stream);
/* out[n][c] = scale * mean over pixels of x[n][p][c] (NHWC): pose = 0.01 * pose_pred(...).mean(3).mean(2), models/PoseExpNet.py:73-75 */
int dn_spatial_mean_fwd(const float* x, int32_t N, int64_t HW, int32_t C, float scale, float* out, dn_stream_t stream);
int dn_spatial_mean_bwd(const float* dout, int32_t N, int64_t HW, int32_t C, float scale, float* dx, dn_stream_t stream);
/* y = 1/x (train.py:445); dx = -dy * y*y */
int dn_reciprocal_fwd(const float* x, float* y, int64_t n, dn_stream_t stream);
int dn_reciprocal_bwd(const float* dy, const float* y, float* dx, int64_t n, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses and metrics (loss_functions.py)
 * ------------------------------------------------------------------------------------------------------------ */
enum dn_masked_loss_kind { DN_LOSS_L1 = 0, DN_LOSS_L2 = 1, DN_LOSS_BERHU = 2, DN_LOSS_SCALE_INV = 3 };
#define DN_LOSS_STATS 8   /* floats of per-group statistics kept between forward and backward */
/* The masked depth-loss family: l1_loss / l2_loss / berhu_loss / Scale_invariant_loss (loss_functions.py:77-189, one
 * group per SAMPLE: G = B, pixels = H*W) and Multiscale_{L1,FULL_L1,L2,berhu,scale_inv}_loss (:217-315, one group per
 * SCALE over the whole batch: G = 1, pixels = B*h*w, weight = 1/2^i, accumulate = (i > 0)).
 * Per group: mean over valid = 0 < gt < max_depth of f(gt, clamp(pred, 1e-3, max_depth)); then
 *   loss[0] = (accumulate ? loss[0] : 0) + weight * (1/G) * sum_g L_g.   Empty mask -> NaN like the reference.
 * stats: [G][DN_LOSS_STATS] (kept for the backward); workspace: dn_masked_loss_workspace_bytes(). */
size_t dn_masked_loss_workspace_bytes(int32_t G, int64_t pixels);
int dn_masked_loss_fwd(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, float weight,
                       int32_t accumulate, float* stats, void* workspace, size_t workspace_bytes, float* loss, dn_stream_t stream);
/* The same forward as ONE launch (kinds L1 / L2 / scale-invariant, G <= 256): the block that arrives last reduces the split partials and
 * turns them into the loss, in the order of the separate launches -- bit-identical to dn_masked_loss_fwd.  `counter`: one int32 in device
 * memory, zero before the first call and left zero (calls that may overlap on the device need counters of their own). */
int dn_masked_loss_fwd_fused(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, float weight,
                             int32_t accumulate, float* stats, void* workspace, size_t workspace_bytes, float* loss, int32_t* counter,
                             dn_stream_t stream);
/* dn_masked_loss_fwd in pieces, for one-process-per-GPU runs of the whole-batch losses (the reference's DataParallel sees the
 * gathered batch on GPU0, loss_functions.py:232-237): pass 0 -> stats[g][0..3] = (sum f, n, sum d, max r) of THIS rank's pixels;
 * the caller all-reduces them (columns 0..2 add, column 3 takes the max); kind berHu then runs pass 1 -> stats[g][0], [2], [4]
 * (sums that depend on the global max; they add across ranks); dn_masked_loss_finalize turns the stats into the loss. */
int dn_masked_loss_stats(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, int32_t pass,
                         float* stats, void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_masked_loss_finalize(float* stats, int32_t G, int32_t kind, float weight, int32_t accumulate, float* loss, dn_stream_t stream);
/* dpred = dloss * weight/G * dL_g/dpred  (0 outside the mask / clamp range).  berHu includes the gradient through its
 * max-residual threshold, like torch autograd.  dloss: device scalar. */
int dn_masked_loss_bwd(const float* gt, const float* pred, const float* stats, const float* dloss, int32_t G, int64_t pixels,
                       float max_depth, int32_t kind, float weight, float* dpred, dn_stream_t stream);
/* One level of generate_{max,avg,bilinear}_pyramid (loss_functions.py:191-215): [N][H][W] -> [N][H/2][W/2];
 * mode 0 max_pool2d(2,2), 1 avg_pool2d(2,2), 2 F.interpolate(scale_factor=0.5, bilinear, align_corners=False). */
int dn_pyramid_down2(const float* in, int32_t N, int32_t H, int32_t W, int32_t mode, float* out, dn_stream_t stream);
/* F.upsample(x, scale_factor=s, mode) of a 1-channel map (Multiscale_FULL_L1_loss, loss_functions.py:243):
 * mode 0 nearest, 1 bilinear (align_corners=False).  [N][h][w] -> [N][h*s][w*s]. */
int dn_upsample_int_fwd(const float* low, int32_t N, int32_t h, int32_t w, int32_t scale, int32_t mode, float* out, dn_stream_t stream);
int dn_upsample_int_bwd(const float* dout, int32_t N, int32_t h, int32_t w, int32_t scale, int32_t mode, float* dlow, dn_stream_t stream);
/* explainability_loss (loss_functions.py:357-364) for one mask tensor of n elements: binary_cross_entropy(mask, 1).
 * partial: [dn_reduce1d_blocks(n)] scratch. */
int32_t dn_reduce1d_blocks(int64_t n);
int dn_explainability_fwd(const float* mask, int64_t n, float weight, int32_t accumulate, float* partial, float* loss, dn_stream_t stream);
int dn_explainability_bwd(const float* mask, const float* dloss, int64_t n, float* dmask, dn_stream_t stream);
/* smooth_loss for ONE map [B][H][W] (loss_functions.py:367-386): sum of the 4 second-difference |.|.mean() terms,
 * scaled by `weight`, accumulated into loss[0] (caller zeroes it).  partial: scratch [dn_smooth_blocks][4]. */
int32_t dn_smooth_blocks(int32_t B, int32_t H, int32_t W);
int dn_smooth2_fwd(const float* map, int32_t B, int32_t H, int32_t W, float weight, float* partial, float* loss,
                   dn_stream_t stream);
int dn_smooth2_bwd(const float* map, const float* dloss, int32_t B, int32_t H, int32_t W, float weight, float* dmap,
                   dn_stream_t stream);
/* compute_errors (loss_functions.py:401-448): out[8] = abs_diff, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3.
 * crop rows [y1,y2) cols [x1,x2) (pass 0,H,0,W for none).  median_scaling != 0: valid_pred *= median(gt)/median(pred)
 * per sample (:432-433, torch.median = lower middle; exact radix select).  scratch: [B][11] floats. */
int dn_compute_errors(const float* gt, const float* pred, int32_t B, int32_t H, int32_t W, float max_depth, int32_t y1,
                      int32_t y2, int32_t x1, int32_t x2, int32_t median_scaling, float* scratch, float* out8,
                      dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Geometry / photometric family (inverse_warp.py, loss_functions.py:317-354, layers.py:199-245)
 * Images are planar [B][3][h][w]; depth [B][h][w]; rotation_mode 0 euler / 1 quat; padding_mode 0 zeros / 1 border;
 * align_corners is F.grid_sample's flag (the reference passes none: False on torch >= 1.3, True on its pinned 1.0.1).
 * ------------------------------------------------------------------------------------------------------------ */
/* pose_vec2mat + `intrinsics @ pose_mat` (inverse_warp.py:141-157,185-188) with the per-scale intrinsics rescaling of
 * loss_functions.py:328-329 folded in (downscale = 1 for plain inverse_warp):
 *   proj[b] (3x4) = K_s @ [R(pose)|t],  kinv_scaled[b] (3x3) = Kinv with columns 0,1 * downscale.
 * pose element (b, j) at pose[b*pose_stride_b + j], j = tx,ty,tz,rx,ry,rz. */
int dn_pose_proj_fwd(const float* pose, int64_t pose_stride_b, const float* K, const float* Kinv, int32_t B, int32_t rotation_mode,
                     float downscale, float* proj, float* kinv_scaled, dn_stream_t stream);
/* dpose from the per-block partial sums of dproj ([B][nblk][12], nblk = dn_warp_blocks(h,w)) written by the *_bwd below. */
int32_t dn_warp_blocks(int32_t h, int32_t w);
int dn_pose_proj_bwd(const float* pose, int64_t pose_stride_b, const float* K, int32_t B, int32_t rotation_mode, float downscale,
                     const float* dproj_partial, int32_t nblk, float* dpose, int64_t dpose_stride_b, int32_t accumulate,
                     dn_stream_t stream);
/* inverse_warp (inverse_warp.py:160-193): pixel2cam, projection, cam2pixel (zeros padding: out-of-range coordinates := 2,
 * no gradient), bilinear grid_sample.  warped: [B][3][h][w]. */
int dn_inverse_warp_fwd(const float* img, const float* depth, const float* proj, const float* kinv, int32_t B, int32_t h, int32_t w,
                        int32_t padding_mode, int32_t align_corners, float* warped, dn_stream_t stream);
/* gradients w.r.t. depth ([B][h][w], overwrite / accumulate) and the projection (partials, see dn_pose_proj_bwd). */
int dn_inverse_warp_bwd(const float* img, const float* depth, const float* proj, const float* kinv, int32_t B, int32_t h, int32_t w,
                        int32_t padding_mode, int32_t align_corners, const float* dwarped, float* ddepth, int32_t accumulate_depth,
                        float* dproj_partial, dn_stream_t stream);
/* One (scale, reference image) term of photometric_reconstruction_loss (loss_functions.py:331-342), fused: warp `ref`,
 * out-of-bound mask = 1 - prod_c(warped_c == 0), diff = (tgt - warped) * oob [* mask], loss (+)= weight * mean|diff|.
 * mask: explainability mask element (b,y,x) at mask[b*mask_stride_b + y*w + x], or NULL.
 * partial: [B * dn_warp_blocks(h,w)] scratch. */
int dn_photometric_fwd(const float* tgt, const float* ref, const float* depth, const float* proj, const float* kinv, const float* mask,
                       int64_t mask_stride_b, int32_t B, int32_t h, int32_t w, int32_t padding_mode, int32_t align_corners, float weight,
                       int32_t accumulate, float* partial, float* loss, dn_stream_t stream);
int dn_photometric_bwd(const float* tgt, const float* ref, const float* depth, const float* proj, const float* kinv, const float* mask,
                       int64_t mask_stride_b, int32_t B, int32_t h, int32_t w, int32_t padding_mode, int32_t align_corners, float weight,
                       const float* dloss, float* ddepth, int32_t accumulate_depth, float* dproj_partial, float* dmask,
                       int64_t dmask_stride_b, dn_stream_t stream);
/* F.interpolate(x, (H/f, W/f), mode='area') for an exact integer factor (loss_functions.py:326-327): f x f mean. */
int dn_area_down(const float* in, int64_t planes, int32_t H, int32_t W, int32_t factor, float* out, dn_stream_t stream);
/* layers.SSIM.forward (layers.py:215-245) on [planes][H][W]; backward workspace: 5*planes*H*W floats; dx / dy nullable. */
int dn_ssim_fwd(const float* x, const float* y, int64_t planes, int32_t H, int32_t W, float* out, dn_stream_t stream);
int dn_ssim_bwd(const float* x, const float* y, const float* dout, int64_t planes, int32_t H, int32_t W, float* workspace, float* dx,
                float* dy, dn_stream_t stream);
/* layers.get_smooth_loss (layers.py:199-212): disp [B][1][H][W], img [B][C][H][W]; partial: [dn_edge_smooth_blocks][2]. */
int32_t dn_edge_smooth_blocks(int32_t B, int32_t H, int32_t W);
int dn_edge_smooth_fwd(const float* disp, const float* img, int32_t B, int32_t C, int32_t H, int32_t W, float* partial, float* loss,
                       dn_stream_t stream);
int dn_edge_smooth_bwd(const float* disp, const float* img, const float* dloss, int32_t B, int32_t C, int32_t H, int32_t W, float* ddisp,
                       dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * DORN ordinal head and loss (models/Disp_vgg_BN_DORN.py:196-227, loss_functions.py:16-74, utils.py:106-175)
 * ------------------------------------------------------------------------------------------------------------ */
/* OrdinalRegressionLayer: pre = logits, element (n, pixel, channel) at pre[n*stride_n + pixel*stride_pix + channel*stride_c]
 * (NHWC: 2K*HW, 2K, 1; NCHW: 2K*HW, 1, HW), channel pairs (A,B) = (2k, 2k+1); ord = planar [N][K][HW] =
 * softmax(clamp(A), clamp(B))[1]; decode = [N][HW] int64 count of ord > 0.5.  dpre has the layout of pre. */
int dn_ordinal_fwd(const float* pre, int64_t stride_n, int64_t stride_pix, int64_t stride_c, int32_t N, int64_t HW, int32_t K, float* ord,
                   int64_t* decode, dn_stream_t stream);
int dn_ordinal_bwd(const float* pre, int64_t stride_n, int64_t stride_pix, int64_t stride_c, const float* ord, const float* dord, int32_t N,
                   int64_t HW, int32_t K, float* dpre, dn_stream_t stream);
/* DORN_loss: target = SID labels int32 [N][HW]; stats[2] = (sum, num_valid) kept for the backward; loss = sum / -num_valid.
 * One process per GPU (the reference's DataParallel sees the gathered batch on GPU0 and divides by ITS valid count,
 * loss_functions.py:69-73): the caller sums `stats` over the ranks between dn_ordinal_loss_fwd and dn_ordinal_loss_finalize
 * (loss = stats[0] / -stats[1] again, from the exchanged pair) and passes grad_scale = world size to the backward, whose 1/num_valid
 * is then the whole batch's; grad_scale = 1 otherwise. */
int32_t dn_ordinal_loss_blocks(int32_t N, int64_t HW);
int dn_ordinal_loss_fwd(const float* ord, const float* gt, const int32_t* target, int32_t N, int64_t HW, int32_t K, float max_depth,
                        float* partial, float* stats, float* loss, dn_stream_t stream);
int dn_ordinal_loss_finalize(const float* stats, float* loss, dn_stream_t stream);
int dn_ordinal_loss_bwd(const float* ord, const float* gt, const int32_t* target, const float* stats, const float* dloss, int32_t N,
                        int64_t HW, int32_t K, float max_depth, float grad_scale, float* dord, dn_stream_t stream);
/* get_labels_sid / get_depth_sid (beta = 80.999 kitti, 10.999 nyu). */
/* Fused head (models/Disp_vgg_BN_DORN.py:112-114,191-227): Dropout2d channel mask (mask[N][16] = 0 or 1/(1-p), NULL = none) ->
 * 1x1 convolution 16 -> 2K (w = conv_ord.weight [2K][16], bias [2K]) -> clamp -> pair softmax.  The 2K-channel logits never reach
 * HBM: forward writes ord [N][K][HW] (planar, the reference's NCHW) and decode [N][HW] int64; backward takes dord (planar) and
 * recomputes the logits, producing dx [N][HW][16] (NHWC, accumulate != 0 adds), dw [2K][16] and dbias [2K] (fixed-order sums of
 * per-block partials in `workspace`: dn_ord_head_bwd_blocks() * (2K*16 + 2K) floats).  Supported (dn_ord_head_supported): 16 input
 * channels, H*W % 64 == 0, K <= 80; otherwise use dn_conv2d_fwd + dn_ordinal_fwd. */
int32_t dn_ord_head_supported(int32_t C_in, int64_t HW, int32_t K);
int32_t dn_ord_head_bwd_blocks(int32_t N, int64_t HW);
int dn_ord_head_fwd(const float* x, const float* mask, const float* w, const float* bias, int32_t N, int64_t HW, int32_t K, float* ord,
                    int64_t* decode, dn_stream_t stream);
int dn_ord_head_bwd(const float* x, const float* mask, const float* w, const float* bias, const float* dord, int32_t N, int64_t HW, int32_t K,
                    float* dx, int32_t accumulate, float* workspace, float* dw, float* dbias, dn_stream_t stream);
int dn_sid_labels(const float* depth, int64_t n, float ordinal_c, float beta, int32_t* labels, dn_stream_t stream);
int dn_sid_depth(const int64_t* labels, int64_t n, float ordinal_c, float beta, float* depth, dn_stream_t stream);
/* Dropout2d apply (and its backward): out[n][p][c] = x[n][p][c] * mask[n][c], NHWC. */
int dn_channel_scale(const float* x, const float* mask, int32_t N, int64_t HW, int32_t C, float* out, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer (train.py:303-305,520-522: torch.optim.Adam, wd = 0) over a flat parameter arena.
 * g is multiplied by grad_scale first (1/world_size folded in).  step >= 1.
 * ------------------------------------------------------------------------------------------------------------ */
int dn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, int32_t step, double grad_scale, dn_stream_t stream);
/* The same update with the step counter and the learning rate on the DEVICE (hyper = {lr, beta1, beta2} as doubles, step = int32
 * counter the call increments first, derived = 4 floats of scratch), so that a captured hipGraph of the training step advances
 * the bias corrections on every replay and a scheduler changes lr by writing hyper[0]. */
int dn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* hyper, double eps, double weight_decay,
                     int32_t* step, float* derived, double grad_scale, dn_stream_t stream);
/* Ranges of one arena updated by several calls (a bucket at a time, as its gradients -- and their all-reduce -- complete, under the rest of
 * the backward pass): dn_adam_tick advances the counter and the derived values ONCE per optimizer step, then every range is a
 * dn_adam_step_dev with step = NULL (no tick; `derived` as the tick left it).  Element-wise arithmetic: the same result as one call. */
int dn_adam_tick(const double* hyper, int32_t* step, float* derived, dn_stream_t stream);
int dn_fill(float* p, float value, int64_t n, dn_stream_t stream);
/* dst[i] = src[i], n floats (the owned copy engine.seed_grad takes of a gradient autograd hands over; on the launch tape like every
 * other kernel of this library, which a framework-side copy would not be). */
int dn_copy(const float* src, float* dst, int64_t n, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Launch tape (round 3).  The reference drives one training step as ~200 framework calls (train.py:437-455: forward, loss, backward,
 * optimizer.step()); at 4 images per GPU -- BASELINE.json's b32 over 8 GPUs -- the device needs ~4 ms for them and the Python side of
 * this library ~4 ms to issue them.  While a tape is being recorded (between dn_tape_begin and dn_tape_end, any thread) every kernel
 * launch of this library is ALSO kept: kernel, grid, block, LDS bytes, stream and the by-value kernel arguments.  dn_tape_replay
 * re-issues them with plain launches on the recorded streams (~2 us of host time each; a hipGraph of the same step replays SLOWER on
 * the device than the eager launches on ROCm 7.2).  The caller guarantees that every pointer the recorded
 * arguments hold is still valid and means the same at replay (graph.TapedStep records under a private memory pool and replays into the
 * same buffers), that the recorded streams exist, and that NO work the step needs was done outside this library while recording.
 *   dn_tape_fence(tape, waiter, waitee)  record "waiter waits for everything enqueued on waitee so far" (event record + stream wait at
 *                                        replay; does nothing now -- the caller fences the live run itself)
 *   dn_tape_fence_device(...)            the same between two streams whose work stays on THIS device (the engine's main and weight-gradient
 *                                        streams): the event is created without the system-scope fence (hipEventDisableSystemFence), which
 *                                        costs the recording stream's queue 2.3-5.9 us instead of 4.5-8.5 (tools/ubench/fence_cost.hip).  NOT for
 *                                        a stream whose results another GPU or the host reads (the communication stream).  When the last thing
 *                                        recorded on `waitee` is a kernel launch, the event is re-issued as that launch's STOP event
 *                                        (hipExtLaunchKernel) -- the dispatch's own completion signal, no marker packet behind the kernel
 *   dn_tape_mark(tape)                   cut: returns the number of the segment that starts here; the caller replays segment by
 *                                        segment and does its own host work in between (a gradient bucket's all-reduce)
 *   dn_tape_pause(tape, 1 / 0)           launches in between are executed but not recorded (such host work, when it is live at replay);
 *                                        a later dn_tape_fence_device never rides on a launch recorded in front of the pause
 *   dn_tape_replay(tape, segment)        segment -1: the whole tape
 * ------------------------------------------------------------------------------------------------------------ */
void* dn_tape_begin(void);
int dn_tape_end(void* tape);
int dn_tape_pause(void* tape, int32_t paused);
int dn_tape_fence(void* tape, dn_stream_t waiter, dn_stream_t waitee);
int dn_tape_fence_device(void* tape, dn_stream_t waiter, dn_stream_t waitee);
int32_t dn_tape_mark(void* tape);
int32_t dn_tape_segments(void* tape);
int64_t dn_tape_launches(void* tape);
int64_t dn_tape_fences(void* tape);
int64_t dn_tape_riding_fences(void* tape);   /* of those: device-scope fences that became the stop event of the launch in front of them */
int dn_tape_replay(void* tape, int32_t segment);
int32_t dn_tape_replay_timed(void* tape, int64_t* host_ns, const char** name, int32_t cap);   /* diagnostics: host time of every op of one replay */
void dn_tape_free(void* tape);

/* ------------------------------------------------------------------------------------------------------------
 * Input pipeline on the device (reference custom_transforms.py: RandomHorizontalFlip :56-72, ArrayToTensor :40-53, Normalize :25-37;
 * datasets/sequence_folders.py:60-77).  src = pre-decoded uint8 frames [B,H,W,C] (NHWC, as imread returns them); flip = one byte per
 * sample (non-zero: mirror along W) or NULL; dst = fp32 [B,C,H,W] with the given plane / sample strides (in floats):
 *     dst[n,c,y,x] = ((float)src[n,y,x',c] / 255 - mean[c]) / std[c],  x' = flip[n] ? W-1-x : x      (IEEE fp32, that order)
 * mean / std: device pointers (general path); mean_host / std_host: the same values on the host, used by the C == 3, W % 4 == 0
 * fast path (pass both).  dn_flip_w applies the same per-sample mirror to the fp32 ground-truth depth [B,H,W] (out of place).
 * ------------------------------------------------------------------------------------------------------------ */
int dn_u8_normalize_flip(const uint8_t* src, const uint8_t* flip, int32_t B, int32_t H, int32_t W, int32_t C, const float* mean, const float* stdv,
                         const float* mean_host, const float* std_host, float* dst, int64_t dst_stride_n, int64_t dst_stride_c, dn_stream_t stream);
int dn_flip_w(const float* src, const uint8_t* flip, int32_t B, int32_t H, int32_t W, float* dst, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Attainable-peak probes (SURVEY.md section 8d "Peaks"; used by bench.py only): a float4 streaming copy of n floats
 * (n % 4 == 0, 16-byte aligned; moves 8*n bytes) and a register-resident v_mfma_f32_32x32x2_f32 loop
 * (out: blocks*256 floats; dn_ubench_mfma_f32_flops = the FLOPs one launch executes).
 * ------------------------------------------------------------------------------------------------------------ */
int dn_ubench_copy(const float* src, float* dst, int64_t n, dn_stream_t stream);
int64_t dn_ubench_mfma_f32_flops(int32_t blocks, int32_t iters);
int dn_ubench_mfma_f32(float* out, int32_t blocks, int32_t iters, dn_stream_t stream);
/* store-only probe: n floats (multiple of 1024) written with 16-byte stores; mode 0: 1 KiB contiguous per wave instruction, mode 1: 64-byte
 * segments at a 256-byte stride (the four-instruction pattern of a 64-channel result whose lanes hold 4 channels of one pixel). */
int dn_ubench_store(float* dst, int64_t n, int32_t mode, dn_stream_t stream);
/* Block -> XCD placement probe: out[x + gx*(y + gy*z)] = HW_REG_XCC_ID of block (x, y, z) of a (gx, gy, gz) grid.  The K-split paths
 * (dn_conv_desc.splitk_ws) meet the partial tiles of one output tile in ONE XCD's L2 (no agent-scope fence); that is only correct if
 * all blocks with the same blockIdx.x land on the same XCD when gridDim.x is a multiple of 8 -- observed on MI355X, promised by
 * nobody.  The caller runs this probe once per device before it hands out a split workspace and leaves splitk_ws NULL (no split,
 * same results) if the placement is anything else. */
int dn_xcd_probe(int32_t* out, int32_t gx, int32_t gy, int32_t gz, dn_stream_t stream);
/* Probe (round 5): a reduction finished by the LAST block of a grid, across XCDs, without an agent-scope fence -- records written with
 * agent-scope relaxed atomic stores, an agent-scope counter, records read back with agent-scope relaxed atomic loads by the block that
 * arrives last (mode 0; mode 1 = plain stores / loads, the control).  data: blocks * rec floats, counter: one zeroed int (left zero),
 * result: 3 ints (+= mismatching floats, last block, += 1 per launch).  tools/exp/last_arrival_probe.py. */
int dn_last_arrival_probe(float* data, int32_t* counter, int32_t* result, int32_t blocks, int32_t rec, int32_t round, int32_t mode, dn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DISPNET_HIP_H_ */
