// GPU side of the input pipeline: pre-decoded uint8 NHWC frames -> the normalised fp32 NCHW batch the networks
// take, with the per-sample horizontal flip of the reference's RandomHorizontalFlip folded in.
//
// Replaces, per image, the host chain of custom_transforms.py: RandomHorizontalFlip (:56-72, np.fliplr copy), ArrayToTensor (:40-53,
// HWC -> CHW transpose, .float()/255) and Normalize (:25-37, t.sub_(m).div_(s)) -- three passes over a float32 image on a DataLoader
// worker, then a 12 B/pixel host->device copy.  Here the host hands over the uint8 frame (3 B/pixel over PCIe) and ONE kernel writes
// the final tensor: 3 B read + 12 B written per pixel, HBM-bound.
//
// Bit-exactness: out = ((float)u8 / 255.f - mean[c]) / std[c] with IEEE fp32 divisions, the reference's operation order.
#include "dn_internal.h"

namespace dn {

constexpr int kInThreads = 256;

// W % 4 == 0: a thread owns 4 consecutive OUTPUT pixels of one row: 12 source bytes (three aligned dwords), one float4 store per plane
__global__ void __launch_bounds__(kInThreads) u8_norm_flip_vec_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ flip,
                                                                      int B, int H, int W, float m0, float m1, float m2, float s0, float s1,
                                                                      float s2, float* __restrict__ dst, long long dst_sn, long long dst_sc) {
  const int W4 = W >> 2;
  const long long total = (long long)B * H * W4;
  for (long long i = blockIdx.x * (long long)kInThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kInThreads) {
    const int g = (int)(i % W4);
    long long r = i / W4;
    const int y = (int)(r % H), n = (int)(r / H);
    const bool f = flip != nullptr && flip[n] != 0;
    const int x0 = g * 4;                                     // first output pixel
    const int sx = f ? (W - 4 - x0) : x0;                     // first source pixel of the 4-group (reversed inside when flipped)
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (((long long)n * H + y) * W + sx) * 3);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    uint8_t px[12];
    px[0] = w0 & 255; px[1] = (w0 >> 8) & 255; px[2] = (w0 >> 16) & 255; px[3] = w0 >> 24;
    px[4] = w1 & 255; px[5] = (w1 >> 8) & 255; px[6] = (w1 >> 16) & 255; px[7] = w1 >> 24;
    px[8] = w2 & 255; px[9] = (w2 >> 8) & 255; px[10] = (w2 >> 16) & 255; px[11] = w2 >> 24;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 o;
      float* po = &o.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sk = f ? 3 - k : k;
        po[k] = __fdiv_rn(__fdiv_rn((float)px[sk * 3 + c], 255.f) - mean[c], stdv[c]);
      }
      *reinterpret_cast<float4*>(dst + (long long)n * dst_sn + (long long)c * dst_sc + (long long)y * W + x0) = o;
    }
  }
}

// any W / C: one thread per output element
__global__ void __launch_bounds__(kInThreads) u8_norm_flip_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ flip, int B, int H,
                                                                  int W, int C, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                  float* __restrict__ dst, long long dst_sn, long long dst_sc) {
  const long long total = (long long)B * C * H * W;
  for (long long i = blockIdx.x * (long long)kInThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kInThreads) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), n = (int)(r / C);
    const int sx = (flip != nullptr && flip[n] != 0) ? W - 1 - x : x;
    const float v = (float)src[(((long long)n * H + y) * W + sx) * C + c];
    dst[(long long)n * dst_sn + (long long)c * dst_sc + (long long)y * W + x] = __fdiv_rn(__fdiv_rn(v, 255.f) - mean[c], stdv[c]);
  }
}

// ground-truth depth [B,H,W] fp32: the same per-sample flip (np.fliplr(gt_depth), custom_transforms.py:64)
__global__ void __launch_bounds__(kInThreads) flip_w_kernel(const float* __restrict__ src, const uint8_t* __restrict__ flip, int B, int H, int W,
                                                            float* __restrict__ dst) {
  const long long total = (long long)B * H * W;
  for (long long i = blockIdx.x * (long long)kInThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kInThreads) {
    const int x = (int)(i % W);
    const long long row = i / W;
    const int n = (int)(row / H);
    const int sx = (flip != nullptr && flip[n] != 0) ? W - 1 - x : x;
    dst[i] = src[row * W + sx];
  }
}

static inline int in_blocks(long long n) {
  long long b = (n + kInThreads - 1) / kInThreads;
  const long long cap = 256 * 16;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_u8_normalize_flip(const uint8_t* src, const uint8_t* flip, int32_t B, int32_t H, int32_t W, int32_t C, const float* mean, const float* stdv,
                         const float* mean_host, const float* std_host, float* dst, int64_t dst_stride_n, int64_t dst_stride_c, dn_stream_t stream) {
  DN_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, DN_ERR_BAD_ARG, "dn_u8_normalize_flip: bad argument");
  DN_REQUIRE(dst_stride_c >= (int64_t)H * W && dst_stride_n >= dst_stride_c * C, DN_ERR_BAD_ARG, "dn_u8_normalize_flip: planes overlap");
  hipStream_t s = as_stream(stream);
  const bool vec = C == 3 && (W & 3) == 0 && mean_host && std_host && (reinterpret_cast<uintptr_t>(src) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (dst_stride_n & 3) == 0 && (dst_stride_c & 3) == 0;
  if (vec) {
    const long long total = (long long)B * H * (W >> 2);
    DN_LAUNCH(u8_norm_flip_vec_kernel, dim3(in_blocks(total)), dim3(kInThreads), 0, s, src, flip, B, H, W, mean_host[0], mean_host[1],
                       mean_host[2], std_host[0], std_host[1], std_host[2], dst, (long long)dst_stride_n, (long long)dst_stride_c);
    return check_launch("u8_norm_flip_vec_kernel");
  }
  DN_REQUIRE(mean && stdv, DN_ERR_BAD_ARG, "dn_u8_normalize_flip: the general path needs mean / std on the device");
  DN_LAUNCH(u8_norm_flip_kernel, dim3(in_blocks((long long)B * C * H * W)), dim3(kInThreads), 0, s, src, flip, B, H, W, C, mean, stdv, dst,
                     (long long)dst_stride_n, (long long)dst_stride_c);
  return check_launch("u8_norm_flip_kernel");
}

int dn_flip_w(const float* src, const uint8_t* flip, int32_t B, int32_t H, int32_t W, float* dst, dn_stream_t stream) {
  DN_REQUIRE(src && dst && src != dst && B > 0 && H > 0 && W > 0, DN_ERR_BAD_ARG, "dn_flip_w: bad argument (not in place)");
  DN_LAUNCH(flip_w_kernel, dim3(in_blocks((long long)B * H * W)), dim3(kInThreads), 0, as_stream(stream), src, flip, B, H, W, dst);
  return check_launch("flip_w_kernel");
}

}  // extern "C"
