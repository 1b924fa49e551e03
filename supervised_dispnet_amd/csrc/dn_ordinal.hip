// DORN ordinal-regression head and loss (reference: models/Disp_vgg_BN_DORN.py:196-227, loss_functions.py:16-74,
// utils.py:106-175).  HBM-bound: each kernel streams the 2K-channel logits / K-channel probabilities once.
// Integer results (decode_c, SID labels) are produced with correctly-rounded double-precision transcendentals rounded to
// float at each step of the reference's float32 expression, so they reproduce torch-CPU's integers.
#include "dn_internal.h"

namespace dn {

__device__ __forceinline__ float wsum_o(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// pre: logits, element (n, pixel p, channel c) at pre[n*sn + p*sp + c*sc] (NHWC from the conv epilogue: sp = 2K, sc = 1; a
// user NCHW tensor: sp = 1, sc = HW); channel 2k = "A", 2k+1 = "B".  ord: planar [N][K][HW] = softmax(clamp(A), clamp(B))[1].
// decode: [N][HW] int64 = #k (ord > 0.5).  One thread per pixel, K iterations; reads are 8-byte pairs, writes are coalesced
// across the wavefront (adjacent pixels).
__global__ void __launch_bounds__(256) ordinal_fwd_kernel(const float* __restrict__ pre, long long sn, long long sp, long long sc, int N,
                                                          long long HW, int K, float* __restrict__ ord, long long* __restrict__ decode) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const float* L = pre + n * sn + p * sp;
    float* O = ord + n * K * HW + p;
    long long cnt = 0;
    for (int k = 0; k < K; ++k) {
      const float la = L[(2 * k) * sc], lb = L[(2 * k + 1) * sc];
      const float a = fminf(fmaxf(la, 1e-8f), 1e8f), b = fminf(fmaxf(lb, 1e-8f), 1e8f);
      const float m = fmaxf(a, b);
      const float ea = (float)exp((double)(a - m)), eb = (float)exp((double)(b - m));   // correctly rounded expf
      const float pb = eb / (ea + eb);
      O[(long long)k * HW] = pb;
      cnt += pb > 0.5f ? 1 : 0;
    }
    decode[i] = cnt;
  }
}

// dpre (NHWC [N][HW][2K]) from dord (planar): P = softmax[1]; dP/dB = P(1-P), dP/dA = -P(1-P); the clamp on the logits
// passes gradient only inside [1e-8, 1e8] (so negative logits get none -- reference quirk)
__global__ void __launch_bounds__(256) ordinal_bwd_kernel(const float* __restrict__ pre, long long sn, long long sp, long long sc,
                                                          const float* __restrict__ ord, const float* __restrict__ dord, int N, long long HW,
                                                          int K, float* __restrict__ dpre) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const float* L = pre + n * sn + p * sp;
    float* D = dpre + n * sn + p * sp;
    const float* O = ord + n * K * HW + p;
    const float* G = dord + n * K * HW + p;
    for (int k = 0; k < K; ++k) {
      const float la = L[(2 * k) * sc], lb = L[(2 * k + 1) * sc];
      const float P = O[(long long)k * HW];
      const float t = G[(long long)k * HW] * P * (1.f - P);
      D[(2 * k) * sc] = (la >= 1e-8f && la <= 1e8f) ? -t : 0.f;
      D[(2 * k + 1) * sc] = (lb >= 1e-8f && lb <= 1e8f) ? t : 0.f;
    }
  }
}

// DORN_loss: per valid pixel (0 < gt < max): sum_{k <= t-1} log clamp(P_k) + sum_{k > t-1} log clamp(1 - P_k), clamp to [1e-8, 1e8]
// partial[block] = (sum, #valid)
__global__ void __launch_bounds__(256) ordinal_loss_fwd_kernel(const float* __restrict__ ord, const float* __restrict__ gt,
                                                               const int* __restrict__ target, int N, long long HW, int K, float max_depth,
                                                               float* __restrict__ partial) {
  const long long total = (long long)N * HW;
  float s = 0.f, c = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float g = gt[i];
    if (!(g > 0.f && g < max_depth)) continue;
    c += 1.f;
    const long long n = i / HW, p = i - n * HW;
    const float* O = ord + n * K * HW + p;
    const int t = target[i];
    // eight planes in flight per thread (one load at a time ran at 1.65 TB/s); the sum keeps its plane order
    int k = 0;
    for (; k + 8 <= K; k += 8) {
      float P[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) P[e] = O[(long long)(k + e) * HW];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (k + e <= t - 1) ? P[e] : 1.f - P[e];
        s += logf(fminf(fmaxf(v, 1e-8f), 1e8f));
      }
    }
    for (; k < K; ++k) {
      const float P = O[(long long)k * HW];
      const float v = (k <= t - 1) ? P : 1.f - P;
      s += logf(fminf(fmaxf(v, 1e-8f), 1e8f));
    }
  }
  s = wsum_o(s);
  c = wsum_o(c);
  __shared__ float lds[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { lds[wave] = s; lds[4 + wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    partial[blockIdx.x * 2 + 1] = (lds[4] + lds[5]) + (lds[6] + lds[7]);
  }
}

// stats = (sum, num_valid); loss = sum / (-num_valid)
__global__ void __launch_bounds__(256) ordinal_loss_finalize_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ stats,
                                                                    float* __restrict__ loss) {
  // fixed-order fp64 sums by 256 threads (the single-thread loop took tens of microseconds on the critical queue)
  __shared__ double lds[8];
  double s = 0, c = 0;
  for (int k = threadIdx.x; k < blocks; k += 256) { s += (double)partial[k * 2]; c += (double)partial[k * 2 + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
  if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = s; lds[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  c = (lds[4] + lds[5]) + (lds[6] + lds[7]);
  stats[0] = (float)s;
  stats[1] = (float)c;
  loss[0] = (float)s / (-(float)c);
}

__global__ void ordinal_loss_refinalize_kernel(const float* __restrict__ stats, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] = stats[0] / (-stats[1]);
}

__global__ void __launch_bounds__(256) ordinal_loss_bwd_kernel(const float* __restrict__ ord, const float* __restrict__ gt,
                                                               const int* __restrict__ target, const float* __restrict__ stats,
                                                               const float* __restrict__ dloss, int N, long long HW, int K, float max_depth,
                                                               float grad_scale, float* __restrict__ dord) {
  const long long total = (long long)N * HW;
  const float up = grad_scale == 1.f ? dloss[0] / (-stats[1]) : (dloss[0] * grad_scale) / (-stats[1]);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float g = gt[i];
    const bool valid = g > 0.f && g < max_depth;
    const long long n = i / HW, p = i - n * HW;
    const float* O = ord + n * K * HW + p;
    float* D = dord + n * K * HW + p;
    const int t = target[i];
    for (int k = 0; k < K; ++k) {
      float d = 0.f;
      if (valid) {
        const float P = O[(long long)k * HW];
        if (k <= t - 1) {
          if (P >= 1e-8f && P <= 1e8f) d = up / P;
        } else {
          const float q = 1.f - P;
          if (q >= 1e-8f && q <= 1e8f) d = -up / q;
        }
      }
      D[(long long)k * HW] = d;
    }
  }
}

// get_labels_sid (utils.py:147-175): int( K * log((depth + 0.999) / 1) / log(beta / 1) ), float32 expression, trunc toward 0
__global__ void sid_labels_kernel(const float* __restrict__ depth, long long n, float K, float beta, int* __restrict__ labels) {
  const float lb = (float)log((double)beta);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float t = depth[i] + 0.999f;
    const float l = (float)log((double)t);
    const float v = (K * l) / lb;
    labels[i] = (int)v;    // NaN / out-of-range inputs (depth <= -0.999) follow the hardware conversion, like torch's .int()
  }
}

// get_depth_sid (utils.py:106-133): 0.5 * (beta^(l/K) + beta^((l+1)/K)) - 0.999
__global__ void sid_depth_kernel(const long long* __restrict__ labels, long long n, float K, float beta, float* __restrict__ depth) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float l = (float)labels[i];
    const float a = (float)pow((double)beta, (double)(l / K));
    const float b = (float)pow((double)beta, (double)((l + 1.f) / K));
    depth[i] = 0.5f * (a + b) - 0.999f;
  }
}

// Dropout2d (models/Disp_vgg_BN_DORN.py:112,191): out[n][p][c] = x[n][p][c] * mask[n][c]  (mask holds 0 or 1/(1-p)); also its backward
__global__ void channel_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, int N, long long HW, int C,
                                     float* __restrict__ out) {
  const long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long n = i / (HW * C);
    out[i] = x[i] * mask[n * C + c];
  }
}

static inline int ew_blocks_o(long long total, int cap = 4096) {
  long long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_ordinal_fwd(const float* pre, int64_t stride_n, int64_t stride_pix, int64_t stride_c, int32_t N, int64_t HW, int32_t K, float* ord,
                   int64_t* decode, dn_stream_t stream) {
  DN_REQUIRE(pre && ord && decode && N > 0 && HW > 0 && K > 0, DN_ERR_BAD_ARG, "dn_ordinal_fwd: bad argument");
  DN_LAUNCH(ordinal_fwd_kernel, dim3(ew_blocks_o((long long)N * HW)), dim3(256), 0, as_stream(stream), pre, (long long)stride_n,
                     (long long)stride_pix, (long long)stride_c, N, (long long)HW, K, ord, reinterpret_cast<long long*>(decode));
  return check_launch("ordinal_fwd_kernel");
}

int dn_ordinal_bwd(const float* pre, int64_t stride_n, int64_t stride_pix, int64_t stride_c, const float* ord, const float* dord, int32_t N,
                   int64_t HW, int32_t K, float* dpre, dn_stream_t stream) {
  DN_REQUIRE(pre && ord && dord && dpre && N > 0 && HW > 0 && K > 0, DN_ERR_BAD_ARG, "dn_ordinal_bwd: bad argument");
  DN_LAUNCH(ordinal_bwd_kernel, dim3(ew_blocks_o((long long)N * HW)), dim3(256), 0, as_stream(stream), pre, (long long)stride_n,
                     (long long)stride_pix, (long long)stride_c, ord, dord, N, (long long)HW, K, dpre);
  return check_launch("ordinal_bwd_kernel");
}

int32_t dn_ordinal_loss_blocks(int32_t N, int64_t HW) { return ew_blocks_o((long long)N * HW, 1024); }

int dn_ordinal_loss_fwd(const float* ord, const float* gt, const int32_t* target, int32_t N, int64_t HW, int32_t K, float max_depth,
                        float* partial, float* stats, float* loss, dn_stream_t stream) {
  DN_REQUIRE(ord && gt && target && partial && stats && loss && N > 0 && HW > 0 && K > 0, DN_ERR_BAD_ARG, "dn_ordinal_loss_fwd: bad argument");
  hipStream_t s = as_stream(stream);
  const int nb = dn_ordinal_loss_blocks(N, HW);
  DN_LAUNCH(ordinal_loss_fwd_kernel, dim3(nb), dim3(256), 0, s, ord, gt, target, N, (long long)HW, K, max_depth, partial);
  DN_LAUNCH(ordinal_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nb, stats, loss);
  return check_launch("ordinal_loss_fwd");
}

int dn_ordinal_loss_finalize(const float* stats, float* loss, dn_stream_t stream) {
  DN_REQUIRE(stats && loss, DN_ERR_BAD_ARG, "dn_ordinal_loss_finalize: bad argument");
  DN_LAUNCH(ordinal_loss_refinalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), stats, loss);
  return check_launch("ordinal_loss_refinalize_kernel");
}

int dn_ordinal_loss_bwd(const float* ord, const float* gt, const int32_t* target, const float* stats, const float* dloss, int32_t N,
                        int64_t HW, int32_t K, float max_depth, float grad_scale, float* dord, dn_stream_t stream) {
  DN_REQUIRE(ord && gt && target && stats && dloss && dord && N > 0 && HW > 0 && K > 0, DN_ERR_BAD_ARG, "dn_ordinal_loss_bwd: bad argument");
  DN_LAUNCH(ordinal_loss_bwd_kernel, dim3(ew_blocks_o((long long)N * HW)), dim3(256), 0, as_stream(stream), ord, gt, target, stats,
                     dloss, N, (long long)HW, K, max_depth, grad_scale, dord);
  return check_launch("ordinal_loss_bwd_kernel");
}

int dn_sid_labels(const float* depth, int64_t n, float ordinal_c, float beta, int32_t* labels, dn_stream_t stream) {
  DN_REQUIRE(depth && labels && n > 0 && ordinal_c > 0.f && beta > 1.f, DN_ERR_BAD_ARG, "dn_sid_labels: bad argument");
  DN_LAUNCH(sid_labels_kernel, dim3(ew_blocks_o(n)), dim3(256), 0, as_stream(stream), depth, (long long)n, ordinal_c, beta, labels);
  return check_launch("sid_labels_kernel");
}

int dn_sid_depth(const int64_t* labels, int64_t n, float ordinal_c, float beta, float* depth, dn_stream_t stream) {
  DN_REQUIRE(depth && labels && n > 0 && ordinal_c > 0.f && beta > 1.f, DN_ERR_BAD_ARG, "dn_sid_depth: bad argument");
  DN_LAUNCH(sid_depth_kernel, dim3(ew_blocks_o(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const long long*>(labels),
                     (long long)n, ordinal_c, beta, depth);
  return check_launch("sid_depth_kernel");
}

int dn_channel_scale(const float* x, const float* mask, int32_t N, int64_t HW, int32_t C, float* out, dn_stream_t stream) {
  DN_REQUIRE(x && mask && out && N > 0 && HW > 0 && C > 0, DN_ERR_BAD_ARG, "dn_channel_scale: bad argument");
  DN_LAUNCH(channel_scale_kernel, dim3(ew_blocks_o((long long)N * HW * C)), dim3(256), 0, as_stream(stream), x, mask, N,
                     (long long)HW, C, out);
  return check_launch("channel_scale_kernel");
}

}  // extern "C"
