// HBM-bound NHWC kernels around the convolutions: BatchNorm statistics / apply / backward, fused BN+ReLU+MaxPool,
// activation backward with bias-gradient column sums, 1-channel upsamples, reciprocal, Adam.  gfx950 only.
// All column reductions are two-stage (per-block partial rows in a caller workspace, then a finalize) so results are
// deterministic and no float atomics are used.
#include "dn_internal.h"
#include "dn_fold.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kMaxReduceBlocks = 4096;

static inline int ew_blocks(long long n_items) {
  long long b = (n_items + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static inline int reduce_blocks(long long rows, int C) {
  int V = (C % 4 == 0) ? 4 : 1;
  int groups = C / V;
  int tpr = 1;
  while (tpr < groups && tpr < kThreads) tpr <<= 1;
  int rpi = kThreads / tpr;
  const int rpt = 2;        // rows each thread walks (2: 8 left a 4-image shard 416 blocks of 8 dependent row visits each)
  long long b = (rows + (long long)rpi * rpt - 1) / ((long long)rpi * rpt);
  const int cap = 1024 < kMaxReduceBlocks ? 1024 : kMaxReduceBlocks;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------------------------- generic column-reduce skeleton
// Op::apply<V>(row, c0, acc[V][NACC]) does the element-wise work for channels [c0, c0+V) of `row` and accumulates.
template <class Op, int V>
__global__ void __launch_bounds__(kThreads) colreduce_kernel(Op op, long long rows, int C, float* __restrict__ partial) {
  constexpr int NACC = Op::NACC;
  const int groups = C / V;
  int tpr = 1;
  while (tpr < groups && tpr < kThreads) tpr <<= 1;
  const int rpi = kThreads / tpr;
  const int rl = threadIdx.x / tpr, gi = threadIdx.x % tpr;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long rbeg = blockIdx.x * per;
  const long long rend = (rbeg + per < rows) ? rbeg + per : rows;
  __shared__ float red[kThreads * V * NACC];
  for (int g0 = 0; g0 < groups; g0 += tpr) {
    const int grp = g0 + gi;
    float acc[V][NACC];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[v][a] = 0.f;
    if (grp < groups)
      for (long long r = rbeg + rl; r < rend; r += rpi) op.template apply<V>(r, grp * V, acc);
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int a = 0; a < NACC; ++a) red[(threadIdx.x * V + v) * NACC + a] = acc[v][a];
    __syncthreads();
    if (rl == 0 && grp < groups) {
#pragma unroll
      for (int v = 0; v < V; ++v)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          float s = 0.f;
          for (int q = 0; q < rpi; ++q) s += red[((q * tpr + gi) * V + v) * NACC + a];
          partial[((long long)blockIdx.x * C + grp * V + v) * NACC + a] = s;
        }
    }
    __syncthreads();
  }
}

template <class Op>
static int launch_colreduce(const Op& op, long long rows, int C, float* partial, hipStream_t s, const char* what) {
  const int blocks = reduce_blocks(rows, C);
  if (C % 4 == 0)
    DN_LAUNCH((colreduce_kernel<Op, 4>), dim3(blocks), dim3(kThreads), 0, s, op, rows, C, partial);
  else
    DN_LAUNCH((colreduce_kernel<Op, 1>), dim3(blocks), dim3(kThreads), 0, s, op, rows, C, partial);
  return check_launch(what);
}

template <int V>
struct Vec {
  float v[V];
};
template <int V>
__device__ __forceinline__ Vec<V> ldv(const float* p) {
  Vec<V> r;
  if constexpr (V == 4) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = t[i];
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = p[i];
  }
  return r;
}
template <int V>
__device__ __forceinline__ void stv(float* p, const Vec<V>& r) {
  if constexpr (V == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = r.v[i];
  }
}

// -------------------------------------------------------------------------------------------------- BN statistics
// one block per channel.  partial[t][c] = (sum, M2 about the tile mean) of row tile t (kBnTileRows rows, the last one ragged);
// merged in fp64: mean = sum_t s_t / N,  M2 = sum_t [ M2_t + n_t (s_t/n_t - mean)^2 ]   (Chan et al.)
constexpr int kBnTileRows = 128;   // = BM of every igemm_conv_kernel instantiation
__global__ void __launch_bounds__(kThreads) bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                                               const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* running_mean, float* running_var,
                                                               float momentum, float eps, float* mean_out, float* invstd_out, float* scale,
                                                               float* shift, long long* num_batches_tracked) {
  const int c = blockIdx.x;
  if (num_batches_tracked != nullptr && c == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;   // nn.BatchNorm2d's step counter
  // One pass over the partial rows: per row t the pair (s_t, M2_t about the tile's own mean) is read once and three sums are kept in
  // fp64 -- S = sum s_t, Q = sum M2_t, P = sum n_t (s_t / n_t - pivot)^2 with pivot = the first tile's mean -- from which
  //   mean = S / N,   M2 = Q + sum_t n_t (m_t - mean)^2 = Q + P - N (mean - pivot)^2      (Chan et al., expanded about the pivot).
  // About a pivot inside the data's range the two terms that cancel are both of the order of the between-tile spread, so a channel
  // with |mean| >> std (mean^2 / var ~ 1e10 and beyond) keeps its variance; the expansion about 0 lost it there.
  __shared__ double red[3][4];
  double s1 = 0.0, q = 0.0, pp = 0.0;
  const double n0 = count < (double)kBnTileRows ? count : (double)kBnTileRows;
  const double pivot = (double)partial[(long long)c * 2] / n0;
#pragma unroll 8
  for (int r = threadIdx.x; r < rows; r += kThreads) {      // (8 strided 8-byte loads in flight)
    const float2 v = *reinterpret_cast<const float2*>(partial + ((long long)r * C + c) * 2);
    const double left = count - (double)r * kBnTileRows;
    const double nt = left < (double)kBnTileRows ? left : (double)kBnTileRows;
    const double dm = (double)v.x / nt - pivot;
    s1 += (double)v.x;
    q += (double)v.y;
    pp += nt * dm * dm;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    q += __shfl_xor(q, o);
    pp += __shfl_xor(pp, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = q;
    red[2][threadIdx.x >> 6] = pp;
  }
  __syncthreads();
  double macc = 0.0, m2tot = 0.0;
  if (threadIdx.x == 0) {
    const double S = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double Q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double P = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    macc = S / count;
    m2tot = Q + (P - count * (macc - pivot) * (macc - pivot));
  }
  if (threadIdx.x == 0) {
    double var = m2tot / count;   // biased; the conv bias shifts the mean only
    if (var < 0.0) var = 0.0;
    const float mean = (float)(macc + (conv_bias ? (double)conv_bias[c] : 0.0));
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// The sliced order of dn_fold.h as kernels of their own (<= kFoldMaxRows partial rows): one block per 64 channels.  The Winograd kernels
// run the same device functions in their last-arriving block when the caller asked for it -- same bits either way.
__global__ void __launch_bounds__(256) bn_finalize_sliced_kernel(const float* __restrict__ partial, int rows, int C, double count, BnFinalizeArgs a) {
  __shared__ double red[3 * kFoldSlices * 64];
  const int c0 = blockIdx.x * 64;
  bn_finalize_sliced<false>(partial, rows, C, c0, C - c0 < 64 ? C - c0 : 64, count, a, red, threadIdx.x);
}

__global__ void __launch_bounds__(256) colsum2_sliced_kernel(const float* __restrict__ partial, int rows, int C, int stride, int offset,
                                                             float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ double red[2 * kFoldSlices * 64];
  const int c0 = blockIdx.x * 64;
  colsum2_sliced<false>(partial, rows, C, stride, offset, c0, C - c0 < 64 ? C - c0 : 64, out0, out1, red, threadIdx.x);
}

__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

// --------------------------------------------------------------------------------------- BN + ReLU + MaxPool 2x2
__global__ void __launch_bounds__(kThreads) bn_relu_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int N, int H, int W, int C,
                                                                    float* __restrict__ pooled, uint8_t* __restrict__ idx) {
  const int PH = H / 2, PW = W / 2, G = C / 4;
  const long long total = (long long)N * PH * PW * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int g = (int)(i % G);
    long long pix = i / G;
    const int px = (int)(pix % PW);
    pix /= PW;
    const int py = (int)(pix % PH), n = (int)(pix / PH);
    const int c = g * 4;
    // scale == nullptr: y is a plain (already activated) tensor -- nn.MaxPool2d(2, 2) after conv + ReLU (models/Disp_vgg.py:79-100)
    const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sh = scale ? *reinterpret_cast<const f32x4*>(shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* base = y + (((long long)n * H + 2 * py) * W + 2 * px) * C + c;
    f32x4 best;
    int bi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((long long)(q >> 1) * W + (q & 1)) * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = scale ? fmaxf(0.f, v[e] * sc[e] + sh[e]) : v[e];
        if (q == 0 || a > best[e]) {
          best[e] = a;
          bi[e] = q;
        }
      }
    }
    const long long o = i * 4;
    *reinterpret_cast<f32x4*>(pooled + o) = best;
    uchar4 code;
    code.x = (uint8_t)(bi[0] | (best[0] > 0.f ? 4 : 0));
    code.y = (uint8_t)(bi[1] | (best[1] > 0.f ? 4 : 0));
    code.z = (uint8_t)(bi[2] | (best[2] > 0.f ? 4 : 0));
    code.w = (uint8_t)(bi[3] | (best[3] > 0.f ? 4 : 0));
    *reinterpret_cast<uchar4*>(idx + o) = code;
  }
}

// gradient of the plain 2x2 max-pool: every input pixel of a window gets the window's gradient if it was the arg-max, else 0
__global__ void __launch_bounds__(kThreads) maxpool2_bwd_kernel(const float* __restrict__ dpooled, const uint8_t* __restrict__ idx, int N, int H,
                                                                int W, int C, float* __restrict__ dx, int accumulate) {
  const int PH = H / 2, PW = W / 2, G = C / 4;
  const long long total = (long long)N * PH * PW * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int g = (int)(i % G);
    long long pix = i / G;
    const int px = (int)(pix % PW);
    pix /= PW;
    const int py = (int)(pix % PH), n = (int)(pix / PH);
    const uchar4 code = *reinterpret_cast<const uchar4*>(idx + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(dpooled + i * 4);
    float* base = dx + (((long long)n * H + 2 * py) * W + 2 * px) * C + g * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* dst = base + ((long long)(q >> 1) * W + (q & 1)) * C;
      f32x4 o = accumulate ? *reinterpret_cast<const f32x4*>(dst) : f32x4{0.f, 0.f, 0.f, 0.f};
      if ((code.x & 3) == q) o[0] += gv[0];
      if ((code.y & 3) == q) o[1] += gv[1];
      if ((code.z & 3) == q) o[2] += gv[2];
      if ((code.w & 3) == q) o[3] += gv[3];
      *reinterpret_cast<f32x4*>(dst) = o;
    }
  }
}

struct PoolBwdOp {
  static constexpr int NACC = 2;
  const float* dpooled;
  const uint8_t* idx;
  const float* y;
  const float* mean;
  const float* invstd;
  float* dz;
  int PH, PW, H, W, C;
  template <int V>
  __device__ __forceinline__ void apply(long long row, int c0, float (&acc)[V][2]) const {
    const int px = (int)(row % PW);
    long long t = row / PW;
    const int py = (int)(t % PH), n = (int)(t / PH);
    const Vec<V> dp = ldv<V>(dpooled + row * C + c0);
    const Vec<V> mu = ldv<V>(mean + c0), is = ldv<V>(invstd + c0);
    const long long base = (((long long)n * H + 2 * py) * W + 2 * px) * C + c0;
    Vec<V> out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < V; ++e) out[q].v[e] = 0.f;
    unsigned codes = 0;                                   // the V routing bytes of this thread in one load when they form a dword
    if constexpr (V == 4) codes = *reinterpret_cast<const unsigned*>(idx + row * C + c0);
    // the four window pixels as whole vectors, all in flight at once (the arg-max positions of a thread's channels touch nearly every
    // 32-byte sector of the window anyway; four coalesced 16-byte loads replace four dependent, divergent 4-byte gathers)
    Vec<V> yq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) yq[q] = ldv<V>(y + base + ((long long)(q >> 1) * W + (q & 1)) * C);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int code = V == 4 ? (int)((codes >> (8 * e)) & 0xffu) : (int)idx[row * C + c0 + e];
      if (code & 4) {
        const int q = code & 3;
        const float g = dp.v[e];
        const float yv = q == 0 ? yq[0].v[e] : (q == 1 ? yq[1].v[e] : (q == 2 ? yq[2].v[e] : yq[3].v[e]));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (qq == q) out[qq].v[e] = g;
        acc[e][0] += g;
        acc[e][1] += g * (yv - mu.v[e]) * is.v[e];
      }
    }
    if (dz != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) stv<V>(dz + base + ((long long)(q >> 1) * W + (q & 1)) * C, out[q]);
    }
  }
};

struct BnReluBwdOp {
  static constexpr int NACC = 2;
  int write;                      // 0: sums only (the apply pass re-derives the mask from y, dn_bn_bwd_apply_relu)
  float* da_dz;
  const float* y;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  int C;
  template <int V>
  __device__ __forceinline__ void apply(long long row, int c0, float (&acc)[V][2]) const {
    const long long o = row * C + c0;
    Vec<V> g = ldv<V>(da_dz + o);
    const Vec<V> yv = ldv<V>(y + o), sc = ldv<V>(scale + c0), sh = ldv<V>(shift + c0), mu = ldv<V>(mean + c0), is = ldv<V>(invstd + c0);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float dz = (yv.v[e] * sc.v[e] + sh.v[e] > 0.f) ? g.v[e] : 0.f;
      g.v[e] = dz;
      acc[e][0] += dz;
      acc[e][1] += dz * (yv.v[e] - mu.v[e]) * is.v[e];
    }
    if (write) stv<V>(da_dz + o, g);
  }
};

struct ActBwdOp {
  static constexpr int NACC = 1;
  const float* src;      // == g: in place
  float* g;
  const float* y_post;
  int act;
  float p0, p1;
  int C;
  template <int V>
  __device__ __forceinline__ void apply(long long row, int c0, float (&acc)[V][1]) const {
    const long long o = row * C + c0;
    Vec<V> gv = ldv<V>(src + o);
    if (act == DN_ACT_NONE) {
      if (src != g) stv<V>(g + o, gv);
    } else {
      const Vec<V> yv = ldv<V>(y_post + o);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float yp = yv.v[e];
        float d = 1.f;
        if (act == DN_ACT_RELU) d = yp > 0.f ? 1.f : 0.f;
        else if (act == DN_ACT_LEAKY) d = yp > 0.f ? 1.f : p0;
        else if (act == DN_ACT_ELU) d = yp > 0.f ? 1.f : (yp + 1.f);
        else if (act == DN_ACT_SIGMOID_AFFINE) {
          const float sg = (yp - p1) / p0;
          d = p0 * sg * (1.f - sg);
        }
        gv.v[e] *= d;
      }
      stv<V>(g + o, gv);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e][0] += gv.v[e];
  }
};

// out[c] = sum_rows partial[(row*C + c)*stride + offset].  One 64-lane wave per channel: lanes stride over the rows
// (fp64 accumulation), wave-shuffle reduction -- the row count is up to 1024, so a serial per-thread loop is latency-bound.
__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ partial, int rows, int C, int stride, int offset,
                                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += (double)partial[((long long)r * C + c) * stride + offset];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[c] = (float)s;
}

// both BatchNorm-backward sums of a channel in one launch: out0[c] = sum_r partial[r][c][offset], out1[c] = sum_r partial[r][c][offset + 1].
// One block of 256 threads per channel (was: one wave per channel, four channels per block): at 32 images the input-gradient epilogues
// leave up to 13 312 partial rows per channel, and 16 blocks of strided 8-byte reads took 22-40 us per layer.  The rows meet in double
// precision in a fixed order (thread-strided partial sums, then a tree over the 256 threads): deterministic.
__global__ void __launch_bounds__(256) colsum2_finalize_kernel(const float* __restrict__ partial, int rows, int C, int stride, int offset,
                                                               float* __restrict__ out0, float* __restrict__ out1) {
  const int c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float* q = partial + ((long long)r * C + c) * stride + offset;
    s0 += (double)q[0];
    s1 += (double)q[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
  }
  __shared__ double red[2][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out0[c] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    out1[c] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

// dy = gamma * invstd * (dz - sum(dz) / n - xhat * sum(dz * xhat) / n), written with explicit fused steps: the four kernels below must
// produce the SAME bits for the same element whatever the compiler hoists out of their loops (two of them keep the per-channel
// factors in registers across iterations), tests/test_gpu_graph.py::test_bn_backward_without_materialised_dz_is_bitwise_the_two_pass_form
__device__ __forceinline__ float bn_bwd_value(float z, float yv, float mu, float is, float ga, float dg, float db, float inv_count) {
#pragma clang fp contract(off)
  const float xhat = (yv - mu) * is;
  float t = __builtin_fmaf(-db, inv_count, z);
  t = __builtin_fmaf(-(xhat * dg), inv_count, t);
  return (ga * is) * t;
}

__global__ void __launch_bounds__(kThreads) bn_bwd_apply_kernel(float* __restrict__ dz_dy, const float* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, long long rows, int C, float inv_count) {
  const int G = C / 4;
  const long long total = rows * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int c = (int)(i % G) * 4;
    const long long o = i * 4;
    f32x4 dz = *reinterpret_cast<const f32x4*>(dz_dy + o);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), dg = *reinterpret_cast<const f32x4*>(dgamma + c),
                db = *reinterpret_cast<const f32x4*>(dbeta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dz[e] = bn_bwd_value(dz[e], yv[e], mu[e], is[e], ga[e], dg[e], db[e], inv_count);
    }
    *reinterpret_cast<f32x4*>(dz_dy + o) = dz;
  }
}

// The same BatchNorm backward with the ReLU mask / the max-pool routing RE-DERIVED here instead of read back from a materialised dz:
// the reduce pass then only produces the two sums (no full-size write), and after a pool the sparse full-resolution dz never exists.
//   relu:  dz = da * (y*scale+shift > 0)                                   (da is overwritten with dy)
//   pool:  dz[n, 2py+a, 2px+b, c] = dpooled[n,py,px,c] if idx says (a,b) is the arg-max and the pooled value was > 0, else 0
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_relu_kernel(float* __restrict__ da_dy, const float* __restrict__ y,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                                     const float* __restrict__ dbeta, long long rows, int C, float inv_count) {
  const int G = C / 4;
  const long long total = rows * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int c = (int)(i % G) * 4;
    const long long o = i * 4;
    f32x4 dz = *reinterpret_cast<const f32x4*>(da_dy + o);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), dg = *reinterpret_cast<const f32x4*>(dgamma + c),
                db = *reinterpret_cast<const f32x4*>(dbeta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = (yv[e] * sc[e] + sh[e] > 0.f) ? dz[e] : 0.f;
      dz[e] = bn_bwd_value(z, yv[e], mu[e], is[e], ga[e], dg[e], db[e], inv_count);
    }
    *reinterpret_cast<f32x4*>(da_dy + o) = dz;
  }
}

// The same pass when C / 4 divides the block size (every layer of the nets here): a thread's channel group is the same in every
// iteration of its grid-stride loop, so the seven per-channel vectors are loaded ONCE (the plain kernel re-reads them and takes a 64-bit
// remainder per element group), and four element groups are in flight per thread.  Same arithmetic, same bits.
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_relu_hoisted_kernel(float* __restrict__ da_dy, const float* __restrict__ y,
                                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                             const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                                             const float* __restrict__ dbeta, long long rows, int C, float inv_count) {
  const int G = C / 4;
  const long long total = rows * G;
  const int c = (int)(threadIdx.x % (unsigned)G) * 4;      // (blockIdx.x * 256 and the grid stride are multiples of G)
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), dg = *reinterpret_cast<const f32x4*>(dgamma + c),
              db = *reinterpret_cast<const f32x4*>(dbeta + c);
  const long long step = (long long)gridDim.x * kThreads;
  for (long long i0 = blockIdx.x * (long long)kThreads + threadIdx.x; i0 < total; i0 += 4 * step) {
    f32x4 dz[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * step;
      if (i < total) {
        dz[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(da_dy + i * 4));
        yv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + i * 4));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * step;
      if (i < total) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = (yv[u][e] * sc[e] + sh[e] > 0.f) ? dz[u][e] : 0.f;
          dz[u][e] = bn_bwd_value(z, yv[u][e], mu[e], is[e], ga[e], dg[e], db[e], inv_count);
        }
        *reinterpret_cast<f32x4*>(da_dy + i * 4) = dz[u];
      }
    }
  }
}

// The dz form (ResNet bottleneck bn3 / downsample BatchNorms, whose gradient arrives as dz from bn_add_relu_bwd) with the per-channel
// vectors hoisted too (round 5: config 4 ran 20 launches of the plain kernel per step at 4.0 TB/s, the hoisted ReLU form does 5.0-5.3):
// the grid stride is a multiple of G = C / 4 (the launcher rounds the grid), so a thread's channel group never changes -- also for
// C = 2048, where G = 512 spans two blocks.  Same arithmetic (bn_bwd_value), same bits.
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_hoisted_kernel(float* __restrict__ dz_dy, const float* __restrict__ y,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                                        const float* __restrict__ dbeta, long long rows, int C, float inv_count) {
  const int G = C / 4;
  const long long total = rows * G;
  const long long first = blockIdx.x * (long long)kThreads + threadIdx.x;
  const int c = (int)(first % G) * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), dg = *reinterpret_cast<const f32x4*>(dgamma + c),
              db = *reinterpret_cast<const f32x4*>(dbeta + c);
  const long long step = (long long)gridDim.x * kThreads;
  for (long long i0 = first; i0 < total; i0 += 4 * step) {
    f32x4 dz[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * step;
      if (i < total) {
        dz[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dz_dy + i * 4));
        yv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + i * 4));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * step;
      if (i < total) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[u][e] = bn_bwd_value(dz[u][e], yv[u][e], mu[e], is[e], ga[e], dg[e], db[e], inv_count);
        *reinterpret_cast<f32x4*>(dz_dy + i * 4) = dz[u];
      }
    }
  }
}

__global__ void __launch_bounds__(kThreads) bn_bwd_apply_pool_kernel(const float* __restrict__ dpooled, const uint8_t* __restrict__ idx,
                                                                     const float* __restrict__ y, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta, int N,
                                                                     int H, int W, int C, float inv_count, float* __restrict__ dy) {
  const int G = C / 4, PH = H / 2, PW = W / 2;
  const long long total = (long long)N * PH * PW * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int g = (int)(i % G);
    long long pix = i / G;
    const int px = (int)(pix % PW);
    pix /= PW;
    const int py = (int)(pix % PH), n = (int)(pix / PH);
    const int c = g * 4;
    const f32x4 dp = *reinterpret_cast<const f32x4*>(dpooled + i * 4);
    const uchar4 code = *reinterpret_cast<const uchar4*>(idx + i * 4);
    const int cd[4] = {code.x, code.y, code.z, code.w};
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), dg = *reinterpret_cast<const f32x4*>(dgamma + c),
                db = *reinterpret_cast<const f32x4*>(dbeta + c);
    const long long base = (((long long)n * H + 2 * py) * W + 2 * px) * C + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long o = base + ((long long)(q >> 1) * W + (q & 1)) * C;
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = ((cd[e] & 4) && (cd[e] & 3) == q) ? dp[e] : 0.f;
        out[e] = bn_bwd_value(z, yv[e], mu[e], is[e], ga[e], dg[e], db[e], inv_count);
      }
      *reinterpret_cast<f32x4*>(dy + o) = out;
    }
  }
}

// ------------------------------------------------------------------------------------------ ResNet bottleneck tail
__global__ void __launch_bounds__(kThreads) bn_add_relu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ r,
                                                                   const float* __restrict__ r_scale, const float* __restrict__ r_shift,
                                                                   long long rows, int C, float* __restrict__ out) {
  const int G = C / 4;
  const long long total = rows * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int c = (int)(i % G) * 4;
    const long long o = i * 4;
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
    const f32x4 rv = r != nullptr ? *reinterpret_cast<const f32x4*>(r + o) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 res;
    if (r_scale != nullptr) {
      const f32x4 rs = *reinterpret_cast<const f32x4*>(r_scale + c), rb = *reinterpret_cast<const f32x4*>(r_shift + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) res[e] = fmaxf(0.f, (yv[e] * sc[e] + sh[e]) + (rv[e] * rs[e] + rb[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) res[e] = fmaxf(0.f, (yv[e] * sc[e] + sh[e]) + rv[e]);
    }
    *reinterpret_cast<f32x4*>(out + o) = res;
  }
}

struct AddReluBwdOp {
  static constexpr int NACC = 4;
  const float* gout;
  const float* out;
  const float* y;
  const float* mean;
  const float* invstd;
  const float* r;
  const float* r_mean;     // nullable: plain identity
  const float* r_invstd;
  float* dz_y;
  float* dr;
  int dr_accumulate;
  int C;
  template <int V>
  __device__ __forceinline__ void apply(long long row, int c0, float (&acc)[V][4]) const {
    const long long o = row * C + c0;
    const Vec<V> g = ldv<V>(gout + o), ov = ldv<V>(out + o), yv = ldv<V>(y + o), mu = ldv<V>(mean + c0), is = ldv<V>(invstd + c0);
    Vec<V> m;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      m.v[e] = ov.v[e] > 0.f ? g.v[e] : 0.f;
      acc[e][0] += m.v[e];
      acc[e][1] += m.v[e] * (yv.v[e] - mu.v[e]) * is.v[e];
    }
    stv<V>(dz_y + o, m);
    if (r_mean != nullptr) {
      const Vec<V> rv = ldv<V>(r + o), rm = ldv<V>(r_mean + c0), ri = ldv<V>(r_invstd + c0);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        acc[e][2] += m.v[e];
        acc[e][3] += m.v[e] * (rv.v[e] - rm.v[e]) * ri.v[e];
      }
      stv<V>(dr + o, m);
    } else if (dr != nullptr) {
      if (dr_accumulate) {
        Vec<V> d = ldv<V>(dr + o);
#pragma unroll
        for (int e = 0; e < V; ++e) d.v[e] += m.v[e];
        stv<V>(dr + o, d);
      } else {
        stv<V>(dr + o, m);
      }
    }
  }
};

// ----------------------------------------------------------------------------------------- MaxPool 3x3 / stride 2 / pad 1
__global__ void __launch_bounds__(kThreads) maxpool3s2_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                                                  float* __restrict__ out, uint8_t* __restrict__ idx) {
  const int G = C / 4;
  const long long total = (long long)N * OH * OW * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int g = (int)(i % G);
    long long pix = i / G;
    const int ox = (int)(pix % OW);
    pix /= OW;
    const int oy = (int)(pix % OH), n = (int)(pix / OH);
    f32x4 best = f32x4{-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    int bi[4] = {0, 0, 0, 0};
    bool any = false;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int iy = 2 * oy - 1 + q / 3, ix = 2 * ox - 1 + q % 3;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)n * H + iy) * W + ix) * C + g * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!any || v[e] > best[e]) {
          best[e] = v[e];
          bi[e] = q;
        }
      any = true;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = best;
    uchar4 code;
    code.x = (uint8_t)bi[0]; code.y = (uint8_t)bi[1]; code.z = (uint8_t)bi[2]; code.w = (uint8_t)bi[3];
    *reinterpret_cast<uchar4*>(idx + i * 4) = code;
  }
}

__global__ void __launch_bounds__(kThreads) maxpool3s2_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx, int N, int H,
                                                                  int W, int C, int OH, int OW, float* __restrict__ dx, int accumulate) {
  const int G = C / 4;
  const long long total = (long long)N * H * W * G;
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int g = (int)(i % G);
    long long pix = i / G;
    const int ix = (int)(pix % W);
    pix /= W;
    const int iy = (int)(pix % H), n = (int)(pix / H);
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(dx + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // windows (oy, ox) with 2*oy - 1 <= iy <= 2*oy + 1
    for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= OH) continue;
      const int qy = iy - (2 * oy - 1);
      for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= OW) continue;
        const int q = qy * 3 + (ix - (2 * ox - 1));
        const long long o = ((((long long)n * OH + oy) * OW + ox) * G + g) * 4;
        const uchar4 code = *reinterpret_cast<const uchar4*>(idx + o);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dout + o);
        if (code.x == q) s[0] += gv[0];
        if (code.y == q) s[1] += gv[1];
        if (code.z == q) s[2] += gv[2];
        if (code.w == q) s[3] += gv[3];
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = s;
  }
}

// backward of nearest x2 upsample for NHWC with C channels
__global__ void upsample2x_nearest_bwd_nhwc_kernel(const float* __restrict__ dfull, int N, int h, int w, int C, float* __restrict__ dlow,
                                                   int accumulate) {
  const long long total = (long long)N * h * w * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int x = (int)(t % w);
    t /= w;
    const int yy = (int)(t % h), n = (int)(t / h);
    const float* b = dfull + ((((long long)n * 2 * h + 2 * yy) * 2 * w + 2 * x) * C + c);
    const long long rs = (long long)2 * w * C;
    float s = (b[0] + b[C]) + (b[rs] + b[rs + C]);
    if (accumulate) s += dlow[i];
    dlow[i] = s;
  }
}

__device__ __forceinline__ int reflect_idx(int v, int n) {
  const int m = n - 1;
  int a = v < 0 ? -v : v;
  a = m - a;
  a = a < 0 ? -a : a;
  return m - a;
}

// dx[n][y][x][c] (+)= sum over padded (py, px) in [-pad, H+pad) x [-pad, W+pad) with reflect(py) == y, reflect(px) == x
__global__ void reflect_fold_kernel(const float* __restrict__ dxp, int N, int H, int W, int C, int pad, float* __restrict__ dx, int accumulate) {
  const long long total = (long long)N * H * W * C;
  const int PH = H + 2 * pad, PW = W + 2 * pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H), n = (int)(t / H);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y;
    if (y >= 1 && y <= pad) ys[ny++] = -y;
    if (y <= H - 2 && y >= H - 1 - pad) ys[ny++] = 2 * (H - 1) - y;
    xs[nx++] = x;
    if (x >= 1 && x <= pad) xs[nx++] = -x;
    if (x <= W - 2 && x >= W - 1 - pad) xs[nx++] = 2 * (W - 1) - x;
    float s = accumulate ? dx[i] : 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += dxp[(((long long)n * PH + ys[a] + pad) * PW + xs[b] + pad) * C + c];
    dx[i] = s;
  }
}

// out[n][c] = scale * mean over the HW pixels of x[n][p][c]   (pose = 0.01 * pose_pred.mean(3).mean(2), models/PoseExpNet.py:73-75)
// one block per sample; threads stride over (pixel, channel) so consecutive lanes read consecutive addresses
__global__ void __launch_bounds__(kThreads) spatial_mean_fwd_kernel(const float* __restrict__ x, long long HW, int C, float scale,
                                                                    float* __restrict__ out) {
  const int n = blockIdx.x;
  const float* X = x + (long long)n * HW * C;
  __shared__ float red[kThreads];
  for (int c0 = 0; c0 < C; c0 += kThreads) {
    // thread t handles channel c0 + (t % cw) for pixels t / cw, t / cw + stride, ...
    const int cw = (C - c0) < kThreads ? (C - c0) : kThreads;
    const int lanes_per_c = kThreads / cw;
    const int c = threadIdx.x % cw, pl = threadIdx.x / cw;
    float s = 0.f;
    if (pl < lanes_per_c)
      for (long long p = pl; p < HW; p += lanes_per_c) s += X[p * C + c0 + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cw) {
      float t = 0.f;
      for (int q = 0; q < lanes_per_c; ++q) t += red[q * cw + threadIdx.x];
      out[(long long)n * C + c0 + threadIdx.x] = scale * (t / (float)HW);
    }
    __syncthreads();
  }
}

__global__ void spatial_mean_bwd_kernel(const float* __restrict__ dout, int N, long long HW, int C, float scale, float* __restrict__ dx) {
  const long long total = (long long)N * HW * C;
  const float k = scale / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long n = i / (HW * C);
    dx[i] = dout[n * C + c] * k;
  }
}

__global__ void sub_div_kernel(const float* __restrict__ x, long long n, float sub, float div, float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = (x[i] - sub) / div;
}

// ----------------------------------------------------------------------------------------------- 1-channel helpers
__global__ void upsample2x_nearest_bwd_kernel(const float* __restrict__ dfull, int N, int h, int w, float* __restrict__ dlow, int accumulate) {
  const long long total = (long long)N * h * w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    const long long t = i / w;
    const int yy = (int)(t % h), n = (int)(t / h);
    const float* b = dfull + (((long long)n * 2 * h + 2 * yy) * 2 * w + 2 * x);
    float s = (b[0] + b[1]) + (b[2 * w] + b[2 * w + 1]);
    if (accumulate) s += dlow[i];
    dlow[i] = s;
  }
}

__device__ __forceinline__ void bilin_src(int o, int in_size, int* i0, int* i1, float* l1) {
  float src = 0.5f * ((float)o + 0.5f) - 0.5f;   // scale 1/2, align_corners = False
  if (src < 0.f) src = 0.f;
  int a = (int)src;
  if (a > in_size - 1) a = in_size - 1;
  *i0 = a;
  *i1 = a + (a < in_size - 1 ? 1 : 0);
  *l1 = src - (float)a;
}

__global__ void upsample2x_bilinear_fwd_kernel(const float* __restrict__ low, int N, int h, int w, int OH, int OW, float* __restrict__ out) {
  const long long total = (long long)N * OH * OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const long long t = i / OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(oy, h, &y0, &y1, &ly);
    bilin_src(ox, w, &x0, &x1, &lx);
    const float* b = low + (long long)n * h * w;
    const float hy = 1.f - ly, hx = 1.f - lx;
    out[i] = hy * (hx * b[y0 * w + x0] + lx * b[y0 * w + x1]) + ly * (hx * b[y1 * w + x0] + lx * b[y1 * w + x1]);
  }
}

__global__ void upsample2x_bilinear_bwd_kernel(const float* __restrict__ dout, int N, int h, int w, int OH, int OW, float* __restrict__ dlow,
                                               int accumulate) {
  const long long total = (long long)N * h * w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    const long long t = i / w;
    const int yy = (int)(t % h), n = (int)(t / h);
    const float* g = dout + (long long)n * OH * OW;
    float s = 0.f;
    for (int oy = 2 * yy - 2; oy <= 2 * yy + 2; ++oy) {
      if (oy < 0 || oy >= OH) continue;
      int y0, y1;
      float ly;
      bilin_src(oy, h, &y0, &y1, &ly);
      float wy = 0.f;
      if (y0 == yy) wy += 1.f - ly;
      if (y1 == yy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = 2 * x - 2; ox <= 2 * x + 2; ++ox) {
        if (ox < 0 || ox >= OW) continue;
        int x0, x1;
        float lx;
        bilin_src(ox, w, &x0, &x1, &lx);
        float wx = 0.f;
        if (x0 == x) wx += 1.f - lx;
        if (x1 == x) wx += lx;
        if (wx != 0.f) s += wy * wx * g[(long long)oy * OW + ox];
      }
    }
    if (accumulate) s += dlow[i];
    dlow[i] = s;
  }
}

__global__ void reciprocal_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = 1.f / x[i];
}
__global__ void reciprocal_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = y[i];
    dx[i] = -dy[i] * v * v;
  }
}

__global__ void fill_kernel(float* __restrict__ p, float v, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// torch.optim.Adam (single-tensor formulation, amsgrad off): lerp for exp_avg, addcmul for exp_avg_sq,
// denom = sqrt(v)/sqrt(bc2) + eps, p -= (lr/bc1) * m / denom.
__global__ void __launch_bounds__(kThreads) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float beta1, float beta2, float eps,
                                                        float weight_decay, float step_size, float bc2_sqrt, float grad_scale) {
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    vi = vi * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// Graph-replayable form: the step counter lives on the device, so a captured launch sequence advances it by itself.
//   hyper = {lr, beta1, beta2} in double (what the host form receives); derived = {step_size, bc2_sqrt, beta1, beta2} in float
__global__ void adam_tick_kernel(int* __restrict__ step, const double* __restrict__ hyper, float* __restrict__ derived) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int t = step[0] + 1;
  step[0] = t;
  // beta^t by repeated squaring (<= 31 dependent multiplies; within an ulp or two of pow() in fp64, i.e. far below the fp32 the
  // corrections are handed on in): the device library's double pow() made this one-thread kernel take 115 us on the critical queue
  auto ipow = [](double b, int e) {
    double r = 1.0;
    while (e > 0) {
      if (e & 1) r *= b;
      b *= b;
      e >>= 1;
    }
    return r;
  };
  const double bc1 = 1.0 - ipow(hyper[1], t);
  const double bc2 = 1.0 - ipow(hyper[2], t);
  derived[0] = (float)(hyper[0] / bc1);
  derived[1] = (float)sqrt(bc2);
  derived[2] = (float)hyper[1];
  derived[3] = (float)hyper[2];
}

__global__ void __launch_bounds__(kThreads) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, long long n, float eps,
                                                            float weight_decay, const float* __restrict__ derived, float grad_scale) {
  const float beta1 = derived[2], beta2 = derived[3], step_size = derived[0], bc2_sqrt = derived[1];
  for (long long i = blockIdx.x * (long long)kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    vi = vi * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

}  // namespace dn

using namespace dn;

extern "C" {

int32_t dn_reduce_blocks(int64_t rows, int32_t C) { return reduce_blocks(rows, C); }

int dn_bn_finalize(const float* partial, int32_t rows, int32_t C, int64_t count, const float* conv_bias, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                   float* scale, float* shift, int64_t* num_batches_tracked, dn_stream_t stream) {
  DN_REQUIRE(partial && gamma && beta && mean && invstd && scale && shift && rows > 0 && C > 0 && count > 0, DN_ERR_BAD_ARG,
             "dn_bn_finalize: bad argument");
  if (rows <= kFoldMaxRows) {          // the order a folded finalize uses (dn_conv_desc.bnf_*): bit-identical results either way
    const BnFinalizeArgs a{conv_bias, gamma, beta, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps,
                           mean, invstd, scale, shift};
    DN_LAUNCH(bn_finalize_sliced_kernel, dim3((C + 63) / 64), dim3(256), 0, as_stream(stream), partial, rows, C, (double)count, a);
    return check_launch("bn_finalize_sliced_kernel");
  }
  DN_LAUNCH(bn_finalize_kernel, dim3(C), dim3(kThreads), 0, as_stream(stream), partial, rows, C, (double)count, conv_bias, gamma,
                     beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                     reinterpret_cast<long long*>(num_batches_tracked));
  return check_launch("bn_finalize_kernel");
}

int dn_bn_eval_affine(int32_t C, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                      float* scale, float* shift, dn_stream_t stream) {
  DN_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && scale && shift, DN_ERR_BAD_ARG, "dn_bn_eval_affine: bad argument");
  DN_LAUNCH(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), C, gamma, beta, running_mean,
                     running_var, eps, scale, shift);
  return check_launch("bn_eval_affine_kernel");
}

int dn_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, int32_t N, int32_t H, int32_t W, int32_t C, float* pooled,
                        uint8_t* idx, dn_stream_t stream) {
  DN_REQUIRE(y && pooled && idx && ((scale == nullptr) == (shift == nullptr)), DN_ERR_BAD_ARG, "dn_bn_relu_pool_fwd: null pointer");
  DN_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && N > 0, DN_ERR_UNSUPPORTED, "dn_bn_relu_pool_fwd: need C%%4==0 and even H,W");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  DN_LAUNCH(bn_relu_pool_fwd_kernel, dim3(ew_blocks(total)), dim3(kThreads), 0, as_stream(stream), y, scale, shift, N, H, W, C,
                     pooled, idx);
  return check_launch("bn_relu_pool_fwd_kernel");
}

int dn_maxpool2_bwd(const float* dpooled, const uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, float* dx, int32_t accumulate,
                    dn_stream_t stream) {
  DN_REQUIRE(dpooled && idx && dx && N > 0, DN_ERR_BAD_ARG, "dn_maxpool2_bwd: bad argument");
  DN_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, DN_ERR_UNSUPPORTED, "dn_maxpool2_bwd: need C%%4==0 and even H,W");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  DN_LAUNCH(maxpool2_bwd_kernel, dim3(ew_blocks(total)), dim3(kThreads), 0, as_stream(stream), dpooled, idx, N, H, W, C, dx, accumulate);
  return check_launch("maxpool2_bwd_kernel");
}

int dn_bn_relu_pool_bwd(const float* dpooled, const uint8_t* idx, const float* y, const float* mean, const float* invstd, int32_t N,
                        int32_t H, int32_t W, int32_t C, float* dz, float* partial, dn_stream_t stream) {
  DN_REQUIRE(dpooled && idx && y && mean && invstd && dz && partial, DN_ERR_BAD_ARG, "dn_bn_relu_pool_bwd: null pointer");
  DN_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, DN_ERR_UNSUPPORTED, "dn_bn_relu_pool_bwd: need C%%4==0 and even H,W");
  PoolBwdOp op{dpooled, idx, y, mean, invstd, dz, H / 2, W / 2, H, W, C};
  return launch_colreduce(op, (long long)N * (H / 2) * (W / 2), C, partial, as_stream(stream), "bn_relu_pool_bwd");
}

int dn_bn_relu_pool_bwd_sums(const float* dpooled, const uint8_t* idx, const float* y, const float* mean, const float* invstd, int32_t N,
                             int32_t H, int32_t W, int32_t C, float* partial, dn_stream_t stream) {
  DN_REQUIRE(dpooled && idx && y && mean && invstd && partial, DN_ERR_BAD_ARG, "dn_bn_relu_pool_bwd_sums: null pointer");
  DN_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, DN_ERR_UNSUPPORTED, "dn_bn_relu_pool_bwd_sums: need C%%4==0 and even H,W");
  PoolBwdOp op{dpooled, idx, y, mean, invstd, nullptr, H / 2, W / 2, H, W, C};
  return launch_colreduce(op, (long long)N * (H / 2) * (W / 2), C, partial, as_stream(stream), "bn_relu_pool_bwd_sums");
}

int dn_bn_relu_bwd_reduce(float* da_dz, const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                          int64_t rows, int32_t C, float* partial, dn_stream_t stream) {
  DN_REQUIRE(da_dz && y && scale && shift && mean && invstd && partial && rows > 0 && C > 0, DN_ERR_BAD_ARG,
             "dn_bn_relu_bwd_reduce: bad argument");
  BnReluBwdOp op{1, da_dz, y, scale, shift, mean, invstd, C};
  return launch_colreduce(op, rows, C, partial, as_stream(stream), "bn_relu_bwd_reduce");
}

int dn_bn_relu_bwd_sums(const float* da, const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                        int64_t rows, int32_t C, float* partial, dn_stream_t stream) {
  DN_REQUIRE(da && y && scale && shift && mean && invstd && partial && rows > 0 && C > 0, DN_ERR_BAD_ARG, "dn_bn_relu_bwd_sums: bad argument");
  BnReluBwdOp op{0, const_cast<float*>(da), y, scale, shift, mean, invstd, C};
  return launch_colreduce(op, rows, C, partial, as_stream(stream), "bn_relu_bwd_sums");
}

static void launch_colsum2(const float* partial, int32_t partial_rows, int32_t C, int32_t partial_stride, int32_t partial_offset, float* out0,
                           float* out1, hipStream_t s) {
  // few rows: the sliced order a folded sum uses (dn_conv_desc.bnb_dgamma / bnb_dbeta) -- bit-identical either way
  if (partial_rows <= kFoldMaxRows && (partial_stride & 1) == 0 && (partial_offset & 1) == 0 && (reinterpret_cast<uintptr_t>(partial) & 7) == 0)
    DN_LAUNCH(colsum2_sliced_kernel, dim3((C + 63) / 64), dim3(256), 0, s, partial, partial_rows, C, partial_stride, partial_offset, out0, out1);
  else
    DN_LAUNCH(colsum2_finalize_kernel, dim3(C), dim3(256), 0, s, partial, partial_rows, C, partial_stride, partial_offset, out0, out1);
}

// partial == nullptr: dgamma / dbeta already hold the two sums (the input-gradient launch that produced the partial rows finished them
// itself: dn_conv_desc.bnb_dgamma / bnb_dbeta, dn_conv_dgrad_folds_bn_sums)
static int bn_bwd_sums_to_params(const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset, int32_t C,
                                 float* dgamma, float* dbeta, hipStream_t s, const char* who) {
  DN_REQUIRE(dgamma && dbeta, DN_ERR_BAD_ARG, "%s: bad argument", who);
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "%s: need C%%4==0", who);
  if (partial == nullptr) return DN_OK;
  DN_REQUIRE(partial_rows > 0, DN_ERR_BAD_ARG, "%s: bad argument", who);
  DN_REQUIRE(partial_stride >= 2 && partial_offset >= 0 && partial_offset + 1 < partial_stride, DN_ERR_BAD_ARG, "%s: partial layout", who);
  launch_colsum2(partial, partial_rows, C, partial_stride, partial_offset, dbeta, dgamma, s);
  return DN_OK;
}

int dn_bn_bwd_apply_relu(float* da_dy, const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* gamma, const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset,
                         int64_t rows, int32_t C, float* dgamma, float* dbeta, dn_stream_t stream) {
  DN_REQUIRE(da_dy && y && scale && shift && mean && invstd && gamma && rows > 0, DN_ERR_BAD_ARG, "dn_bn_bwd_apply_relu: bad argument");
  hipStream_t s = as_stream(stream);
  int rc = bn_bwd_sums_to_params(partial, partial_rows, partial_stride, partial_offset, C, dgamma, dbeta, s, "dn_bn_bwd_apply_relu");
  if (rc != DN_OK) return rc;
  const bool hoisted = C % 4 == 0 && kThreads % (C / 4) == 0 && !false;
  set_last_kernel(hoisted ? "dn::bn_bwd_apply_relu_hoisted_kernel" : "dn::bn_bwd_apply_relu_kernel");
  if (hoisted)
    DN_LAUNCH(bn_bwd_apply_relu_hoisted_kernel, dim3(ew_blocks((rows * (C / 4) + 3) / 4)), dim3(kThreads), 0, s, da_dy, y, scale, shift, mean,
              invstd, gamma, dgamma, dbeta, (long long)rows, C, (float)(1.0 / (double)rows));
  else
    DN_LAUNCH(bn_bwd_apply_relu_kernel, dim3(ew_blocks(rows * (C / 4))), dim3(kThreads), 0, s, da_dy, y, scale, shift, mean, invstd, gamma,
                     dgamma, dbeta, (long long)rows, C, (float)(1.0 / (double)rows));
  return check_launch("bn_bwd_apply_relu");
}

int dn_bn_bwd_apply_pool(const float* dpooled, const uint8_t* idx, const float* y, const float* mean, const float* invstd, const float* gamma,
                         const float* partial, int32_t partial_rows, int32_t partial_stride, int32_t partial_offset, int32_t N, int32_t H,
                         int32_t W, int32_t C, float* dy, float* dgamma, float* dbeta, dn_stream_t stream) {
  DN_REQUIRE(dpooled && idx && y && mean && invstd && gamma && dy && N > 0, DN_ERR_BAD_ARG, "dn_bn_bwd_apply_pool: bad argument");
  DN_REQUIRE(H % 2 == 0 && W % 2 == 0, DN_ERR_UNSUPPORTED, "dn_bn_bwd_apply_pool: need even H, W");
  hipStream_t s = as_stream(stream);
  int rc = bn_bwd_sums_to_params(partial, partial_rows, partial_stride, partial_offset, C, dgamma, dbeta, s, "dn_bn_bwd_apply_pool");
  if (rc != DN_OK) return rc;
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  DN_LAUNCH(bn_bwd_apply_pool_kernel, dim3(ew_blocks(total)), dim3(kThreads), 0, s, dpooled, idx, y, mean, invstd, gamma, dgamma, dbeta, N,
                     H, W, C, (float)(1.0 / ((double)N * H * W)), dy);
  return check_launch("bn_bwd_apply_pool");
}

int dn_bn_bwd_apply(float* dz_dy, const float* y, const float* mean, const float* invstd, const float* gamma, const float* partial,
                    int32_t partial_rows, int32_t partial_stride, int32_t partial_offset, int64_t rows, int32_t C, float* dgamma,
                    float* dbeta, dn_stream_t stream) {
  DN_REQUIRE(dz_dy && y && mean && invstd && gamma && partial && dgamma && dbeta && rows > 0, DN_ERR_BAD_ARG, "dn_bn_bwd_apply: bad argument");
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "dn_bn_bwd_apply: need C%%4==0");
  DN_REQUIRE(partial_stride >= 2 && partial_offset >= 0 && partial_offset + 1 < partial_stride, DN_ERR_BAD_ARG, "dn_bn_bwd_apply: partial layout");
  hipStream_t s = as_stream(stream);
  launch_colsum2(partial, partial_rows, C, partial_stride, partial_offset, dbeta, dgamma, s);
  // hoisted form: a grid whose stride (blocks x 256 threads) is a multiple of the channel groups G = C / 4
  const int G = C / 4;
  int blocks = ew_blocks((rows * (long long)G + 3) / 4);
  int ga = G, gb = kThreads;                           // per = lcm(G, 256) / 256 = G / gcd(G, 256) blocks span a whole number of pixels
  while (gb != 0) {
    const int t = ga % gb;
    ga = gb;
    gb = t;
  }
  const int per = G / ga;
  if (per <= 64) {
    blocks = (blocks + per - 1) / per * per;
    set_last_kernel("dn::bn_bwd_apply_hoisted_kernel");
    DN_LAUNCH(bn_bwd_apply_hoisted_kernel, dim3(blocks), dim3(kThreads), 0, s, dz_dy, y, mean, invstd, gamma, dgamma, dbeta, (long long)rows, C,
              (float)(1.0 / (double)rows));
  } else {
    set_last_kernel("dn::bn_bwd_apply_kernel");
    DN_LAUNCH(bn_bwd_apply_kernel, dim3(ew_blocks(rows * (C / 4))), dim3(kThreads), 0, s, dz_dy, y, mean, invstd, gamma, dgamma, dbeta,
              (long long)rows, C, (float)(1.0 / (double)rows));
  }
  return check_launch("bn_bwd_apply");
}

int dn_act_bwd_reduce(float* g, const float* y_post, int32_t act, float p0, float p1, int64_t rows, int32_t C, float* partial,
                      dn_stream_t stream) {
  DN_REQUIRE(g && partial && rows > 0 && C > 0 && (act == DN_ACT_NONE || y_post), DN_ERR_BAD_ARG, "dn_act_bwd_reduce: bad argument");
  ActBwdOp op{g, g, y_post, act, p0, p1, C};
  return launch_colreduce(op, rows, C, partial, as_stream(stream), "act_bwd_reduce");
}

// out of place: g_out = g_in * act'(.) -- the incoming gradient is the framework's tensor and must not be written (engine.seed_grad used to
// copy it first: one more launch on the critical stream)
int dn_act_bwd_reduce_from(const float* g_in, float* g_out, const float* y_post, int32_t act, float p0, float p1, int64_t rows, int32_t C,
                           float* partial, dn_stream_t stream) {
  DN_REQUIRE(g_in && g_out && partial && rows > 0 && C > 0 && (act == DN_ACT_NONE || y_post), DN_ERR_BAD_ARG, "dn_act_bwd_reduce_from: bad argument");
  ActBwdOp op{g_in, g_out, y_post, act, p0, p1, C};
  return launch_colreduce(op, rows, C, partial, as_stream(stream), "act_bwd_reduce_from");
}

int dn_colsum_finalize(const float* partial, int32_t rows, int32_t C, int32_t stride, int32_t offset, float* out, dn_stream_t stream) {
  DN_REQUIRE(partial && out && rows > 0 && C > 0 && stride > 0 && offset >= 0 && offset < stride, DN_ERR_BAD_ARG, "dn_colsum_finalize: bad argument");
  DN_LAUNCH(colsum_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), partial, rows, C, stride, offset, out);
  return check_launch("colsum_finalize_kernel");
}

int dn_upsample2x_nearest_bwd(const float* dfull, int32_t N, int32_t h, int32_t w, float* dlow, int32_t accumulate, dn_stream_t stream) {
  DN_REQUIRE(dfull && dlow && N > 0 && h > 0 && w > 0, DN_ERR_BAD_ARG, "dn_upsample2x_nearest_bwd: bad argument");
  DN_LAUNCH(upsample2x_nearest_bwd_kernel, dim3(ew_blocks((long long)N * h * w)), dim3(256), 0, as_stream(stream), dfull, N, h, w,
                     dlow, accumulate);
  return check_launch("upsample2x_nearest_bwd_kernel");
}

int dn_upsample2x_bilinear_fwd(const float* low, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* out, dn_stream_t stream) {
  DN_REQUIRE(low && out && N > 0 && OH <= 2 * h && OW <= 2 * w && OH > 0 && OW > 0, DN_ERR_BAD_ARG, "dn_upsample2x_bilinear_fwd: bad argument");
  DN_LAUNCH(upsample2x_bilinear_fwd_kernel, dim3(ew_blocks((long long)N * OH * OW)), dim3(256), 0, as_stream(stream), low, N, h, w,
                     OH, OW, out);
  return check_launch("upsample2x_bilinear_fwd_kernel");
}

int dn_upsample2x_bilinear_bwd(const float* dout, int32_t N, int32_t h, int32_t w, int32_t OH, int32_t OW, float* dlow, int32_t accumulate,
                               dn_stream_t stream) {
  DN_REQUIRE(dout && dlow && N > 0 && OH <= 2 * h && OW <= 2 * w && OH > 0 && OW > 0, DN_ERR_BAD_ARG, "dn_upsample2x_bilinear_bwd: bad argument");
  DN_LAUNCH(upsample2x_bilinear_bwd_kernel, dim3(ew_blocks((long long)N * h * w)), dim3(256), 0, as_stream(stream), dout, N, h, w,
                     OH, OW, dlow, accumulate);
  return check_launch("upsample2x_bilinear_bwd_kernel");
}

int dn_bn_add_relu_fwd(const float* y, const float* scale, const float* shift, const float* r, const float* r_scale, const float* r_shift,
                       int64_t rows, int32_t C, float* out, dn_stream_t stream) {
  DN_REQUIRE(y && scale && shift && out && rows > 0 && C > 0 && ((r_scale == nullptr) == (r_shift == nullptr)) && (r || !r_scale),
             DN_ERR_BAD_ARG, "dn_bn_add_relu_fwd: bad argument");
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "dn_bn_add_relu_fwd: need C%%4==0");
  DN_LAUNCH(bn_add_relu_fwd_kernel, dim3(ew_blocks(rows * (C / 4))), dim3(kThreads), 0, as_stream(stream), y, scale, shift, r, r_scale,
                     r_shift, (long long)rows, C, out);
  return check_launch("bn_add_relu_fwd_kernel");
}

int dn_bn_add_relu_bwd(const float* gout, const float* out, const float* y, const float* mean, const float* invstd, const float* r,
                       const float* r_mean, const float* r_invstd, int64_t rows, int32_t C, float* dz_y, float* dr, int32_t dr_accumulate,
                       float* partial, dn_stream_t stream) {
  DN_REQUIRE(gout && out && y && mean && invstd && dz_y && partial && rows > 0 && C > 0, DN_ERR_BAD_ARG, "dn_bn_add_relu_bwd: bad argument");
  DN_REQUIRE((r_mean == nullptr) == (r_invstd == nullptr), DN_ERR_BAD_ARG, "dn_bn_add_relu_bwd: r_mean / r_invstd go together");
  DN_REQUIRE(r_mean == nullptr || (r && dr), DN_ERR_BAD_ARG, "dn_bn_add_relu_bwd: the downsample branch needs r and dr");
  AddReluBwdOp op{gout, out, y, mean, invstd, r, r_mean, r_invstd, dz_y, dr, dr_accumulate, C};
  return launch_colreduce(op, rows, C, partial, as_stream(stream), "bn_add_relu_bwd");
}

int32_t dn_maxpool3s2_out(int32_t H, int32_t ceil_mode) {
  if (!ceil_mode) return (H - 1) / 2 + 1;                  // floor((H + 2 - 3) / 2) + 1
  int o = H / 2 + 1;                                        // ceil((H - 1) / 2) + 1
  if ((o - 1) * 2 >= H + 1) --o;                            // (ATen: the last window must start inside the input or its left padding)
  return o;
}

int dn_maxpool3s2_fwd(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ceil_mode, float* out, uint8_t* idx, dn_stream_t stream) {
  DN_REQUIRE(x && out && idx && N > 0 && H > 0 && W > 0, DN_ERR_BAD_ARG, "dn_maxpool3s2_fwd: bad argument");
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "dn_maxpool3s2_fwd: need C%%4==0");
  const int OH = dn_maxpool3s2_out(H, ceil_mode), OW = dn_maxpool3s2_out(W, ceil_mode);
  DN_LAUNCH(maxpool3s2_fwd_kernel, dim3(ew_blocks((long long)N * OH * OW * (C / 4))), dim3(kThreads), 0, as_stream(stream), x, N, H, W,
                     C, OH, OW, out, idx);
  return check_launch("maxpool3s2_fwd_kernel");
}

int dn_maxpool3s2_bwd(const float* dout, const uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ceil_mode, float* dx,
                      int32_t accumulate, dn_stream_t stream) {
  DN_REQUIRE(dout && idx && dx && N > 0 && H > 0 && W > 0, DN_ERR_BAD_ARG, "dn_maxpool3s2_bwd: bad argument");
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "dn_maxpool3s2_bwd: need C%%4==0");
  const int OH = dn_maxpool3s2_out(H, ceil_mode), OW = dn_maxpool3s2_out(W, ceil_mode);
  DN_LAUNCH(maxpool3s2_bwd_kernel, dim3(ew_blocks((long long)N * H * W * (C / 4))), dim3(kThreads), 0, as_stream(stream), dout, idx, N, H,
                     W, C, OH, OW, dx, accumulate);
  return check_launch("maxpool3s2_bwd_kernel");
}

int dn_upsample2x_nearest_bwd_nhwc(const float* dfull, int32_t N, int32_t h, int32_t w, int32_t C, float* dlow, int32_t accumulate,
                                   dn_stream_t stream) {
  DN_REQUIRE(dfull && dlow && N > 0 && h > 0 && w > 0 && C > 0, DN_ERR_BAD_ARG, "dn_upsample2x_nearest_bwd_nhwc: bad argument");
  DN_LAUNCH(upsample2x_nearest_bwd_nhwc_kernel, dim3(ew_blocks((long long)N * h * w * C)), dim3(256), 0, as_stream(stream), dfull, N, h,
                     w, C, dlow, accumulate);
  return check_launch("upsample2x_nearest_bwd_nhwc_kernel");
}

int dn_reflect_fold(const float* dxp, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, float* dx, int32_t accumulate, dn_stream_t stream) {
  DN_REQUIRE(dxp && dx && N > 0 && C > 0 && pad >= 1 && H > pad && W > pad, DN_ERR_BAD_ARG, "dn_reflect_fold: bad argument (needs H, W > pad)");
  DN_LAUNCH(reflect_fold_kernel, dim3(ew_blocks((long long)N * H * W * C)), dim3(256), 0, as_stream(stream), dxp, N, H, W, C, pad, dx,
                     accumulate);
  return check_launch("reflect_fold_kernel");
}

int dn_sub_div(const float* x, int64_t n, float sub, float div, float* out, dn_stream_t stream) {
  DN_REQUIRE(x && out && n > 0 && div != 0.f, DN_ERR_BAD_ARG, "dn_sub_div: bad argument");
  DN_LAUNCH(sub_div_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), x, (long long)n, sub, div, out);
  return check_launch("sub_div_kernel");
}

int dn_spatial_mean_fwd(const float* x, int32_t N, int64_t HW, int32_t C, float scale, float* out, dn_stream_t stream) {
  DN_REQUIRE(x && out && N > 0 && HW > 0 && C > 0, DN_ERR_BAD_ARG, "dn_spatial_mean_fwd: bad argument");
  DN_LAUNCH(spatial_mean_fwd_kernel, dim3(N), dim3(kThreads), 0, as_stream(stream), x, (long long)HW, C, scale, out);
  return check_launch("spatial_mean_fwd_kernel");
}

int dn_spatial_mean_bwd(const float* dout, int32_t N, int64_t HW, int32_t C, float scale, float* dx, dn_stream_t stream) {
  DN_REQUIRE(dout && dx && N > 0 && HW > 0 && C > 0, DN_ERR_BAD_ARG, "dn_spatial_mean_bwd: bad argument");
  DN_LAUNCH(spatial_mean_bwd_kernel, dim3(ew_blocks((long long)N * HW * C)), dim3(256), 0, as_stream(stream), dout, N, (long long)HW, C,
                     scale, dx);
  return check_launch("spatial_mean_bwd_kernel");
}

int dn_reciprocal_fwd(const float* x, float* y, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(x && y && n > 0, DN_ERR_BAD_ARG, "dn_reciprocal_fwd: bad argument");
  DN_LAUNCH(reciprocal_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), x, y, (long long)n);
  return check_launch("reciprocal_fwd_kernel");
}

int dn_reciprocal_bwd(const float* dy, const float* y, float* dx, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(dy && y && dx && n > 0, DN_ERR_BAD_ARG, "dn_reciprocal_bwd: bad argument");
  DN_LAUNCH(reciprocal_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), dy, y, dx, (long long)n);
  return check_launch("reciprocal_bwd_kernel");
}

int dn_fill(float* p, float value, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(p && n >= 0, DN_ERR_BAD_ARG, "dn_fill: bad argument");
  if (n == 0) return DN_OK;
  DN_LAUNCH(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), p, value, (long long)n);
  return check_launch("fill_kernel");
}

__global__ void __launch_bounds__(256) copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

int dn_copy(const float* src, float* dst, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(src && dst && n >= 0, DN_ERR_BAD_ARG, "dn_copy: bad argument");
  if (n == 0) return DN_OK;
  DN_LAUNCH(copy_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), src, dst, (long long)n);
  return check_launch("copy_kernel");
}

int dn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int32_t step, double grad_scale, dn_stream_t stream) {
  DN_REQUIRE(p && g && m && v && n > 0 && step >= 1, DN_ERR_BAD_ARG, "dn_adam_step: bad argument");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  DN_LAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(kThreads), 0, as_stream(stream), p, g, m, v, (long long)n, (float)beta1, (float)beta2, (float)eps,
                     (float)weight_decay, step_size, bc2_sqrt, (float)grad_scale);
  return check_launch("adam_kernel");
}

int dn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* hyper, double eps, double weight_decay,
                     int32_t* step, float* derived, double grad_scale, dn_stream_t stream) {
  DN_REQUIRE(p && g && m && v && n > 0 && hyper && derived, DN_ERR_BAD_ARG, "dn_adam_step_dev: bad argument");
  hipStream_t s = as_stream(stream);
  if (step != nullptr) DN_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, s, step, hyper, derived);
  DN_LAUNCH(adam_dev_kernel, dim3(ew_blocks(n)), dim3(kThreads), 0, s, p, g, m, v, (long long)n, (float)eps, (float)weight_decay,
                     derived, (float)grad_scale);
  return check_launch("adam_dev_kernel");
}

int dn_adam_tick(const double* hyper, int32_t* step, float* derived, dn_stream_t stream) {
  DN_REQUIRE(hyper && step && derived, DN_ERR_BAD_ARG, "dn_adam_tick: bad argument");
  DN_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, as_stream(stream), step, hyper, derived);
  return check_launch("adam_tick_kernel");
}

}  // extern "C"
