// Per-pixel depth losses and metrics (reference: loss_functions.py).  HBM/latency-bound reductions; wavefront (64-lane)
// shuffles for the spatial reductions, no float atomics (deterministic), no host synchronisation.
#include "dn_internal.h"

namespace dn {

constexpr int kLossThreads = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of NV values per thread; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* lds /* [NV * 16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i * 16 + wave] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += lds[i * 16 + w];
      v[i] = s;
    }
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ------------------------------------------------------------------------------------------- masked L1 / L2
// one block per sample: (sum f(gt - clamp(pred)), count) over valid = 0 < gt < max_depth
__global__ void __launch_bounds__(kLossThreads) masked_loss_stats_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                                         long long pixels, float max_depth, int kind,
                                                                         float* __restrict__ stats) {
  const int b = blockIdx.x;
  const float* g = gt + (long long)b * pixels;
  const float* p = pred + (long long)b * pixels;
  float acc[2] = {0.f, 0.f};
  for (long long i = threadIdx.x; i < pixels; i += kLossThreads) {
    const float gv = g[i];
    if (gv > 0.f && gv < max_depth) {
      const float d = gv - clampf(p[i], 1e-3f, max_depth);
      acc[0] += (kind == DN_LOSS_L1) ? fabsf(d) : d * d;
      acc[1] += 1.f;
    }
  }
  __shared__ float lds[2 * 16];
  block_sum<2>(acc, lds);
  if (threadIdx.x == 0) {
    stats[b * 2 + 0] = acc[0];
    stats[b * 2 + 1] = acc[1];
  }
}

__global__ void masked_loss_finalize_kernel(const float* __restrict__ stats, int B, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += stats[b * 2] / stats[b * 2 + 1];   // 0/0 -> NaN like mean of empty
    loss[0] = s / (float)B;
  }
}

__global__ void masked_loss_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ pred, const float* __restrict__ stats,
                                       const float* __restrict__ dloss, int B, long long pixels, float max_depth, int kind,
                                       float* __restrict__ dpred) {
  const long long total = (long long)B * pixels;
  const float dl = dloss[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / pixels);
    const float gv = gt[i], pv = pred[i];
    float out = 0.f;
    if (gv > 0.f && gv < max_depth && pv >= 1e-3f && pv <= max_depth) {
      const float d = pv - gv;   // d/dpred of f(gt - pred)
      const float fp = (kind == DN_LOSS_L1) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
      out = dl * fp / (stats[b * 2 + 1] * (float)B);
    }
    dpred[i] = out;
  }
}

// ------------------------------------------------------------------------------------ second-order smoothness
// terms (loss_functions.py:368-383) for map m[b][y][x]:
//   dx2  = dx[y][x+1]-dx[y][x]   (x < W-2)          dx[y][x] = m[y][x+1]-m[y][x]
//   dxdy = dx[y+1][x]-dx[y][x]   (y < H-1, x < W-1)
//   dydx = dy[y][x+1]-dy[y][x]   (y < H-1, x < W-1)  dy[y][x] = m[y+1][x]-m[y][x]
//   dy2  = dy[y+1][x]-dy[y][x]   (y < H-2)
__global__ void __launch_bounds__(256) smooth2_fwd_kernel(const float* __restrict__ m, int B, int H, int W, float* __restrict__ partial) {
  const long long total = (long long)B * H * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const float* c = m + i;
    const float v00 = c[0];
    if (x < W - 2) acc[0] += fabsf((c[2] - c[1]) - (c[1] - v00));
    if (y < H - 1 && x < W - 1) {
      const float v01 = c[1], v10 = c[W], v11 = c[W + 1];
      acc[1] += fabsf((v11 - v10) - (v01 - v00));
      acc[2] += fabsf((v11 - v01) - (v10 - v00));
    }
    if (y < H - 2) acc[3] += fabsf((c[2 * W] - c[W]) - (c[W] - v00));
  }
  __shared__ float lds[4 * 16];
  block_sum<4>(acc, lds);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) partial[blockIdx.x * 4 + k] = acc[k];
}

__global__ void smooth2_finalize_kernel(const float* __restrict__ partial, int blocks, int B, int H, int W, float weight,
                                        float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[4] = {0, 0, 0, 0};
  for (int b = 0; b < blocks; ++b)
    for (int k = 0; k < 4; ++k) s[k] += (double)partial[b * 4 + k];
  const double n0 = (double)B * H * (W - 2), n1 = (double)B * (H - 1) * (W - 1), n3 = (double)B * (H - 2) * W;
  const float v = (float)(s[0] / n0) + (float)(s[1] / n1) + (float)(s[2] / n1) + (float)(s[3] / n3);
  loss[0] += v * weight;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) smooth2_bwd_kernel(const float* __restrict__ m, const float* __restrict__ dloss, int B, int H, int W,
                                                          float weight, float* __restrict__ dmap) {
  const long long total = (long long)B * H * W;
  const float n0 = (float)B * H * (W - 2), n1 = (float)B * (H - 1) * (W - 1), n3 = (float)B * (H - 2) * W;
  const float scale = dloss[0] * weight;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const float* c = m + i;
    auto at = [&](int dy, int dx) { return c[dy * W + dx]; };
    float g = 0.f;
    // dx2 term anchored at (y, xa): touches xa (+1), xa+1 (-2), xa+2 (+1)
    auto t0 = [&](int xa) { return sgn((at(0, xa + 2 - x) - at(0, xa + 1 - x)) - (at(0, xa + 1 - x) - at(0, xa - x))); };
    float a0 = 0.f;
    if (x <= W - 3) a0 += t0(x);
    if (x - 1 >= 0 && x - 1 <= W - 3) a0 -= 2.f * t0(x - 1);
    if (x - 2 >= 0) a0 += t0(x - 2);
    g += a0 / n0;
    // dy2 term anchored at (ya, x)
    auto t3 = [&](int ya) { return sgn((at(ya + 2 - y, 0) - at(ya + 1 - y, 0)) - (at(ya + 1 - y, 0) - at(ya - y, 0))); };
    float a3 = 0.f;
    if (y <= H - 3) a3 += t3(y);
    if (y - 1 >= 0 && y - 1 <= H - 3) a3 -= 2.f * t3(y - 1);
    if (y - 2 >= 0) a3 += t3(y - 2);
    g += a3 / n3;
    // mixed terms anchored at (ya, xa), ya < H-1, xa < W-1: +m[ya+1][xa+1] -m[ya+1][xa] -m[ya][xa+1] +m[ya][xa]
    float a12 = 0.f;
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int ox = 0; ox < 2; ++ox) {
        const int ya = y - oy, xa = x - ox;
        if (ya < 0 || xa < 0 || ya > H - 2 || xa > W - 2) continue;
        const float v00 = at(ya - y, xa - x), v01 = at(ya - y, xa + 1 - x), v10 = at(ya + 1 - y, xa - x), v11 = at(ya + 1 - y, xa + 1 - x);
        const float coef = (oy == ox) ? 1.f : -1.f;   // (oy,ox)=(1,1):+, (1,0):-, (0,1):-, (0,0):+
        a12 += coef * (sgn((v11 - v10) - (v01 - v00)) + sgn((v11 - v01) - (v10 - v00)));
      }
    g += a12 / n1;
    dmap[i] = g * scale;
  }
}

// ------------------------------------------------------------------------------------------- compute_errors
// one block per sample -> scratch[b][9] = sums of (abs_diff, abs_rel, sq_rel, sq_err, sq_log_err, a1, a2, a3), count
__global__ void __launch_bounds__(kLossThreads) errors_stats_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int H, int W,
                                                                    float max_depth, int y1, int y2, int x1, int x2,
                                                                    float* __restrict__ scratch) {
  const int b = blockIdx.x;
  const long long pixels = (long long)H * W;
  const float* g = gt + b * pixels;
  const float* p = pred + b * pixels;
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  const float t1 = 1.25f, t2 = (float)(1.25 * 1.25), t3 = (float)(1.25 * 1.25 * 1.25);
  for (long long i = threadIdx.x; i < pixels; i += kLossThreads) {
    const int x = (int)(i % W), y = (int)(i / W);
    const float gv = g[i];
    if (gv > 0.f && gv < max_depth && y >= y1 && y < y2 && x >= x1 && x < x2) {
      const float pv = clampf(p[i], 1e-3f, max_depth);
      const float d = gv - pv;
      const float thr = fmaxf(gv / pv, pv / gv);
      const float dl = logf(gv) - logf(pv);
      acc[0] += fabsf(d);
      acc[1] += fabsf(d) / gv;
      acc[2] += d * d / gv;
      acc[3] += d * d;
      acc[4] += dl * dl;
      acc[5] += thr < t1 ? 1.f : 0.f;
      acc[6] += thr < t2 ? 1.f : 0.f;
      acc[7] += thr < t3 ? 1.f : 0.f;
      acc[8] += 1.f;
    }
  }
  __shared__ float lds[9 * 16];
  block_sum<9>(acc, lds);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 9; ++k) scratch[b * 9 + k] = acc[k];
}

__global__ void errors_finalize_kernel(const float* __restrict__ scratch, int B, float* __restrict__ out8) {
  const int k = threadIdx.x;
  if (k >= 8 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float n = scratch[b * 9 + 8];
    float v = scratch[b * 9 + k] / n;
    if (k == 3 || k == 4) v = sqrtf(v);
    s += v;
  }
  out8[k] = s / (float)B;
}

static inline int smooth_blocks(int B, int H, int W) {
  long long b = ((long long)B * H * W + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_masked_loss_fwd(const float* gt, const float* pred, int32_t B, int64_t pixels, float max_depth, int32_t kind, float* sample_stats,
                       float* loss, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && sample_stats && loss && B > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_fwd: bad argument");
  DN_REQUIRE(kind == DN_LOSS_L1 || kind == DN_LOSS_L2, DN_ERR_BAD_ARG, "dn_masked_loss_fwd: bad kind %d", kind);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(masked_loss_stats_kernel, dim3(B), dim3(kLossThreads), 0, s, gt, pred, (long long)pixels, max_depth, kind, sample_stats);
  hipLaunchKernelGGL(masked_loss_finalize_kernel, dim3(1), dim3(64), 0, s, sample_stats, B, loss);
  return check_launch("masked_loss_fwd");
}

int dn_masked_loss_bwd(const float* gt, const float* pred, const float* sample_stats, const float* dloss, int32_t B, int64_t pixels,
                       float max_depth, int32_t kind, float* dpred, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && sample_stats && dloss && dpred && B > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_bwd: bad argument");
  long long total = (long long)B * pixels;
  int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(masked_loss_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gt, pred, sample_stats, dloss, B,
                     (long long)pixels, max_depth, kind, dpred);
  return check_launch("masked_loss_bwd_kernel");
}

int32_t dn_smooth_blocks(int32_t B, int32_t H, int32_t W) { return smooth_blocks(B, H, W); }

int dn_smooth2_fwd(const float* map, int32_t B, int32_t H, int32_t W, float weight, float* partial, float* loss, dn_stream_t stream) {
  DN_REQUIRE(map && partial && loss && B > 0 && H >= 3 && W >= 3, DN_ERR_BAD_ARG, "dn_smooth2_fwd: bad argument (needs H,W >= 3)");
  hipStream_t s = as_stream(stream);
  const int blocks = smooth_blocks(B, H, W);
  hipLaunchKernelGGL(smooth2_fwd_kernel, dim3(blocks), dim3(256), 0, s, map, B, H, W, partial);
  hipLaunchKernelGGL(smooth2_finalize_kernel, dim3(1), dim3(64), 0, s, partial, blocks, B, H, W, weight, loss);
  return check_launch("smooth2_fwd");
}

int dn_smooth2_bwd(const float* map, const float* dloss, int32_t B, int32_t H, int32_t W, float weight, float* dmap, dn_stream_t stream) {
  DN_REQUIRE(map && dloss && dmap && B > 0 && H >= 3 && W >= 3, DN_ERR_BAD_ARG, "dn_smooth2_bwd: bad argument");
  hipLaunchKernelGGL(smooth2_bwd_kernel, dim3(smooth_blocks(B, H, W)), dim3(256), 0, as_stream(stream), map, dloss, B, H, W, weight, dmap);
  return check_launch("smooth2_bwd_kernel");
}

int dn_compute_errors(const float* gt, const float* pred, int32_t B, int32_t H, int32_t W, float max_depth, int32_t y1, int32_t y2,
                      int32_t x1, int32_t x2, float* scratch, float* out8, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && scratch && out8 && B > 0 && H > 0 && W > 0, DN_ERR_BAD_ARG, "dn_compute_errors: bad argument");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(errors_stats_kernel, dim3(B), dim3(kLossThreads), 0, s, gt, pred, H, W, max_depth, y1, y2, x1, x2, scratch);
  hipLaunchKernelGGL(errors_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, B, out8);
  return check_launch("compute_errors");
}

}  // extern "C"
