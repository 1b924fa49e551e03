// Per-pixel depth losses and metrics (reference: loss_functions.py).  HBM/latency-bound reductions; wavefront (64-lane)
// shuffles for the spatial reductions, no float atomics (deterministic), no host synchronisation.
#include "dn_fold.h"
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

// ------------------------------------------------------------------------------------- masked depth losses
// A "group" is a run of `pixels` consecutive elements that shares one mask mean: one sample for l1/l2/berhu/scale-inv
// (loss_functions.py:77-189), the whole batch for the Multiscale_* family (:217-315).  valid = 0 < gt < max_depth,
// p = clamp(pred, 1e-3, max_depth), d = gt - p, r = |d|.
//   L1   : mean r                 L2 : mean d^2
//   berHu: c = 0.2 max r;  mean( r > c ? (r^2 + c^2)/(2c) : r )      (the max is differentiable, like torch's)
//   SI   : mean (|gt|-|p|)^2 - 0.5 (sum d)^2 / n^2
// stats[g][8]: 0 sum f | 1 n | 2 sum d (SI) or sum_{r>c} (0.5 - r^2/(2c^2)) (berHu) | 3 max r | 4 #(r == max) | 5 group loss
constexpr int kStat = 8;
constexpr int kSplitMax = 64;

static inline int loss_splits(long long pixels) {
  long long s = (pixels + 2047) / 2048;       // 8 pixels per thread: a 4-image shard of the metric's batch is 104 blocks, not 16 (40 -> 10 us)
  return (int)(s < 1 ? 1 : (s > kSplitMax ? kSplitMax : s));
}

// pass A: per (split, group) partial [4] = (sum f, n, sum d, max r); berHu only needs n and max here
__global__ void __launch_bounds__(256) masked_stats_kernel(const float* __restrict__ gt, const float* __restrict__ pred, long long pixels,
                                                           float max_depth, int kind, float* __restrict__ partial) {
  const int g = blockIdx.y, sp = blockIdx.x, nsp = gridDim.x;
  const float* G = gt + (long long)g * pixels;
  const float* P = pred + (long long)g * pixels;
  float acc[3] = {0.f, 0.f, 0.f};
  float mx = 0.f;
  for (long long i = sp * 256ll + threadIdx.x; i < pixels; i += nsp * 256ll) {
    const float gv = G[i];
    if (gv > 0.f && gv < max_depth) {
      const float p = clampf(P[i], 1e-3f, max_depth);
      const float d = gv - p;
      float f;
      if (kind == DN_LOSS_L1) f = fabsf(d);
      else if (kind == DN_LOSS_L2) f = d * d;
      else if (kind == DN_LOSS_SCALE_INV) { const float e = fabsf(gv) - fabsf(p); f = e * e; }
      else f = 0.f;
      acc[0] += f;
      acc[1] += 1.f;
      acc[2] += d;
      mx = fmaxf(mx, fabsf(d));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __shared__ float lds[3 * 16];
  __shared__ float lmx[4];
  if ((threadIdx.x & 63) == 0) lmx[threadIdx.x >> 6] = mx;
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0) {
    float* o = partial + ((long long)g * nsp + sp) * 4;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    o[3] = fmaxf(fmaxf(lmx[0], lmx[1]), fmaxf(lmx[2], lmx[3]));
  }
}

// The three launches of the forward (pass A, the reduction over the splits, the division) as ONE: every block stores its partial with
// agent-scope stores, the block that arrives last (dn_fold.h) reduces the splits of every group -- thread g walks group g's splits in index
// order, exactly masked_reduceA_kernel -- and thread 0 turns the groups into the loss in group order, exactly masked_loss_finalize_kernel:
// bit-identical to the three launches (tests/test_gpu_losses.py).  A 4-image step spends 6-7 us on each of the two small launches and
// another 7 on the fill of the statistics rows, on its critical stream.  G <= kFoldLossGroups; not berHu (a second pass over the pixels).
constexpr int kFoldLossGroups = 256;

__global__ void __launch_bounds__(256) masked_stats_fold_kernel(const float* __restrict__ gt, const float* __restrict__ pred, long long pixels,
                                                                float max_depth, int kind, float* __restrict__ partial, int G,
                                                                float* __restrict__ stats, float weight, int accumulate, float* __restrict__ loss,
                                                                int* __restrict__ counter) {
  const int g = blockIdx.y, sp = blockIdx.x, nsp = gridDim.x;
  const float* Gp = gt + (long long)g * pixels;
  const float* P = pred + (long long)g * pixels;
  float acc[3] = {0.f, 0.f, 0.f};
  float mx = 0.f;
  for (long long i = sp * 256ll + threadIdx.x; i < pixels; i += nsp * 256ll) {
    const float gv = Gp[i];
    if (gv > 0.f && gv < max_depth) {
      const float p = clampf(P[i], 1e-3f, max_depth);
      const float d = gv - p;
      float f;
      if (kind == DN_LOSS_L1) f = fabsf(d);
      else if (kind == DN_LOSS_L2) f = d * d;
      else if (kind == DN_LOSS_SCALE_INV) { const float e = fabsf(gv) - fabsf(p); f = e * e; }
      else f = 0.f;
      acc[0] += f;
      acc[1] += 1.f;
      acc[2] += d;
      mx = fmaxf(mx, fabsf(d));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __shared__ float lds[3 * 16];
  __shared__ float lmx[4];
  __shared__ int last_flag;
  __shared__ float red[3][kFoldLossGroups];
  if ((threadIdx.x & 63) == 0) lmx[threadIdx.x >> 6] = mx;
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0) {
    float* o = partial + ((long long)g * nsp + sp) * 4;
    fold_store(o + 0, acc[0]);
    fold_store(o + 1, acc[1]);
    fold_store(o + 2, acc[2]);
    fold_store(o + 3, fmaxf(fmaxf(lmx[0], lmx[1]), fmaxf(lmx[2], lmx[3])));
  }
  if (!fold_last_arrival(counter, (int)(gridDim.x * gridDim.y), &last_flag)) return;
  const int t = threadIdx.x;
  if (t < G) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, m = 0.f;
    for (int k = 0; k < nsp; ++k) {
      const float* p = partial + ((long long)t * nsp + k) * 4;
      const float2 a = fold_load2<true>(p), b = fold_load2<true>(p + 2);
      s0 += a.x; s1 += a.y; s2 += b.x; m = fmaxf(m, b.y);
    }
    float* o = stats + t * kStat;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = m; o[4] = 0.f;
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
  }
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int q = 0; q < G; ++q) {
      const float n = red[1][q];
      float L = red[0][q] / n;
      if (kind == DN_LOSS_SCALE_INV) L = L - (red[2][q] * red[2][q]) * 0.5f / (n * n);
      stats[q * kStat + 5] = L;
      s += L;
    }
    const float v = weight * (s / (float)G);
    loss[0] = accumulate ? loss[0] + v : v;
  }
}

// reduces the split partials of every group (one thread per group; G and nsp are small)
__global__ void masked_reduceA_kernel(const float* __restrict__ partial, int G, int nsp, float* __restrict__ stats) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
  for (int k = 0; k < nsp; ++k) {
    const float* p = partial + ((long long)g * nsp + k) * 4;
    s0 += p[0]; s1 += p[1]; s2 += p[2]; mx = fmaxf(mx, p[3]);
  }
  float* o = stats + g * kStat;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = mx; o[4] = 0.f;
}

// berHu pass B: with c = 0.2 max r known: partial [4] = (sum f, sum_{r>c} (0.5 - r^2/(2c^2)), #(r == max), 0)
__global__ void __launch_bounds__(256) berhu_stats_kernel(const float* __restrict__ gt, const float* __restrict__ pred, long long pixels,
                                                          float max_depth, const float* __restrict__ stats, float* __restrict__ partial) {
  const int g = blockIdx.y, sp = blockIdx.x, nsp = gridDim.x;
  const float* G = gt + (long long)g * pixels;
  const float* P = pred + (long long)g * pixels;
  const float mx = stats[g * kStat + 3];
  const float c = 0.2f * mx;
  float acc[3] = {0.f, 0.f, 0.f};
  for (long long i = sp * 256ll + threadIdx.x; i < pixels; i += nsp * 256ll) {
    const float gv = G[i];
    if (gv > 0.f && gv < max_depth) {
      const float r = fabsf(gv - clampf(P[i], 1e-3f, max_depth));
      if (r > c) {
        acc[0] += (r * r + c * c) / (2.f * c);
        acc[1] += 0.5f - (r * r) / (2.f * c * c);
      } else {
        acc[0] += r;
      }
      if (r == mx) acc[2] += 1.f;
    }
  }
  __shared__ float lds[3 * 16];
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0) {
    float* o = partial + ((long long)g * nsp + sp) * 4;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = 0.f;
  }
}

__global__ void masked_reduceB_kernel(const float* __restrict__ partial, int G, int nsp, float* __restrict__ stats) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < nsp; ++k) {
    const float* p = partial + ((long long)g * nsp + k) * 4;
    s0 += p[0]; s1 += p[1]; s2 += p[2];
  }
  float* o = stats + g * kStat;
  o[0] = s0; o[2] = s1; o[4] = s2;
}

// loss[0] = (accumulate ? loss[0] : 0) + weight * (1/G) sum_g L_g        (0/0 -> NaN like the mean of an empty selection)
__global__ void masked_loss_finalize_kernel(float* __restrict__ stats, int G, int kind, float weight, int accumulate, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) {
    float* st = stats + g * kStat;
    const float n = st[1];
    float L = st[0] / n;
    if (kind == DN_LOSS_SCALE_INV) L = L - (st[2] * st[2]) * 0.5f / (n * n);
    st[5] = L;
    s += L;
  }
  const float v = weight * (s / (float)G);
  loss[0] = accumulate ? loss[0] + v : v;
}

__global__ void masked_loss_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ pred, const float* __restrict__ stats,
                                       const float* __restrict__ dloss, int G, long long pixels, float max_depth, int kind, float weight,
                                       float* __restrict__ dpred) {
  const long long total = (long long)G * pixels;
  const float up = dloss[0] * weight / (float)G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i / pixels);
    const float* st = stats + g * kStat;
    const float gv = gt[i], pv = pred[i];
    float out = 0.f;
    if (gv > 0.f && gv < max_depth && pv >= 1e-3f && pv <= max_depth) {   // clamp passes gradient inside [lo, hi]
      const float n = st[1];
      const float d = pv - gv;                                             // = -(gt - p)
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);             // d|gt-p|/dp
      float fp;
      if (kind == DN_LOSS_L1) fp = sg / n;
      else if (kind == DN_LOSS_L2) fp = 2.f * d / n;
      else if (kind == DN_LOSS_SCALE_INV) fp = 2.f * d / n + st[2] / (n * n);   // st[2] = sum (gt - p)
      else {
        const float mx = st[3], c = 0.2f * mx, r = fabsf(d);
        float dr = (r > c) ? r / c : 1.f;
        if (r == mx) dr += 0.2f * st[2] / st[4];                           // through c = 0.2 max r, ties share evenly
        fp = sg * dr / n;
      }
      out = up * fp;
    }
    dpred[i] = out;
  }
}

// ------------------------------------------------------------------- ground-truth pyramids (loss_functions.py:191-215)
// one level: out[n][y][x] from the 2x2 block of in; mode 0 max_pool2d, 1 avg_pool2d (((a+b)+c)+d)/4, 2 bilinear x0.5
// align_corners=False = ((0.25a + 0.25b) + 0.25c) + 0.25d  (ATen's operation order, probed)
__global__ void pyramid_down2_kernel(const float* __restrict__ in, int N, int H, int W, int mode, float* __restrict__ out) {
  const int oh = H / 2, ow = W / 2;
  const long long total = (long long)N * oh * ow;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % ow), y = (int)((i / ow) % oh), n = (int)(i / ((long long)ow * oh));
    const float* p = in + ((long long)n * H + 2 * y) * W + 2 * x;
    const float a = p[0], b = p[1], c = p[W], d = p[W + 1];
    float v;
    if (mode == 0) v = fmaxf(fmaxf(a, b), fmaxf(c, d));
    else if (mode == 1) v = (((a + b) + c) + d) / 4.f;
    else v = ((0.25f * a + 0.25f * b) + 0.25f * c) + 0.25f * d;
    out[i] = v;
  }
}

// -------------------------------------------- integer-factor upsample of a 1-channel map (Multiscale_FULL_L1_loss :243)
// mode 0 nearest, 1 bilinear align_corners=False.  fwd gathers; bwd gathers too (each low pixel sums its dependants).
__device__ __forceinline__ void bil_src(int o, int scale, int n_in, int* i0, int* i1, float* l1) {
  float s = ((float)o + 0.5f) / (float)scale - 0.5f;
  if (s < 0.f) s = 0.f;
  const int a = (int)s;
  *i0 = a;
  *i1 = a + (a < n_in - 1 ? 1 : 0);
  *l1 = s - (float)a;
}

__global__ void upsample_int_fwd_kernel(const float* __restrict__ low, int N, int h, int w, int scale, int mode, float* __restrict__ out) {
  const int OH = h * scale, OW = w * scale;
  const long long total = (long long)N * OH * OW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % OW), y = (int)((i / OW) % OH), n = (int)(i / ((long long)OW * OH));
    const float* L = low + (long long)n * h * w;
    if (mode == 0) {
      out[i] = L[(y / scale) * w + x / scale];
    } else {
      int y0, y1, x0, x1;
      float ly, lx;
      bil_src(y, scale, h, &y0, &y1, &ly);
      bil_src(x, scale, w, &x0, &x1, &lx);
      const float hy = 1.f - ly, hx = 1.f - lx;
      out[i] = hy * (hx * L[y0 * w + x0] + lx * L[y0 * w + x1]) + ly * (hx * L[y1 * w + x0] + lx * L[y1 * w + x1]);
    }
  }
}

__global__ void upsample_int_bwd_kernel(const float* __restrict__ dout, int N, int h, int w, int scale, int mode, float* __restrict__ dlow) {
  const int OH = h * scale, OW = w * scale;
  const long long total = (long long)N * h * w;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % w), y = (int)((i / w) % h), n = (int)(i / ((long long)w * h));
    const float* D = dout + (long long)n * OH * OW;
    float s = 0.f;
    if (mode == 0) {
      for (int dy = 0; dy < scale; ++dy)
        for (int dx = 0; dx < scale; ++dx) s += D[(y * scale + dy) * OW + x * scale + dx];
    } else {
      // output rows that can touch low row y: those whose (y0, y1) contains y  ->  o in [(y-1)*scale - scale, (y+1)*scale + scale)
      const int oy_lo = max(0, (y - 1) * scale - scale), oy_hi = min(OH, (y + 2) * scale + scale);
      const int ox_lo = max(0, (x - 1) * scale - scale), ox_hi = min(OW, (x + 2) * scale + scale);
      for (int oy = oy_lo; oy < oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bil_src(oy, scale, h, &y0, &y1, &ly);
        float wy = 0.f;
        if (y0 == y) wy += 1.f - ly;
        if (y1 == y) wy += ly;
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox < ox_hi; ++ox) {
          int x0, x1;
          float lx;
          bil_src(ox, scale, w, &x0, &x1, &lx);
          float wx = 0.f;
          if (x0 == x) wx += 1.f - lx;
          if (x1 == x) wx += lx;
          if (wx != 0.f) s += wy * wx * D[oy * OW + ox];
        }
      }
    }
    dlow[i] = s;
  }
}

// ----------------------------------------------------------------- explainability_loss (loss_functions.py:357-364)
// binary_cross_entropy(mask, 1) = mean(-max(log(m), -100)); accumulated into loss[0] by the finalize
__global__ void __launch_bounds__(256) neglog_sum_kernel(const float* __restrict__ m, long long n, float* __restrict__ partial) {
  float acc[1] = {0.f};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc[0] -= fmaxf(logf(m[i]), -100.f);
  __shared__ float lds[16];
  block_sum<1>(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

// fixed-order fp64 sum of n strided floats by one block of 256 threads (see dn_warp.hip: the single-thread loops took 50-110 us)
template <int STRIDE>
__device__ __forceinline__ double loss_block_sum_f64(const float* __restrict__ v, int n, int offset, double* lds4) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[(long long)i * STRIDE + offset];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

__global__ void __launch_bounds__(256) mean_finalize_kernel(const float* __restrict__ partial, int blocks, double count, float weight,
                                                            int accumulate, float* __restrict__ loss) {
  __shared__ double lds4[4];
  const double s = loss_block_sum_f64<1>(partial, blocks, 0, lds4);
  if (threadIdx.x != 0) return;
  const float v = (float)(s / count) * weight;
  loss[0] = accumulate ? loss[0] + v : v;
}

__global__ void neglog_bwd_kernel(const float* __restrict__ m, const float* __restrict__ dloss, long long n, float* __restrict__ dm) {
  const float up = dloss[0] / (float)n;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = m[i];
    dm[i] = up * (x - 1.f) / fmaxf((1.f - x) * x, 1e-12f);      // ATen's binary_cross_entropy_backward with target = 1
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

__global__ void __launch_bounds__(256) smooth2_finalize_kernel(const float* __restrict__ partial, int blocks, int B, int H, int W, float weight,
                                                               float* __restrict__ loss) {
  __shared__ double lds4[4];
  double s[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = loss_block_sum_f64<4>(partial, blocks, k, lds4);
  if (threadIdx.x != 0) return;
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
                                                                    const float* __restrict__ medians, float* __restrict__ scratch) {
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
      float pv = clampf(p[i], 1e-3f, max_depth);
      if (medians != nullptr) pv = pv * medians[b * 2] / medians[b * 2 + 1];   // valid_pred * median(gt) / median(pred), :432-433
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

// per-sample medians of the valid gt and of the valid clamped pred (torch.median: the LOWER middle for an even count):
// exact 4-pass radix select on the float bit patterns (all values are positive, so the bits order like the values).
// one block per sample; medians[b] = (median gt, median pred)
__global__ void __launch_bounds__(kLossThreads) median_select_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int H, int W,
                                                                     float max_depth, int y1, int y2, int x1, int x2,
                                                                     float* __restrict__ medians) {
  const int b = blockIdx.x;
  const long long pixels = (long long)H * W;
  const float* g = gt + b * pixels;
  const float* p = pred + b * pixels;
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_rank;
  for (int which = 0; which < 2; ++which) {
    unsigned prefix = 0, rank = 0;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int k = threadIdx.x; k < 256; k += kLossThreads) hist[k] = 0;
      __syncthreads();
      for (long long i = threadIdx.x; i < pixels; i += kLossThreads) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float gv = g[i];
        if (gv > 0.f && gv < max_depth && y >= y1 && y < y2 && x >= x1 && x < x2) {
          const float v = which == 0 ? gv : clampf(p[i], 1e-3f, max_depth);
          const unsigned bits = __float_as_uint(v);
          if (pass == 0 || (bits >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned r = rank;
        if (pass == 0) {
          unsigned n = 0;
          for (int k = 0; k < 256; ++k) n += hist[k];
          r = n > 0 ? (n - 1) / 2 : 0;
        }
        unsigned k = 0;
        for (; k < 255; ++k) {
          if (r < hist[k]) break;
          r -= hist[k];
        }
        s_prefix = prefix | (k << shift);
        s_rank = r;
      }
      __syncthreads();
      prefix = s_prefix;
      rank = s_rank;
      __syncthreads();
    }
    if (threadIdx.x == 0) medians[b * 2 + which] = __uint_as_float(prefix);
  }
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

size_t dn_masked_loss_workspace_bytes(int32_t G, int64_t pixels) {
  if (G <= 0 || pixels <= 0) return 0;
  return (size_t)G * loss_splits(pixels) * 4 * sizeof(float);
}

int dn_masked_loss_fwd(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, float weight,
                       int32_t accumulate, float* stats, void* workspace, size_t workspace_bytes, float* loss, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && stats && loss && workspace && G > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_fwd: bad argument");
  DN_REQUIRE(kind >= DN_LOSS_L1 && kind <= DN_LOSS_SCALE_INV, DN_ERR_BAD_ARG, "dn_masked_loss_fwd: bad kind %d", kind);
  DN_REQUIRE(workspace_bytes >= dn_masked_loss_workspace_bytes(G, pixels), DN_ERR_WORKSPACE, "dn_masked_loss_fwd: workspace too small");
  hipStream_t s = as_stream(stream);
  const int nsp = loss_splits(pixels);
  float* partial = reinterpret_cast<float*>(workspace);
  DN_LAUNCH(masked_stats_kernel, dim3(nsp, G), dim3(256), 0, s, gt, pred, (long long)pixels, max_depth, kind, partial);
  DN_LAUNCH(masked_reduceA_kernel, dim3((G + 63) / 64), dim3(64), 0, s, partial, G, nsp, stats);
  if (kind == DN_LOSS_BERHU) {
    DN_LAUNCH(berhu_stats_kernel, dim3(nsp, G), dim3(256), 0, s, gt, pred, (long long)pixels, max_depth, stats, partial);
    DN_LAUNCH(masked_reduceB_kernel, dim3((G + 63) / 64), dim3(64), 0, s, partial, G, nsp, stats);
  }
  DN_LAUNCH(masked_loss_finalize_kernel, dim3(1), dim3(64), 0, s, stats, G, kind, weight, accumulate, loss);
  return check_launch("masked_loss_fwd");
}

int dn_masked_loss_fwd_fused(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, float weight,
                             int32_t accumulate, float* stats, void* workspace, size_t workspace_bytes, float* loss, int32_t* counter,
                             dn_stream_t stream) {
  DN_REQUIRE(gt && pred && stats && loss && workspace && counter && G > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_fwd_fused: bad argument");
  DN_REQUIRE((kind == DN_LOSS_L1 || kind == DN_LOSS_L2 || kind == DN_LOSS_SCALE_INV) && G <= kFoldLossGroups, DN_ERR_BAD_ARG,
             "dn_masked_loss_fwd_fused: kind %d / %d groups take dn_masked_loss_fwd", kind, G);
  DN_REQUIRE(workspace_bytes >= dn_masked_loss_workspace_bytes(G, pixels), DN_ERR_WORKSPACE, "dn_masked_loss_fwd_fused: workspace too small");
  const int nsp = loss_splits(pixels);
  DN_LAUNCH(masked_stats_fold_kernel, dim3(nsp, G), dim3(256), 0, as_stream(stream), gt, pred, (long long)pixels, max_depth, (int)kind,
            reinterpret_cast<float*>(workspace), (int)G, stats, weight, (int)accumulate, loss, reinterpret_cast<int*>(counter));
  return check_launch("masked_stats_fold_kernel");
}

// The same forward in pieces, for data-parallel runs of the whole-batch (Multiscale_*) losses: pass 0 fills stats[g][0..3]
// (sum f, n, sum d, max r) from this rank's pixels; the ranks exchange them (sums add, the maximum takes the max); for berHu pass
// 1 then fills stats[g][0], [2], [4] with the sums that depend on the global maximum (exchanged again, they add); finalize turns
// the global stats into the loss.  dn_masked_loss_bwd with those stats yields d(global loss)/d(local prediction).
int dn_masked_loss_stats(const float* gt, const float* pred, int32_t G, int64_t pixels, float max_depth, int32_t kind, int32_t pass,
                         float* stats, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && stats && workspace && G > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_stats: bad argument");
  DN_REQUIRE(kind >= DN_LOSS_L1 && kind <= DN_LOSS_SCALE_INV && (pass == 0 || (pass == 1 && kind == DN_LOSS_BERHU)), DN_ERR_BAD_ARG,
             "dn_masked_loss_stats: bad kind %d / pass %d", kind, pass);
  DN_REQUIRE(workspace_bytes >= dn_masked_loss_workspace_bytes(G, pixels), DN_ERR_WORKSPACE, "dn_masked_loss_stats: workspace too small");
  hipStream_t s = as_stream(stream);
  const int nsp = loss_splits(pixels);
  float* partial = reinterpret_cast<float*>(workspace);
  if (pass == 0) {
    DN_LAUNCH(masked_stats_kernel, dim3(nsp, G), dim3(256), 0, s, gt, pred, (long long)pixels, max_depth, kind, partial);
    DN_LAUNCH(masked_reduceA_kernel, dim3((G + 63) / 64), dim3(64), 0, s, partial, G, nsp, stats);
  } else {
    DN_LAUNCH(berhu_stats_kernel, dim3(nsp, G), dim3(256), 0, s, gt, pred, (long long)pixels, max_depth, stats, partial);
    DN_LAUNCH(masked_reduceB_kernel, dim3((G + 63) / 64), dim3(64), 0, s, partial, G, nsp, stats);
  }
  return check_launch("masked_loss_stats");
}

int dn_masked_loss_finalize(float* stats, int32_t G, int32_t kind, float weight, int32_t accumulate, float* loss, dn_stream_t stream) {
  DN_REQUIRE(stats && loss && G > 0 && kind >= DN_LOSS_L1 && kind <= DN_LOSS_SCALE_INV, DN_ERR_BAD_ARG, "dn_masked_loss_finalize: bad argument");
  DN_LAUNCH(masked_loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), stats, G, kind, weight, accumulate, loss);
  return check_launch("masked_loss_finalize_kernel");
}

int dn_masked_loss_bwd(const float* gt, const float* pred, const float* stats, const float* dloss, int32_t G, int64_t pixels,
                       float max_depth, int32_t kind, float weight, float* dpred, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && stats && dloss && dpred && G > 0 && pixels > 0, DN_ERR_BAD_ARG, "dn_masked_loss_bwd: bad argument");
  long long total = (long long)G * pixels;
  int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  DN_LAUNCH(masked_loss_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gt, pred, stats, dloss, G,
                     (long long)pixels, max_depth, kind, weight, dpred);
  return check_launch("masked_loss_bwd_kernel");
}

static inline int ew_blocks(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

int dn_pyramid_down2(const float* in, int32_t N, int32_t H, int32_t W, int32_t mode, float* out, dn_stream_t stream) {
  DN_REQUIRE(in && out && N > 0 && H >= 2 && W >= 2 && mode >= 0 && mode <= 2, DN_ERR_BAD_ARG, "dn_pyramid_down2: bad argument");
  DN_LAUNCH(pyramid_down2_kernel, dim3(ew_blocks((long long)N * (H / 2) * (W / 2))), dim3(256), 0, as_stream(stream), in, N, H, W, mode, out);
  return check_launch("pyramid_down2_kernel");
}

int dn_upsample_int_fwd(const float* low, int32_t N, int32_t h, int32_t w, int32_t scale, int32_t mode, float* out, dn_stream_t stream) {
  DN_REQUIRE(low && out && N > 0 && h > 0 && w > 0 && scale >= 1 && (mode == 0 || mode == 1), DN_ERR_BAD_ARG, "dn_upsample_int_fwd: bad argument");
  DN_LAUNCH(upsample_int_fwd_kernel, dim3(ew_blocks((long long)N * h * w * scale * scale)), dim3(256), 0, as_stream(stream), low, N, h, w, scale, mode, out);
  return check_launch("upsample_int_fwd_kernel");
}

int dn_upsample_int_bwd(const float* dout, int32_t N, int32_t h, int32_t w, int32_t scale, int32_t mode, float* dlow, dn_stream_t stream) {
  DN_REQUIRE(dout && dlow && N > 0 && h > 0 && w > 0 && scale >= 1 && (mode == 0 || mode == 1), DN_ERR_BAD_ARG, "dn_upsample_int_bwd: bad argument");
  DN_LAUNCH(upsample_int_bwd_kernel, dim3(ew_blocks((long long)N * h * w)), dim3(256), 0, as_stream(stream), dout, N, h, w, scale, mode, dlow);
  return check_launch("upsample_int_bwd_kernel");
}

int32_t dn_reduce1d_blocks(int64_t n) { return ew_blocks(n) > 1024 ? 1024 : ew_blocks(n); }

int dn_explainability_fwd(const float* mask, int64_t n, float weight, int32_t accumulate, float* partial, float* loss, dn_stream_t stream) {
  DN_REQUIRE(mask && partial && loss && n > 0, DN_ERR_BAD_ARG, "dn_explainability_fwd: bad argument");
  hipStream_t s = as_stream(stream);
  const int blocks = dn_reduce1d_blocks(n);
  DN_LAUNCH(neglog_sum_kernel, dim3(blocks), dim3(256), 0, s, mask, (long long)n, partial);
  DN_LAUNCH(mean_finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, (double)n, weight, accumulate, loss);
  return check_launch("explainability_fwd");
}

int dn_explainability_bwd(const float* mask, const float* dloss, int64_t n, float* dmask, dn_stream_t stream) {
  DN_REQUIRE(mask && dloss && dmask && n > 0, DN_ERR_BAD_ARG, "dn_explainability_bwd: bad argument");
  DN_LAUNCH(neglog_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), mask, dloss, (long long)n, dmask);
  return check_launch("neglog_bwd_kernel");
}

int32_t dn_smooth_blocks(int32_t B, int32_t H, int32_t W) { return smooth_blocks(B, H, W); }

int dn_smooth2_fwd(const float* map, int32_t B, int32_t H, int32_t W, float weight, float* partial, float* loss, dn_stream_t stream) {
  DN_REQUIRE(map && partial && loss && B > 0 && H >= 3 && W >= 3, DN_ERR_BAD_ARG, "dn_smooth2_fwd: bad argument (needs H,W >= 3)");
  hipStream_t s = as_stream(stream);
  const int blocks = smooth_blocks(B, H, W);
  DN_LAUNCH(smooth2_fwd_kernel, dim3(blocks), dim3(256), 0, s, map, B, H, W, partial);
  DN_LAUNCH(smooth2_finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, B, H, W, weight, loss);
  return check_launch("smooth2_fwd");
}

int dn_smooth2_bwd(const float* map, const float* dloss, int32_t B, int32_t H, int32_t W, float weight, float* dmap, dn_stream_t stream) {
  DN_REQUIRE(map && dloss && dmap && B > 0 && H >= 3 && W >= 3, DN_ERR_BAD_ARG, "dn_smooth2_bwd: bad argument");
  DN_LAUNCH(smooth2_bwd_kernel, dim3(smooth_blocks(B, H, W)), dim3(256), 0, as_stream(stream), map, dloss, B, H, W, weight, dmap);
  return check_launch("smooth2_bwd_kernel");
}

int dn_compute_errors(const float* gt, const float* pred, int32_t B, int32_t H, int32_t W, float max_depth, int32_t y1, int32_t y2,
                      int32_t x1, int32_t x2, int32_t median_scaling, float* scratch, float* out8, dn_stream_t stream) {
  DN_REQUIRE(gt && pred && scratch && out8 && B > 0 && H > 0 && W > 0, DN_ERR_BAD_ARG, "dn_compute_errors: bad argument");
  hipStream_t s = as_stream(stream);
  float* medians = nullptr;
  if (median_scaling) {
    medians = scratch + (size_t)B * 9;
    DN_LAUNCH(median_select_kernel, dim3(B), dim3(kLossThreads), 0, s, gt, pred, H, W, max_depth, y1, y2, x1, x2, medians);
  }
  DN_LAUNCH(errors_stats_kernel, dim3(B), dim3(kLossThreads), 0, s, gt, pred, H, W, max_depth, y1, y2, x1, x2, medians, scratch);
  DN_LAUNCH(errors_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, B, out8);
  return check_launch("compute_errors");
}

}  // extern "C"
