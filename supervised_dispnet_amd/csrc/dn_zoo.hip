// Pointwise pieces of the FCRN / ASPP nets (reference models/FCRN.py, models/ASPP.py, models/res_aspp.py), gfx950 only:
// batch statistics of a materialised tensor in the conv epilogues' partial layout, BatchNorm apply without ReLU, an element-wise
// activation, and the one-channel bilinear resize of F.interpolate (either align_corners).  HBM / latency bound, deterministic
// (gathers and fixed-order sums, no float atomics).
#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kZThreads = 256;
constexpr int kTileRows = 128;             // = kBnTileRows of dn_pointwise.hip: the row tile dn_bn_finalize merges

inline int zblocks(long long n) {
  long long b = (n + kZThreads - 1) / kZThreads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
}  // namespace

// one block per (row tile, 64-channel group): thread (c, q) sums rows q, q+4, ... of its channel, the four partial sums meet in LDS;
// second pass about the tile's own mean (the tile is re-read from L2)
__global__ void __launch_bounds__(kZThreads) bn_stats_partial_kernel(const float* __restrict__ x, long long rows, int C, float* __restrict__ partial) {
  __shared__ float red[4][64];
  __shared__ float mean_s[64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * kTileRows;
  const long long left = rows - r0;
  const int n = left < kTileRows ? (int)left : kTileRows;
  float s = 0.f;
  if (c < C)
    for (int r = q; r < n; r += 4) s += x[(r0 + r) * C + c];
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  float tot = 0.f;
  if (q == 0) {
    tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    mean_s[threadIdx.x] = tot / (float)n;
  }
  __syncthreads();
  const float mu = mean_s[threadIdx.x & 63];
  float m2 = 0.f;
  if (c < C)
    for (int r = q; r < n; r += 4) {
      const float d = x[(r0 + r) * C + c] - mu;
      m2 += d * d;
    }
  __syncthreads();
  red[q][threadIdx.x & 63] = m2;
  __syncthreads();
  if (q == 0 && c < C) {
    float* dst = partial + ((long long)blockIdx.x * C + c) * 2;
    dst[0] = tot;
    dst[1] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

__global__ void __launch_bounds__(kZThreads) bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             long long rows, int C, float* __restrict__ out) {
  const int G = C / 4;
  const long long total = rows * G;
  for (long long i = blockIdx.x * (long long)kZThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kZThreads) {
    const int g = (int)(i % G);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * g), sh = *reinterpret_cast<const f32x4*>(shift + 4 * g);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * sc[e] + sh[e];
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

__global__ void __launch_bounds__(kZThreads) act_fwd_kernel(const float* __restrict__ x, long long n, int act, float p0, float p1, float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)kZThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kZThreads) {
    const float v = x[i];
    float o = v;
    switch (act) {
      case DN_ACT_RELU: o = v > 0.f ? v : 0.f; break;
      case DN_ACT_LEAKY: o = v > 0.f ? v : v * p0; break;
      case DN_ACT_ELU: o = v > 0.f ? v : (expf(v) - 1.f); break;
      case DN_ACT_SIGMOID_AFFINE: o = p0 / (1.f + expf(-v)) + p1; break;
      default: break;
    }
    out[i] = o;
  }
}


// x[n][2y+a][2x+b][c] += bias4[a][b][c]: FCRN's up-projection interleaves four convolutions that each carry their own bias
// (models/FCRN.py:56-64,97-107); the engine runs the four as ONE 6x6 / stride-2 transposed convolution and adds the biases here
__global__ void __launch_bounds__(kZThreads) phase_bias_add_kernel(float* __restrict__ x, int N, int H2, int W2, int C, const float* __restrict__ bias4) {
  const int G = C / 4;
  const long long total = (long long)N * H2 * W2 * G;
  for (long long i = blockIdx.x * (long long)kZThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kZThreads) {
    const int g = (int)(i % G);
    const long long pix = i / G;
    const int xx = (int)(pix % W2), yy = (int)((pix / W2) % H2);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias4 + (((yy & 1) * 2 + (xx & 1)) * C) + 4 * g);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += b[e];
    *reinterpret_cast<f32x4*>(x + i * 4) = v;
  }
}

// out4[a][b][c] = sum over n, y, x of g[n][2y+a][2x+b][c]: the four bias gradients.  One block per (phase, 64-channel group, pixel
// slice); slices are summed in a fixed order by the caller-visible second stage below (deterministic).
constexpr int kPhaseSlices = 32;
__global__ void __launch_bounds__(kZThreads) phase_colsum_kernel(const float* __restrict__ g, int N, int H, int W, int C, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int ph = blockIdx.x, a = ph >> 1, b = ph & 1;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const long long npix = (long long)N * H * W;
  const long long per = (npix + kPhaseSlices - 1) / kPhaseSlices;
  const long long p0 = (long long)blockIdx.z * per, p1 = p0 + per < npix ? p0 + per : npix;
  float s = 0.f;
  if (c < C)
    for (long long pix = p0 + q; pix < p1; pix += 4) {
      const int xx = (int)(pix % W);
      const long long t = pix / W;
      const int yy = (int)(t % H), n = (int)(t / H);
      s += g[(((long long)n * 2 * H + 2 * yy + a) * 2 * W + 2 * xx + b) * C + c];
    }
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < C) part[((long long)blockIdx.z * 4 + ph) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void phase_colsum_finalize_kernel(const float* __restrict__ part, int C, float* __restrict__ out4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * C) return;
  float s = 0.f;
  for (int z = 0; z < kPhaseSlices; ++z) s += part[(long long)z * 4 * C + i];
  out4[i] = s;
}

// ATen's area_pixel_compute_scale / area_pixel_compute_source_index / guard_index_and_lambda for mode = 'bilinear', in fp32
struct LinMap {
  float scale;
  int in, align;
  __device__ __forceinline__ void at(int dst, int* i0, int* i1, float* l1) const {
    float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
    if (!align && src < 0.f) src = 0.f;
    int a = (int)src;
    if (a > in - 1) a = in - 1;
    float l = src - (float)a;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
    *i0 = a;
    *i1 = a + (a < in - 1 ? 1 : 0);
    *l1 = l;
  }
};

static inline float lin_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}

__global__ void __launch_bounds__(kZThreads) resize_bilinear_fwd_kernel(const float* __restrict__ in, int N, int IH, int IW, int OH, int OW, LinMap my,
                                                                        LinMap mx, float* __restrict__ out) {
  const long long total = (long long)N * OH * OW;
  for (long long i = blockIdx.x * (long long)kZThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kZThreads) {
    const int ox = (int)(i % OW);
    const long long t = i / OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    my.at(oy, &y0, &y1, &ly);
    mx.at(ox, &x0, &x1, &lx);
    const float* p = in + (long long)n * IH * IW;
    const float v00 = p[(long long)y0 * IW + x0], v01 = p[(long long)y0 * IW + x1], v10 = p[(long long)y1 * IW + x0], v11 = p[(long long)y1 * IW + x1];
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

// din[iy][ix] = sum over the output pixels whose two-point stencils touch (iy, ix): the source index is monotone in the output index,
// so the candidates form a short run around iy / scale; each thread re-derives the stencil of every candidate (a gather: fixed order)
__global__ void __launch_bounds__(kZThreads) resize_bilinear_bwd_kernel(const float* __restrict__ dout, int N, int IH, int IW, int OH, int OW, LinMap my,
                                                                        LinMap mx, float inv_sy, float inv_sx, float* __restrict__ din, int accumulate) {
  const long long total = (long long)N * IH * IW;
  for (long long i = blockIdx.x * (long long)kZThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kZThreads) {
    const int ix = (int)(i % IW);
    const long long t = i / IW;
    const int iy = (int)(t % IH), n = (int)(t / IH);
    // (half-pixel centres shift the run by up to 0.5 / scale output pixels)
    const int mgy = (int)ceilf(0.5f * inv_sy) + 2, mgx = (int)ceilf(0.5f * inv_sx) + 2;
    int oy_lo = (int)floorf(((float)iy - 1.f) * inv_sy) - mgy, oy_hi = (int)ceilf(((float)iy + 1.f) * inv_sy) + mgy;
    int ox_lo = (int)floorf(((float)ix - 1.f) * inv_sx) - mgx, ox_hi = (int)ceilf(((float)ix + 1.f) * inv_sx) + mgx;
    if (my.scale == 0.f) { oy_lo = 0; oy_hi = OH - 1; }
    if (mx.scale == 0.f) { ox_lo = 0; ox_hi = OW - 1; }
    oy_lo = oy_lo < 0 ? 0 : oy_lo;
    ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi;
    ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
    const float* g = dout + (long long)n * OH * OW;
    float s = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      my.at(oy, &y0, &y1, &ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        mx.at(ox, &x0, &x1, &lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx != 0.f) row += wx * g[(long long)oy * OW + ox];
      }
      s += wy * row;
    }
    din[i] = accumulate ? din[i] + s : s;
  }
}

}  // namespace dn

using namespace dn;

extern "C" {

int32_t dn_bn_stats_rows(int64_t rows) { return (int32_t)((rows + kTileRows - 1) / kTileRows); }

int dn_bn_stats_partial(const float* x, int64_t rows, int32_t C, float* partial, dn_stream_t stream) {
  DN_REQUIRE(x && partial && rows > 0 && C > 0, DN_ERR_BAD_ARG, "dn_bn_stats_partial: bad argument");
  DN_REQUIRE((rows + kTileRows - 1) / kTileRows < 65536ll * 32768ll, DN_ERR_UNSUPPORTED, "dn_bn_stats_partial: too many rows");
  DN_LAUNCH(bn_stats_partial_kernel, dim3((unsigned)((rows + kTileRows - 1) / kTileRows), (C + 63) / 64), dim3(kZThreads), 0, as_stream(stream), x,
                     (long long)rows, C, partial);
  return check_launch("bn_stats_partial_kernel");
}

int dn_bn_apply_fwd(const float* y, const float* scale, const float* shift, int64_t rows, int32_t C, float* out, dn_stream_t stream) {
  DN_REQUIRE(y && scale && shift && out && rows > 0 && C > 0, DN_ERR_BAD_ARG, "dn_bn_apply_fwd: bad argument");
  DN_REQUIRE(C % 4 == 0, DN_ERR_UNSUPPORTED, "dn_bn_apply_fwd: need C%%4==0");
  DN_LAUNCH(bn_apply_kernel, dim3(zblocks(rows * (C / 4))), dim3(kZThreads), 0, as_stream(stream), y, scale, shift, (long long)rows, C, out);
  return check_launch("bn_apply_kernel");
}

int dn_act_fwd(const float* x, int64_t n, int32_t act, float p0, float p1, float* out, dn_stream_t stream) {
  DN_REQUIRE(x && out && n > 0 && act >= DN_ACT_NONE && act <= DN_ACT_SIGMOID_AFFINE, DN_ERR_BAD_ARG, "dn_act_fwd: bad argument");
  DN_LAUNCH(act_fwd_kernel, dim3(zblocks(n)), dim3(kZThreads), 0, as_stream(stream), x, (long long)n, act, p0, p1, out);
  return check_launch("act_fwd_kernel");
}


int dn_phase_bias_add(float* x, int32_t N, int32_t H2, int32_t W2, int32_t C, const float* bias4, dn_stream_t stream) {
  DN_REQUIRE(x && bias4 && N > 0 && H2 > 0 && W2 > 0 && C > 0, DN_ERR_BAD_ARG, "dn_phase_bias_add: bad argument");
  DN_REQUIRE(C % 4 == 0 && H2 % 2 == 0 && W2 % 2 == 0, DN_ERR_UNSUPPORTED, "dn_phase_bias_add: need C%%4==0 and even extents");
  DN_LAUNCH(phase_bias_add_kernel, dim3(zblocks((long long)N * H2 * W2 * (C / 4))), dim3(kZThreads), 0, as_stream(stream), x, N, H2, W2, C, bias4);
  return check_launch("phase_bias_add_kernel");
}

size_t dn_phase_colsum_workspace_bytes(int32_t C) { return (size_t)kPhaseSlices * 4 * (size_t)(C > 0 ? C : 0) * sizeof(float); }

int dn_phase_colsum(const float* g, int32_t N, int32_t H, int32_t W, int32_t C, float* workspace, float* out4, dn_stream_t stream) {
  DN_REQUIRE(g && workspace && out4 && N > 0 && H > 0 && W > 0 && C > 0, DN_ERR_BAD_ARG, "dn_phase_colsum: bad argument");
  DN_LAUNCH(phase_colsum_kernel, dim3(4, (C + 63) / 64, kPhaseSlices), dim3(kZThreads), 0, as_stream(stream), g, N, H, W, C, workspace);
  int rc = check_launch("phase_colsum_kernel");
  if (rc != DN_OK) return rc;
  DN_LAUNCH(phase_colsum_finalize_kernel, dim3((4 * C + 255) / 256), dim3(256), 0, as_stream(stream), workspace, C, out4);
  return check_launch("phase_colsum_finalize_kernel");
}

int dn_resize_bilinear_fwd(const float* in, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t align_corners, float* out,
                           dn_stream_t stream) {
  DN_REQUIRE(in && out && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, DN_ERR_BAD_ARG, "dn_resize_bilinear_fwd: bad argument");
  const LinMap my{lin_scale(IH, OH, align_corners), IH, align_corners ? 1 : 0}, mx{lin_scale(IW, OW, align_corners), IW, align_corners ? 1 : 0};
  DN_LAUNCH(resize_bilinear_fwd_kernel, dim3(zblocks((long long)N * OH * OW)), dim3(kZThreads), 0, as_stream(stream), in, N, IH, IW, OH, OW, my,
                     mx, out);
  return check_launch("resize_bilinear_fwd_kernel");
}

int dn_resize_bilinear_bwd(const float* dout, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t align_corners, float* din,
                           int32_t accumulate, dn_stream_t stream) {
  DN_REQUIRE(dout && din && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, DN_ERR_BAD_ARG, "dn_resize_bilinear_bwd: bad argument");
  const LinMap my{lin_scale(IH, OH, align_corners), IH, align_corners ? 1 : 0}, mx{lin_scale(IW, OW, align_corners), IW, align_corners ? 1 : 0};
  const float isy = my.scale > 0.f ? 1.f / my.scale : 0.f, isx = mx.scale > 0.f ? 1.f / mx.scale : 0.f;
  DN_LAUNCH(resize_bilinear_bwd_kernel, dim3(zblocks((long long)N * IH * IW)), dim3(kZThreads), 0, as_stream(stream), dout, N, IH, IW, OH, OW, my,
                     mx, isy, isx, din, accumulate);
  return check_launch("resize_bilinear_bwd_kernel");
}

}  // extern "C"
