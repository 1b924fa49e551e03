// Geometry / photometric family (reference: inverse_warp.py, loss_functions.py:317-354, layers.py:199-245).
// All kernels are HBM / gather-latency bound: one thread per target pixel, the 3x4 projection of its sample in SGPR-like
// uniform registers, 4-tap bilinear gather of the 3 image planes, wavefront-shuffle reductions for the per-sample pose
// gradients and the loss sums.  No atomics (deterministic), no host synchronisation.
#include "dn_internal.h"

namespace dn {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of NV per-thread values into out[0..NV) by thread 0 (256-thread blocks)
template <int NV>
__device__ __forceinline__ void block_sum256(float (&v)[NV], float* lds /* [NV*4] */, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wsum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i * 4 + wave] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = (lds[i * 4] + lds[i * 4 + 1]) + (lds[i * 4 + 2] + lds[i * 4 + 3]);
}

// ------------------------------------------------------------------------------------------------ pose -> projection
__device__ void mat3_mul(const float* a, const float* b, float* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__device__ void euler_mats(const float* ang, float* X, float* Y, float* Z) {
  const float cx = cosf(ang[0]), sx = sinf(ang[0]), cy = cosf(ang[1]), sy = sinf(ang[1]), cz = cosf(ang[2]), sz = sinf(ang[2]);
  const float x[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  const float y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  const float z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) { X[i] = x[i]; Y[i] = y[i]; Z[i] = z[i]; }
}

__device__ void quat_rot(const float* q /*w x y z, normalised*/, float* R) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z, wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
}

// one thread per sample.  K_s = rows 0,1 of K divided by downscale; Kinv_s = columns 0,1 of Kinv multiplied by it
// (loss_functions.py:328-329).  proj = K_s @ [R|t]  (inverse_warp.py:185-188).  pose: (tx,ty,tz,rx,ry,rz), stride pose_sb.
__global__ void pose_proj_fwd_kernel(const float* __restrict__ pose, long long pose_sb, const float* __restrict__ K,
                                     const float* __restrict__ Kinv, int B, int rot_mode, float downscale, float* __restrict__ proj,
                                     float* __restrict__ kinv_s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* v = pose + b * pose_sb;
  float R[9];
  if (rot_mode == 0) {
    float X[9], Y[9], Z[9], XY[9];
    euler_mats(v + 3, X, Y, Z);
    mat3_mul(X, Y, XY);
    mat3_mul(XY, Z, R);
  } else {
    float q[4] = {1.f, v[3], v[4], v[5]};
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    quat_rot(q, R);
  }
  float Ks[9];
  for (int i = 0; i < 9; ++i) Ks[i] = (i < 6) ? K[b * 9 + i] / downscale : K[b * 9 + i];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) proj[b * 12 + r * 4 + c] = Ks[r * 3] * R[c] + Ks[r * 3 + 1] * R[3 + c] + Ks[r * 3 + 2] * R[6 + c];
    proj[b * 12 + r * 4 + 3] = Ks[r * 3] * v[0] + Ks[r * 3 + 1] * v[1] + Ks[r * 3 + 2] * v[2];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) kinv_s[b * 9 + r * 3 + c] = (c < 2) ? Kinv[b * 9 + r * 3 + c] * downscale : Kinv[b * 9 + r * 3 + c];
}

// dproj partials [B][nblk][12] -> dpose[b] (6 floats, stride dpose_sb), overwrite or accumulate
__global__ void pose_proj_bwd_kernel(const float* __restrict__ pose, long long pose_sb, const float* __restrict__ K, int B, int rot_mode,
                                     float downscale, const float* __restrict__ dproj_partial, int nblk, float* __restrict__ dpose,
                                     long long dpose_sb, int accumulate) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float dP[12];
  for (int i = 0; i < 12; ++i) dP[i] = 0.f;
  for (int k = 0; k < nblk; ++k)
    for (int i = 0; i < 12; ++i) dP[i] += dproj_partial[((long long)b * nblk + k) * 12 + i];
  float Ks[9];
  for (int i = 0; i < 9; ++i) Ks[i] = (i < 6) ? K[b * 9 + i] / downscale : K[b * 9 + i];
  // dT = Ks^T dP  (3x4)
  float dT[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) dT[r * 4 + c] = Ks[r] * dP[c] + Ks[3 + r] * dP[4 + c] + Ks[6 + r] * dP[8 + c];
  const float* v = pose + b * pose_sb;
  float g[6];
  g[0] = dT[3]; g[1] = dT[7]; g[2] = dT[11];
  float dR[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) dR[r * 3 + c] = dT[r * 4 + c];
  if (rot_mode == 0) {
    float X[9], Y[9], Z[9], T1[9], T2[9];
    euler_mats(v + 3, X, Y, Z);
    const float cx = cosf(v[3]), sx = sinf(v[3]), cy = cosf(v[4]), sy = sinf(v[4]), cz = cosf(v[5]), sz = sinf(v[5]);
    const float dX[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
    const float dY[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
    const float dZ[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    auto dot9 = [&](const float* m) { float s = 0.f; for (int i = 0; i < 9; ++i) s += dR[i] * m[i]; return s; };
    mat3_mul(dX, Y, T1); mat3_mul(T1, Z, T2); g[3] = dot9(T2);
    mat3_mul(X, dY, T1); mat3_mul(T1, Z, T2); g[4] = dot9(T2);
    mat3_mul(X, Y, T1);  mat3_mul(T1, dZ, T2); g[5] = dot9(T2);
  } else {
    float u[4] = {1.f, v[3], v[4], v[5]};
    const float n = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
    float q[4];
    for (int i = 0; i < 4; ++i) q[i] = u[i] / n;
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    // dL/dq from R(q)
    float dq[4];
    dq[0] = 2.f * (w * (dR[0] + dR[4] + dR[8]) + z * (dR[3] - dR[1]) + y * (dR[2] - dR[6]) + x * (dR[7] - dR[5]));
    dq[1] = 2.f * (x * (dR[0] - dR[4] - dR[8]) + y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) + w * (dR[7] - dR[5]));
    dq[2] = 2.f * (y * (-dR[0] + dR[4] - dR[8]) + x * (dR[1] + dR[3]) + w * (dR[2] - dR[6]) + z * (dR[5] + dR[7]));
    dq[3] = 2.f * (z * (-dR[0] - dR[4] + dR[8]) + w * (dR[3] - dR[1]) + x * (dR[2] + dR[6]) + y * (dR[5] + dR[7]));
    float dotp = 0.f;
    for (int i = 0; i < 4; ++i) dotp += q[i] * dq[i];
    for (int i = 1; i < 4; ++i) g[2 + i] = (dq[i] - q[i] * dotp) / n;
  }
  float* o = dpose + b * dpose_sb;
  for (int i = 0; i < 6; ++i) o[i] = accumulate ? o[i] + g[i] : g[i];
}

// --------------------------------------------------------------------------------------------------- warp (+ photometric)
struct WarpArgs {
  const float* img;      // [B][3][h][w] source image (planar)
  const float* depth;    // [B][h][w]
  const float* proj;     // [B][12]
  const float* kinv;     // [B][9]
  int B, h, w;
  int padding;           // 0 zeros, 1 border
  int align;             // grid_sample align_corners
  // photometric extras (nullable)
  const float* tgt;      // [B][3][h][w]
  const float* mask;     // explainability mask element (b, y, x) at mask + b*mask_sb + y*w + x
  long long mask_sb;
};

struct WarpPix {
  float cx, cy, cz;      // Kinv @ (j, i, 1)
  float X, Y, Z, Zc;
  float xn, yn;          // after the zeros-padding substitution
  bool xdead, ydead;     // coordinate replaced by 2 (no gradient)
  float ix, iy;          // un-normalised (after border clipping)
  float gx_mult, gy_mult;  // d ix / d xn (0 when clipped)
  int x0, y0;
  float v[3];            // warped values
  float tap[4][3];       // nw, ne, sw, se values (0 when out of bounds)
};

__device__ __forceinline__ void warp_pixel(const WarpArgs& a, int b, int i, int j, WarpPix* q) {
  const float* Ki = a.kinv + b * 9;
  const float* P = a.proj + b * 12;
  const float fj = (float)j, fi = (float)i;
  q->cx = Ki[0] * fj + Ki[1] * fi + Ki[2];
  q->cy = Ki[3] * fj + Ki[4] * fi + Ki[5];
  q->cz = Ki[6] * fj + Ki[7] * fi + Ki[8];
  const float d = a.depth[((long long)b * a.h + i) * a.w + j];
  const float c0 = q->cx * d, c1 = q->cy * d, c2 = q->cz * d;
  q->X = P[0] * c0 + P[1] * c1 + P[2] * c2 + P[3];
  q->Y = P[4] * c0 + P[5] * c1 + P[6] * c2 + P[7];
  q->Z = P[8] * c0 + P[9] * c1 + P[10] * c2 + P[11];
  q->Zc = fmaxf(q->Z, 1e-3f);
  float xn = 2.f * (q->X / q->Zc) / (float)(a.w - 1) - 1.f;
  float yn = 2.f * (q->Y / q->Zc) / (float)(a.h - 1) - 1.f;
  q->xdead = q->ydead = false;
  if (a.padding == 0) {
    if (xn > 1.f || xn < -1.f) { xn = 2.f; q->xdead = true; }
    if (yn > 1.f || yn < -1.f) { yn = 2.f; q->ydead = true; }
  }
  q->xn = xn;
  q->yn = yn;
  float ix, iy, mx, my;
  if (a.align) {
    ix = ((xn + 1.f) / 2.f) * (float)(a.w - 1);
    iy = ((yn + 1.f) / 2.f) * (float)(a.h - 1);
    mx = (float)(a.w - 1) / 2.f;
    my = (float)(a.h - 1) / 2.f;
  } else {
    ix = ((xn + 1.f) * (float)a.w - 1.f) / 2.f;
    iy = ((yn + 1.f) * (float)a.h - 1.f) / 2.f;
    mx = (float)a.w / 2.f;
    my = (float)a.h / 2.f;
  }
  if (a.padding == 1) {   // clip_coordinates_set_grad
    if (ix <= 0.f) { ix = 0.f; mx = 0.f; } else if (ix >= (float)(a.w - 1)) { ix = (float)(a.w - 1); mx = 0.f; }
    if (iy <= 0.f) { iy = 0.f; my = 0.f; } else if (iy >= (float)(a.h - 1)) { iy = (float)(a.h - 1); my = 0.f; }
  }
  q->ix = ix; q->iy = iy; q->gx_mult = mx; q->gy_mult = my;
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  // huge coordinates (|ix| beyond int range) are out of bounds for every tap
  const bool sane = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  const int x0 = sane ? (int)fx0 : -10, y0 = sane ? (int)fy0 : -10;
  q->x0 = x0; q->y0 = y0;
  const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
  const float* I = a.img + (long long)b * 3 * a.h * a.w;
  const long long plane = (long long)a.h * a.w;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const bool in = sane && xx >= 0 && xx < a.w && yy >= 0 && yy < a.h;
#pragma unroll
    for (int c = 0; c < 3; ++c) q->tap[t][c] = in ? I[c * plane + (long long)yy * a.w + xx] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    q->v[c] = q->tap[0][c] * (wx0 * wy0) + q->tap[1][c] * (wx1 * wy0) + q->tap[2][c] * (wx0 * wy1) + q->tap[3][c] * (wx1 * wy1);
}

// grid (blocks_per_sample, B).  Writes `warped` when given; with a.tgt set also the per-block sum of |diff| (photometric).
__global__ void __launch_bounds__(256) warp_fwd_kernel(const WarpArgs a, float* __restrict__ warped, float* __restrict__ partial) {
  const int b = blockIdx.y;
  const long long plane = (long long)a.h * a.w;
  float acc[1] = {0.f};
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < plane; p += (long long)gridDim.x * 256) {
    const int i = (int)(p / a.w), j = (int)(p % a.w);
    WarpPix q;
    warp_pixel(a, b, i, j, &q);
    if (warped != nullptr)
#pragma unroll
      for (int c = 0; c < 3; ++c) warped[((long long)b * 3 + c) * plane + p] = q.v[c];
    if (a.tgt != nullptr) {
      const float oob = (q.v[0] == 0.f && q.v[1] == 0.f && q.v[2] == 0.f) ? 0.f : 1.f;
      const float m = a.mask != nullptr ? a.mask[b * a.mask_sb + p] : 1.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[0] += fabsf((a.tgt[((long long)b * 3 + c) * plane + p] - q.v[c]) * oob * m);
    }
  }
  if (partial != nullptr) {
    __shared__ float lds[4];
    block_sum256<1>(acc, lds, partial + (long long)b * gridDim.x + blockIdx.x);
  }
}

// Fixed-order fp64 sum of n floats by one block of 256 threads (thread t takes elements t, t + 256, ...; then a shuffle tree and four
// wave totals added in index order): the single-thread loop this replaces walked up to 6656 dependent loads (53-112 us per launch,
// 0.42 ms of a photometric step -- profiles/r04_a_photo128_kernel_stats.csv).  Result valid in thread 0.
template <int STRIDE>
__device__ __forceinline__ double block_sum_f64(const float* __restrict__ v, int n, int offset, double* lds4) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[(long long)i * STRIDE + offset];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// loss[0] (+)= weight * sum(partial) / count
__global__ void __launch_bounds__(256) sum_finalize_kernel(const float* __restrict__ partial, int n, double count, float weight, int accumulate,
                                                           float* __restrict__ loss) {
  __shared__ double lds4[4];
  const double s = block_sum_f64<1>(partial, n, 0, lds4);
  if (threadIdx.x != 0) return;
  const float v = (float)(s / count) * weight;
  loss[0] = accumulate ? loss[0] + v : v;
}

// Backward.  Upstream gradient: `dwarped` [B][3][h][w] (inverse_warp) or, when a.tgt is set, generated in place from the
// photometric loss: d mean|diff| / d warped_c = -sign(diff_c) * oob * mask / count, scaled by dloss[0] * weight.
// Outputs: ddepth (overwrite / accumulate), dproj partial [B][gridDim.x][12], optional dmask (photometric, overwrite).
__global__ void __launch_bounds__(256) warp_bwd_kernel(const WarpArgs a, const float* __restrict__ dwarped, const float* __restrict__ dloss,
                                                       float scale, float* __restrict__ ddepth, int accumulate_depth,
                                                       float* __restrict__ dproj_partial, float* __restrict__ dmask, long long dmask_sb) {
  const int b = blockIdx.y;
  const long long plane = (long long)a.h * a.w;
  const float* P = a.proj + b * 12;
  const float up = (dloss != nullptr ? dloss[0] : 1.f) * scale;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < plane; p += (long long)gridDim.x * 256) {
    const int i = (int)(p / a.w), j = (int)(p % a.w);
    WarpPix q;
    warp_pixel(a, b, i, j, &q);
    float g[3];
    if (a.tgt != nullptr) {
      const float oob = (q.v[0] == 0.f && q.v[1] == 0.f && q.v[2] == 0.f) ? 0.f : 1.f;
      const float m = a.mask != nullptr ? a.mask[b * a.mask_sb + p] : 1.f;
      float dm = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float e = a.tgt[((long long)b * 3 + c) * plane + p] - q.v[c];
        const float diff = e * oob * m;
        const float s = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        g[c] = -s * oob * m * up;
        dm += s * e * oob * up;
      }
      if (dmask != nullptr) dmask[b * dmask_sb + p] = dm;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = dwarped[((long long)b * 3 + c) * plane + p] * up;
    }
    // d out / d (ix, iy)
    const float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
    const float wx1 = q.ix - fx0, wx0 = (fx0 + 1.f) - q.ix, wy1 = q.iy - fy0, wy0 = (fy0 + 1.f) - q.iy;
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gix += g[c] * (-q.tap[0][c] * wy0 + q.tap[1][c] * wy0 - q.tap[2][c] * wy1 + q.tap[3][c] * wy1);
      giy += g[c] * (-q.tap[0][c] * wx0 - q.tap[1][c] * wx1 + q.tap[2][c] * wx0 + q.tap[3][c] * wx1);
    }
    const float gxn = q.xdead ? 0.f : gix * q.gx_mult;
    const float gyn = q.ydead ? 0.f : giy * q.gy_mult;
    // xn = 2 (X/Zc)/(w-1) - 1
    const float kx = 2.f / (float)(a.w - 1), ky = 2.f / (float)(a.h - 1);
    const float dX = gxn * kx / q.Zc;
    const float dY = gyn * ky / q.Zc;
    float dZ = -(gxn * kx * q.X + gyn * ky * q.Y) / (q.Zc * q.Zc);
    if (!(q.Z >= 1e-3f)) dZ = 0.f;                                   // clamp(min) passes gradient only where Z >= min
    const float d = a.depth[(long long)b * plane + p];
    const float c0 = q.cx * d, c1 = q.cy * d, c2 = q.cz * d;
    // dcam = Prot^T dXYZ ; ddepth = dcam . (Kinv pix)
    const float dc0 = P[0] * dX + P[4] * dY + P[8] * dZ;
    const float dc1 = P[1] * dX + P[5] * dY + P[9] * dZ;
    const float dc2 = P[2] * dX + P[6] * dY + P[10] * dZ;
    const float dd = dc0 * q.cx + dc1 * q.cy + dc2 * q.cz;
    float* o = ddepth + (long long)b * plane + p;
    *o = accumulate_depth ? *o + dd : dd;
    acc[0] += dX * c0; acc[1] += dX * c1; acc[2] += dX * c2; acc[3] += dX;
    acc[4] += dY * c0; acc[5] += dY * c1; acc[6] += dY * c2; acc[7] += dY;
    acc[8] += dZ * c0; acc[9] += dZ * c1; acc[10] += dZ * c2; acc[11] += dZ;
  }
  __shared__ float lds[12 * 4];
  block_sum256<12>(acc, lds, dproj_partial + ((long long)b * gridDim.x + blockIdx.x) * 12);
}

// F.interpolate(mode='area') to an exact 1/f size = f x f mean (row-major sum, one division), [N][C][H][W] planar
__global__ void area_down_kernel(const float* __restrict__ in, long long planes, int H, int W, int f, float* __restrict__ out) {
  const int oh = H / f, ow = W / f;
  const long long total = planes * oh * ow;
  const float cnt = (float)(f * f);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const long long pl = i / ((long long)ow * oh);
    const float* p = in + (pl * H + (long long)y * f) * W + (long long)x * f;
    float s = 0.f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += p[dy * W + dx];
    out[i] = s / cnt;
  }
}

// ------------------------------------------------------------------------------------------------------- SSIM
// layers.py:215-245: ReflectionPad2d(1), 3x3 mean; out = clamp((1 - n/d)/2, 0, 1) per element of [planes][H][W]
__device__ __forceinline__ int reflect1(int v, int n) { return v < 0 ? -v : (v >= n ? 2 * n - 2 - v : v); }

struct SsimLocal {
  float mx, my, sx, sy, sxy, n, d;
};

__device__ __forceinline__ SsimLocal ssim_stats(const float* X, const float* Y, int H, int W, int y, int x) {
  float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = reflect1(y + dy, H), xx = reflect1(x + dx, W);
      const float a = X[yy * W + xx], b = Y[yy * W + xx];
      sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
    }
  SsimLocal r;
  r.mx = sx / 9.f;
  r.my = sy / 9.f;
  r.sx = sxx / 9.f - r.mx * r.mx;
  r.sy = syy / 9.f - r.my * r.my;
  r.sxy = sxy / 9.f - r.mx * r.my;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  r.n = (2.f * r.mx * r.my + C1) * (2.f * r.sxy + C2);
  r.d = (r.mx * r.mx + r.my * r.my + C1) * (r.sx + r.sy + C2);
  return r;
}

__global__ void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long long planes, int H, int W,
                                float* __restrict__ out) {
  const long long plane = (long long)H * W, total = planes * plane;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pl = i / plane;
    const int p = (int)(i - pl * plane);
    const SsimLocal s = ssim_stats(x + pl * plane, y + pl * plane, H, W, p / W, p % W);
    out[i] = fminf(fmaxf((1.f - s.n / s.d) / 2.f, 0.f), 1.f);
  }
}

// pass 1 of the backward: per window centre the five coefficients  g * d out / d (mu_x, mu_y, E[xx], E[yy], E[xy])
__global__ void ssim_bwd_coef_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                     long long planes, int H, int W, float* __restrict__ coef /* [5][planes*H*W] */) {
  const long long plane = (long long)H * W, total = planes * plane;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pl = i / plane;
    const int p = (int)(i - pl * plane);
    const SsimLocal s = ssim_stats(x + pl * plane, y + pl * plane, H, W, p / W, p % W);
    const float raw = (1.f - s.n / s.d) / 2.f;
    float up = (raw >= 0.f && raw <= 1.f) ? g[i] : 0.f;       // clamp passes gradient inside [0, 1]
    // out = (1 - n/d)/2 ;  dn, dd
    const float dn = -0.5f / s.d * up, dd = 0.5f * s.n / (s.d * s.d) * up;
    const float A1 = 2.f * s.mx * s.my + C1, A2 = 2.f * s.sxy + C2, B1 = s.mx * s.mx + s.my * s.my + C1, B2 = s.sx + s.sy + C2;
    // n = A1*A2, d = B1*B2
    const float dA1 = dn * A2, dA2 = dn * A1, dB1 = dd * B2, dB2 = dd * B1;
    // sxy = Exy - mx*my ; sx = Exx - mx^2 ; sy = Eyy - my^2
    const float dsxy = 2.f * dA2, dsx = dB2, dsy = dB2;
    const float dmx = dA1 * 2.f * s.my + dB1 * 2.f * s.mx - dsxy * s.my - dsx * 2.f * s.mx;
    const float dmy = dA1 * 2.f * s.mx + dB1 * 2.f * s.my - dsxy * s.mx - dsy * 2.f * s.my;
    coef[i] = dmx;
    coef[total + i] = dmy;
    coef[2 * total + i] = dsx;
    coef[3 * total + i] = dsy;
    coef[4 * total + i] = dsxy;
  }
}

// pass 2: every pixel q collects from the window centres whose (reflection-padded) 3x3 window covers it
__global__ void ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef,
                                       long long planes, int H, int W, float* __restrict__ dx, float* __restrict__ dy) {
  const long long plane = (long long)H * W, total = planes * plane;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pl = i / plane;
    const int p = (int)(i - pl * plane);
    const int qy = p / W, qx = p % W;
    const float xv = x[i], yv = y[i];
    const float* c = coef + pl * plane;
    // padded coordinates that map onto q: q itself, -1 (if q == 1), H (if q == H-2)
    int uy[3], ux[3], ny = 0, nx = 0;
    uy[ny++] = qy; if (qy == 1) uy[ny++] = -1; if (qy == H - 2) uy[ny++] = H;
    ux[nx++] = qx; if (qx == 1) ux[nx++] = -1; if (qx == W - 2) ux[nx++] = W;
    float gx = 0.f, gy = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b)
        for (int dyc = -1; dyc <= 1; ++dyc)
          for (int dxc = -1; dxc <= 1; ++dxc) {
            const int cy = uy[a] + dyc, cx = ux[b] + dxc;     // window centre
            if (cy < 0 || cy >= H || cx < 0 || cx >= W) continue;
            const long long k = (long long)cy * W + cx;
            const float dmx = c[k], dmy = c[total + k], dsx = c[2 * total + k], dsy = c[3 * total + k], dsxy = c[4 * total + k];
            gx += (dmx + 2.f * xv * dsx + yv * dsxy) / 9.f;
            gy += (dmy + 2.f * yv * dsy + xv * dsxy) / 9.f;
          }
    if (dx != nullptr) dx[i] = gx;
    if (dy != nullptr) dy[i] = gy;
  }
}

// --------------------------------------------------------------------------------- edge-aware smoothness (layers.py:199-212)
// loss = mean_x |d_x disp| exp(-mean_c |d_x img|) + mean_y (...)   disp [B][1][H][W], img [B][C][H][W]
__global__ void __launch_bounds__(256) edge_smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img, int B, int C,
                                                              int H, int W, float* __restrict__ partial) {
  const long long plane = (long long)H * W, total = (long long)B * plane;
  float acc[2] = {0.f, 0.f};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / plane);
    const int p = (int)(i - (long long)b * plane);
    const int y = p / W, x = p % W;
    const float* I = img + (long long)b * C * plane + p;
    if (x < W - 1) {
      float gi = 0.f;
      for (int c = 0; c < C; ++c) gi += fabsf(I[c * plane] - I[c * plane + 1]);
      acc[0] += fabsf(disp[i] - disp[i + 1]) * expf(-(gi / (float)C));
    }
    if (y < H - 1) {
      float gi = 0.f;
      for (int c = 0; c < C; ++c) gi += fabsf(I[c * plane] - I[c * plane + W]);
      acc[1] += fabsf(disp[i] - disp[i + W]) * expf(-(gi / (float)C));
    }
  }
  __shared__ float lds[8];
  block_sum256<2>(acc, lds, partial + blockIdx.x * 2);
}

__global__ void __launch_bounds__(256) edge_smooth_finalize_kernel(const float* __restrict__ partial, int blocks, int B, int H, int W,
                                                                   float* __restrict__ loss) {
  __shared__ double lds4[4];
  const double sx = block_sum_f64<2>(partial, blocks, 0, lds4);
  const double sy = block_sum_f64<2>(partial, blocks, 1, lds4);
  if (threadIdx.x != 0) return;
  loss[0] = (float)(sx / ((double)B * H * (W - 1))) + (float)(sy / ((double)B * (H - 1) * W));
}

__global__ void edge_smooth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ img, const float* __restrict__ dloss, int B,
                                       int C, int H, int W, float* __restrict__ ddisp) {
  const long long plane = (long long)H * W, total = (long long)B * plane;
  const float nx = (float)B * H * (W - 1), ny = (float)B * (H - 1) * W;
  const float up = dloss[0];
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / plane);
    const int p = (int)(i - (long long)b * plane);
    const int y = p / W, x = p % W;
    const float* I = img + (long long)b * C * plane + p;
    auto wgt = [&](long long off0, long long off1) {
      float gi = 0.f;
      for (int c = 0; c < C; ++c) gi += fabsf(I[c * plane + off0] - I[c * plane + off1]);
      return expf(-(gi / (float)C));
    };
    auto sgnf = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    float g = 0.f;
    if (x < W - 1) g += sgnf(disp[i] - disp[i + 1]) * wgt(0, 1) / nx;
    if (x > 0) g -= sgnf(disp[i - 1] - disp[i]) * wgt(-1, 0) / nx;
    if (y < H - 1) g += sgnf(disp[i] - disp[i + W]) * wgt(0, W) / ny;
    if (y > 0) g -= sgnf(disp[i - W] - disp[i]) * wgt(-W, 0) / ny;
    ddisp[i] = g * up;
  }
}

static inline int ew_blocks(long long total, int cap = 4096) {
  long long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

static inline int warp_blocks(int h, int w) { return ew_blocks((long long)h * w, 64); }

static int fill_warp_args(WarpArgs* a, const float* img, const float* depth, const float* proj, const float* kinv, int B, int h, int w,
                          int padding, int align) {
  DN_REQUIRE(img && depth && proj && kinv && B > 0 && h > 1 && w > 1, DN_ERR_BAD_ARG, "warp: bad argument (needs h,w > 1)");
  DN_REQUIRE((padding == 0 || padding == 1) && (align == 0 || align == 1), DN_ERR_BAD_ARG, "warp: padding %d / align %d", padding, align);
  a->img = img; a->depth = depth; a->proj = proj; a->kinv = kinv;
  a->B = B; a->h = h; a->w = w; a->padding = padding; a->align = align;
  a->tgt = nullptr; a->mask = nullptr; a->mask_sb = 0;
  return DN_OK;
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_pose_proj_fwd(const float* pose, int64_t pose_stride_b, const float* K, const float* Kinv, int32_t B, int32_t rotation_mode,
                     float downscale, float* proj, float* kinv_scaled, dn_stream_t stream) {
  DN_REQUIRE(pose && K && Kinv && proj && kinv_scaled && B > 0 && (rotation_mode == 0 || rotation_mode == 1) && downscale > 0.f,
             DN_ERR_BAD_ARG, "dn_pose_proj_fwd: bad argument");
  DN_LAUNCH(pose_proj_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), pose, (long long)pose_stride_b, K, Kinv, B,
                     rotation_mode, downscale, proj, kinv_scaled);
  return check_launch("pose_proj_fwd_kernel");
}

int32_t dn_warp_blocks(int32_t h, int32_t w) { return warp_blocks(h, w); }

int dn_pose_proj_bwd(const float* pose, int64_t pose_stride_b, const float* K, int32_t B, int32_t rotation_mode, float downscale,
                     const float* dproj_partial, int32_t nblk, float* dpose, int64_t dpose_stride_b, int32_t accumulate,
                     dn_stream_t stream) {
  DN_REQUIRE(pose && K && dproj_partial && dpose && B > 0 && nblk > 0, DN_ERR_BAD_ARG, "dn_pose_proj_bwd: bad argument");
  DN_LAUNCH(pose_proj_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), pose, (long long)pose_stride_b, K, B,
                     rotation_mode, downscale, dproj_partial, nblk, dpose, (long long)dpose_stride_b, accumulate);
  return check_launch("pose_proj_bwd_kernel");
}

int dn_inverse_warp_fwd(const float* img, const float* depth, const float* proj, const float* kinv, int32_t B, int32_t h, int32_t w,
                        int32_t padding_mode, int32_t align_corners, float* warped, dn_stream_t stream) {
  WarpArgs a;
  int rc = fill_warp_args(&a, img, depth, proj, kinv, B, h, w, padding_mode, align_corners);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(warped != nullptr, DN_ERR_BAD_ARG, "dn_inverse_warp_fwd: null output");
  DN_LAUNCH(warp_fwd_kernel, dim3(warp_blocks(h, w), B), dim3(256), 0, as_stream(stream), a, warped, (float*)nullptr);
  return check_launch("warp_fwd_kernel");
}

int dn_inverse_warp_bwd(const float* img, const float* depth, const float* proj, const float* kinv, int32_t B, int32_t h, int32_t w,
                        int32_t padding_mode, int32_t align_corners, const float* dwarped, float* ddepth, int32_t accumulate_depth,
                        float* dproj_partial, dn_stream_t stream) {
  WarpArgs a;
  int rc = fill_warp_args(&a, img, depth, proj, kinv, B, h, w, padding_mode, align_corners);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(dwarped && ddepth && dproj_partial, DN_ERR_BAD_ARG, "dn_inverse_warp_bwd: null pointer");
  DN_LAUNCH(warp_bwd_kernel, dim3(warp_blocks(h, w), B), dim3(256), 0, as_stream(stream), a, dwarped, (const float*)nullptr, 1.f,
                     ddepth, accumulate_depth, dproj_partial, (float*)nullptr, 0ll);
  return check_launch("warp_bwd_kernel");
}

int dn_photometric_fwd(const float* tgt, const float* ref, const float* depth, const float* proj, const float* kinv, const float* mask,
                       int64_t mask_stride_b, int32_t B, int32_t h, int32_t w, int32_t padding_mode, int32_t align_corners, float weight,
                       int32_t accumulate, float* partial, float* loss, dn_stream_t stream) {
  WarpArgs a;
  int rc = fill_warp_args(&a, ref, depth, proj, kinv, B, h, w, padding_mode, align_corners);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(tgt && partial && loss, DN_ERR_BAD_ARG, "dn_photometric_fwd: null pointer");
  a.tgt = tgt; a.mask = mask; a.mask_sb = mask_stride_b;
  hipStream_t s = as_stream(stream);
  const int nb = warp_blocks(h, w);
  DN_LAUNCH(warp_fwd_kernel, dim3(nb, B), dim3(256), 0, s, a, (float*)nullptr, partial);
  DN_LAUNCH(sum_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nb * B, (double)B * 3 * h * w, weight, accumulate, loss);
  return check_launch("photometric_fwd");
}

int dn_photometric_bwd(const float* tgt, const float* ref, const float* depth, const float* proj, const float* kinv, const float* mask,
                       int64_t mask_stride_b, int32_t B, int32_t h, int32_t w, int32_t padding_mode, int32_t align_corners, float weight,
                       const float* dloss, float* ddepth, int32_t accumulate_depth, float* dproj_partial, float* dmask,
                       int64_t dmask_stride_b, dn_stream_t stream) {
  WarpArgs a;
  int rc = fill_warp_args(&a, ref, depth, proj, kinv, B, h, w, padding_mode, align_corners);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(tgt && dloss && ddepth && dproj_partial, DN_ERR_BAD_ARG, "dn_photometric_bwd: null pointer");
  a.tgt = tgt; a.mask = mask; a.mask_sb = mask_stride_b;
  const float scale = weight / ((float)B * 3.f * (float)h * (float)w);
  DN_LAUNCH(warp_bwd_kernel, dim3(warp_blocks(h, w), B), dim3(256), 0, as_stream(stream), a, (const float*)nullptr, dloss, scale,
                     ddepth, accumulate_depth, dproj_partial, dmask, (long long)dmask_stride_b);
  return check_launch("warp_bwd_kernel(photometric)");
}

int dn_area_down(const float* in, int64_t planes, int32_t H, int32_t W, int32_t factor, float* out, dn_stream_t stream) {
  DN_REQUIRE(in && out && planes > 0 && factor >= 1 && H % factor == 0 && W % factor == 0, DN_ERR_BAD_ARG,
             "dn_area_down: %dx%d is not a multiple of %d", H, W, factor);
  DN_LAUNCH(area_down_kernel, dim3(ew_blocks(planes * (H / factor) * (W / factor))), dim3(256), 0, as_stream(stream), in,
                     (long long)planes, H, W, factor, out);
  return check_launch("area_down_kernel");
}

int dn_ssim_fwd(const float* x, const float* y, int64_t planes, int32_t H, int32_t W, float* out, dn_stream_t stream) {
  DN_REQUIRE(x && y && out && planes > 0 && H >= 2 && W >= 2, DN_ERR_BAD_ARG, "dn_ssim_fwd: bad argument");
  DN_LAUNCH(ssim_fwd_kernel, dim3(ew_blocks(planes * H * W)), dim3(256), 0, as_stream(stream), x, y, (long long)planes, H, W, out);
  return check_launch("ssim_fwd_kernel");
}

int dn_ssim_bwd(const float* x, const float* y, const float* dout, int64_t planes, int32_t H, int32_t W, float* workspace /* 5*planes*H*W */,
                float* dx, float* dy, dn_stream_t stream) {
  DN_REQUIRE(x && y && dout && workspace && planes > 0 && H >= 2 && W >= 2, DN_ERR_BAD_ARG, "dn_ssim_bwd: bad argument");
  hipStream_t s = as_stream(stream);
  const int nb = ew_blocks(planes * H * W);
  DN_LAUNCH(ssim_bwd_coef_kernel, dim3(nb), dim3(256), 0, s, x, y, dout, (long long)planes, H, W, workspace);
  DN_LAUNCH(ssim_bwd_gather_kernel, dim3(nb), dim3(256), 0, s, x, y, workspace, (long long)planes, H, W, dx, dy);
  return check_launch("ssim_bwd");
}

int dn_edge_smooth_fwd(const float* disp, const float* img, int32_t B, int32_t C, int32_t H, int32_t W, float* partial, float* loss,
                       dn_stream_t stream) {
  DN_REQUIRE(disp && img && partial && loss && B > 0 && C > 0 && H >= 2 && W >= 2, DN_ERR_BAD_ARG, "dn_edge_smooth_fwd: bad argument");
  hipStream_t s = as_stream(stream);
  const int nb = ew_blocks((long long)B * H * W, 1024);
  DN_LAUNCH(edge_smooth_fwd_kernel, dim3(nb), dim3(256), 0, s, disp, img, B, C, H, W, partial);
  DN_LAUNCH(edge_smooth_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nb, B, H, W, loss);
  return check_launch("edge_smooth_fwd");
}

int dn_edge_smooth_bwd(const float* disp, const float* img, const float* dloss, int32_t B, int32_t C, int32_t H, int32_t W, float* ddisp,
                       dn_stream_t stream) {
  DN_REQUIRE(disp && img && dloss && ddisp && B > 0 && C > 0 && H >= 2 && W >= 2, DN_ERR_BAD_ARG, "dn_edge_smooth_bwd: bad argument");
  DN_LAUNCH(edge_smooth_bwd_kernel, dim3(ew_blocks((long long)B * H * W)), dim3(256), 0, as_stream(stream), disp, img, dloss, B, C,
                     H, W, ddisp);
  return check_launch("edge_smooth_bwd_kernel");
}

int32_t dn_edge_smooth_blocks(int32_t B, int32_t H, int32_t W) { return ew_blocks((long long)B * H * W, 1024); }

}  // extern "C"
