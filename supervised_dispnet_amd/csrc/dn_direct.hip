// Direct (matrix-core-free) kernels for the one-channel disparity heads -- predict_disp, models/Disp_vgg_BN.py:66-70:
// conv3x3(C -> 1) + alpha*sigmoid + beta -- forward, input gradient and weight gradient.  2*9*C flops per pixel against
// 4*C bytes of input: arithmetic intensity ~4.4 flop/B, i.e. HBM-bound; on the implicit-GEMM path the single
// output column is padded to a 32-wide MFMA tile (97 % of the matrix work wasted, 0.3-0.7 ms per launch at 128x416 b32).
// Here: one thread per (pixel, 4-channel group), NHWC float4 loads, the 9*C weights in LDS, wave shuffles for the reductions.
// Dispatch happens inside dn_conv2d_fwd / dn_conv2d_dgrad / dn_conv2d_wgrad (same ABI, same packed-weight layout).
#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float head_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

// ---------------------------------------------------------------------------------------------------------- forward
// y[pix] = act(bias + sum_{tap, c} x[pix*s + tap][c] * w[tap][c]);  packed weights row 0 is exactly w[tap][c] (k = tap*C + c).
// Lanes: the C/4 channel groups of a pixel are adjacent lanes (LG = log2(C/4)); the partial dot products are combined with
// xor-shuffles inside the group.  Block = 256 threads = 256 >> LG pixels.
__global__ void __launch_bounds__(256) head_fwd_kernel(const IgemmParams p, int LG) {
  extern __shared__ float wsm[];                     // [ntaps][C]
  const KOperand& S = p.in[0];
  const int C = S.C, ntaps = p.ph[0].ntaps;
  for (int i = threadIdx.x; i < ntaps * C; i += 256) wsm[i] = p.w[i];
  __syncthreads();
  const int cg = threadIdx.x & ((1 << LG) - 1);
  // grid-stride over the pixel groups: the weight preload + barrier above is paid once per block, not once per 64 pixels
  for (long long base = (long long)blockIdx.x * (256 >> LG); base < p.M; base += (long long)gridDim.x * (256 >> LG)) {
    const long long pix = base + (threadIdx.x >> LG);
    const bool live = pix < p.M;
    unsigned gx, gy;
    const unsigned t = fastdiv_dev(live ? (unsigned)pix : 0u, (unsigned)p.GW, p.mGW, &gx);
    const int n = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
    const int by = (int)gy * p.sy, bx = (int)gx * p.sx;
    float acc = 0.f;
    for (int j = 0; j < ntaps; ++j) {
      const int iy = by + p.tdy[j], ix = bx + p.tdx[j];
      if (live && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(S.p + (long long)n * S.sn + (long long)iy * S.sh + (long long)ix * S.sw + cg * 4);
        const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + j * C + cg * 4);
        acc += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
      }
    }
    for (int d = 1; d < (1 << LG); d <<= 1) acc += __shfl_xor(acc, d);
    if (live && cg == 0) {
      const KResult& R = p.out[0];
      float v = head_act(acc + (p.bias ? p.bias[0] : 0.f), p.act, p.act_p0, p.act_p1);
      float* o = R.p + (long long)n * R.sn + (long long)gy * R.sh + (long long)gx * R.sw;
      if (R.accumulate) v += *o;
      *o = v;
    }
  }
}

// --------------------------------------------------------------------------------------------------- input gradient
// dx[q][c] (+)= sum_tap dy[q + tap'] * w[c][tap];  the dgrad plan's packed weights are [c][Kp] with k = tap index and the
// tap tables already hold the flipped offsets.  One thread per (pixel, 4-channel group): 9 scalar dy reads, one 16-byte store.
__global__ void __launch_bounds__(256) head_dgrad_kernel(const IgemmParams p, int LG, int Kp) {
  extern __shared__ float wsm[];                     // [ntaps][Ntot]   (transposed on the way in)
  const int C = p.Ntot, ntaps = p.ph[0].ntaps;
  for (int i = threadIdx.x; i < ntaps * C; i += 256) {
    const int c = i / ntaps, j = i - c * ntaps;
    wsm[j * C + c] = p.w[(long long)c * Kp + j];
  }
  __syncthreads();
  const KOperand& G = p.in[0];
  const int cg = threadIdx.x & ((1 << LG) - 1);
  for (long long pix = (long long)blockIdx.x * (256 >> LG) + (threadIdx.x >> LG); pix < p.M; pix += (long long)gridDim.x * (256 >> LG)) {
    unsigned gx, gy;
    const unsigned t = fastdiv_dev((unsigned)pix, (unsigned)p.GW, p.mGW, &gx);
    const int n = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < ntaps; ++j) {
      const int iy = (int)gy + p.tdy[j], ix = (int)gx + p.tdx[j];
      if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
        const float g = G.p[(long long)n * G.sn + (long long)iy * G.sh + (long long)ix * G.sw];
        const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + j * C + cg * 4);
        acc += g * w;
      }
    }
    const KResult& R = p.out[0];
    f32x4* o = reinterpret_cast<f32x4*>(R.p + (long long)n * R.sn + (long long)gy * R.sh + (long long)gx * R.sw + cg * 4);
    if (R.accumulate) acc += *o;
    *o = acc;
  }
}

// -------------------------------------------------------------------------------------------------- weight gradient
// dw[c][tap] = sum_pix dy[pix - tap_offset] * x[pix][c]  (x read ONCE, the 9 dy neighbours come from L1).
// Thread = (pixel lane, 4-channel group) with 9 float4 accumulators; pixels strided over the grid; lanes of one channel group
// are folded with xor-shuffles, waves through LDS; each block writes one partial [ntaps][C] slab, head_wgrad_reduce_kernel
// sums the slabs in a fixed order (deterministic) into the framework layout [1][C][R][S].
constexpr int kHeadMaxTaps = 9;

__global__ void __launch_bounds__(256) head_wgrad_kernel(const IgemmParams p, int LG, float* __restrict__ slabs) {
  __shared__ float red[4][kHeadMaxTaps * 4 * 64];
  const KOperand& S = p.in[0];
  const int C = S.C, ntaps = p.ph[0].ntaps;
  const int cg = threadIdx.x & ((1 << LG) - 1);
  const int lanes_per_px = 1 << LG, px_per_block = 256 >> LG;
  f32x4 acc[kHeadMaxTaps];
#pragma unroll
  for (int j = 0; j < kHeadMaxTaps; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // x pixel q = (n, iy, ix) pairs with dy at the output position (oy, ox) for which oy*s + tdy = iy, i.e. stride-1 heads: oy = iy - tdy
  const long long npix = (long long)p.N * p.IH * p.IW;
  for (long long q = (long long)blockIdx.x * px_per_block + (threadIdx.x >> LG); q < npix; q += (long long)gridDim.x * px_per_block) {
    const int ix = (int)(q % p.IW);
    const long long tq = q / p.IW;
    const int iy = (int)(tq % p.IH), n = (int)(tq / p.IH);
    const f32x4 x = *reinterpret_cast<const f32x4*>(S.p + (long long)n * S.sn + (long long)iy * S.sh + (long long)ix * S.sw + cg * 4);
#pragma unroll
    for (int j = 0; j < kHeadMaxTaps; ++j) {
      if (j < ntaps) {
        const int oy = iy - p.tdy[j], ox = ix - p.tdx[j];
        if ((unsigned)oy < (unsigned)p.GH && (unsigned)ox < (unsigned)p.GW) acc[j] += p.g[((long long)n * p.GH + oy) * p.GW + ox] * x;
      }
    }
  }
  // fold the pixel lanes of this wave (lanes that share cg differ in bits >= LG)
#pragma unroll
  for (int j = 0; j < kHeadMaxTaps; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[j][e];
      for (int d = lanes_per_px; d < 64; d <<= 1) v += __shfl_xor(v, d);
      acc[j][e] = v;
    }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < lanes_per_px) {
#pragma unroll
    for (int j = 0; j < kHeadMaxTaps; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][(j * 4 + e) * 64 + lane] = acc[j][e];
  }
  __syncthreads();
  float* slab = slabs + (long long)blockIdx.x * ntaps * C;
  for (int i = threadIdx.x; i < ntaps * C; i += 256) {
    const int j = i / C, c = i - j * C;
    const int idx = (j * 4 + (c & 3)) * 64 + (c >> 2);
    slab[i] = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
  }
}

// one block per weight element: 256 threads fold the slabs in a fixed tree (deterministic), no serial 1024-long chain
// (cgs > 1: the slab of block b holds only the sixteen channels of group b % cgs -- head_wgrad2_kernel)
__global__ void __launch_bounds__(256) head_wgrad_reduce_kernel(const IgemmParams p, const float* __restrict__ slabs, int nslabs, int cgs,
                                                                float* __restrict__ dw) {
  __shared__ float part[4];
  const int C = p.in[0].C, ntaps = p.ph[0].ntaps;
  const int i = blockIdx.x;
  float s = 0.f;
  const int b0 = cgs > 1 ? ((i % C) >> 4) : 0;
  for (int b = b0 + threadIdx.x * cgs; b < nslabs; b += 256 * cgs) s += slabs[(long long)b * ntaps * C + i];
  for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int j = i / C, c = i - j * C;
    dw[((long long)c * p.R + p.tr[j]) * p.S + p.ts[j]] = (part[0] + part[1]) + (part[2] + part[3]);      // [1][C][R][S]
  }
}

// =====================================================================================================================
// Second generation of the three head kernels (r02) for the dense 3x3 / stride 1 / pad 1 heads with C % 16 == 0 -- all four
// predict_disp layers of the DispNets.  What bounded the first generation was not HBM but the number of vector-memory
// instructions (a CU retires about one per 40-60 cycles whatever its width): 9 neighbour fetches per (pixel, 4-channel group).
//   forward   separable form  y[p] = sum_tap t_tap[p + off_tap],  t_tap[q] = <x[q], w[tap]>: a block reads an (8+2) x (64+2)
//             input tile ONCE (64 contiguous bytes per thread), leaves the nine partial dot products in LDS and sums nine LDS
//             words per output pixel.
//   dgrad     one thread per pixel: 9 neighbour values of the 1-channel gradient, 16 channels x 9 taps of FMAs from LDS-broadcast
//             weights, whole-pixel stores.
//   wgrad     dW[tap][c] = sum_q dy[q - off_tap] x[q][c] as 16x16x4 MFMAs (rows = taps, columns = 16 channels, contraction over
//             pixels): a wave stages 64 pixels of x (float4 loads) and the three gradient rows it needs in LDS and feeds both
//             operands from there -- 5 global + ~36 LDS instructions per 64 pixels instead of 40 global ones.
// =====================================================================================================================
// Forward tile TY x TX (8 x 64 on the large maps; 4 x 16 where that would leave fewer than ~4 blocks per CU: the 16 x 52 head ran 64
// blocks of serial 128-channel dot products, 28 us for 60 MFLOP).  An item is (input pixel, 16-channel group), groups on adjacent lanes:
// 64 contiguous bytes per lane, a pixel's channels contiguous across its lanes, the partial dot products folded by xor-shuffles.
template <int TY, int TX>
__global__ void __launch_bounds__(256) head_fwd2_kernel(const IgemmParams p, int tilesX, int tilesY, int LGc) {
  constexpr int LD = TX + 4;
  extern __shared__ float sm[];
  const KOperand& S = p.in[0];
  const int C = S.C;
  float* wsm = sm;                                   // [9][C]
  float* t = sm + 9 * C;                             // [9][TY + 2][LD]
  for (int i = threadIdx.x; i < 9 * C; i += 256) wsm[i] = p.w[i];
  int b = blockIdx.x;
  const int tx = b % tilesX;
  b /= tilesX;
  const int ty = b % tilesY, n = b / tilesY;
  const int y0 = ty * TY, x0 = tx * TX;
  __syncthreads();
  constexpr int IN_W = TX + 2, IN_N = (TY + 2) * IN_W;
  const int G = 1 << LGc;                            // 16-channel groups: C / 16
  for (int base = 0; base < (IN_N << LGc); base += 256) {
    const int it = base + (int)threadIdx.x;
    const int grp = it & (G - 1), idx = it >> LGc;
    const int r = idx / IN_W, ci = idx - r * IN_W;
    const int iy = y0 - 1 + r, ix = x0 - 1 + ci;
    float tj[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) tj[j] = 0.f;
    if (idx < IN_N && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
      const float* xp = S.p + (long long)n * S.sn + (long long)iy * S.sh + (long long)ix * S.sw + 16 * grp;
      f32x4 xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const f32x4*>(xp + 4 * q);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + j * C + 16 * grp + 4 * q);
          a += xv[q][0] * w[0] + xv[q][1] * w[1] + xv[q][2] * w[2] + xv[q][3] * w[3];
        }
        tj[j] = a;
      }
    }
    for (int d = 1; d < G; d <<= 1) {                 // (fixed order: deterministic)
#pragma unroll
      for (int j = 0; j < 9; ++j) tj[j] += __shfl_xor(tj[j], d);
    }
    if (idx < IN_N && grp == 0) {
#pragma unroll
      for (int j = 0; j < 9; ++j) t[(j * (TY + 2) + r) * LD + ci] = tj[j];
    }
  }
  __syncthreads();
  const KResult& R = p.out[0];
  const float bias = p.bias ? p.bias[0] : 0.f;
  for (int o = threadIdx.x; o < TY * TX; o += 256) {
    const int oy = o / TX, ox = o - oy * TX;
    const int gy = y0 + oy, gx = x0 + ox;
    if (gy < p.GH && gx < p.GW) {
      float acc = bias;
#pragma unroll
      for (int j = 0; j < 9; ++j) acc += t[(j * (TY + 2) + oy + 1 + p.tdy[j]) * LD + ox + 1 + p.tdx[j]];
      float v = head_act(acc, p.act, p.act_p0, p.act_p1);
      float* op = R.p + (long long)n * R.sn + (long long)gy * R.sh + (long long)gx * R.sw;
      if (R.accumulate) v += *op;
      *op = v;
      // depth = 1 / disp (train.py:445) leaves with the disparity: the caller's reciprocal() then launches nothing
      if (p.recip_out != nullptr) p.recip_out[((long long)n * p.GH + gy) * p.GW + gx] = 1.f / v;
    }
  }
}

// one thread per (pixel, 16-channel group): 9 neighbour values of the 1-channel gradient, 16 channels x 9 taps from the LDS weights,
// 64 contiguous bytes stored per lane (a pixel's groups on adjacent lanes)
__global__ void __launch_bounds__(256) head_dgrad2_kernel(const IgemmParams p, int Kp, int LGc) {
  extern __shared__ float wsm[];                     // [9][C]   (transposed on the way in)
  const int C = p.Ntot;
  for (int i = threadIdx.x; i < 9 * C; i += 256) {
    const int c = i / 9, j = i - c * 9;
    wsm[j * C + c] = p.w[(long long)c * Kp + j];
  }
  __syncthreads();
  const KOperand& G = p.in[0];
  const KResult& R = p.out[0];
  const long long items = (long long)p.M << LGc;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
    const int grp = (int)(it & ((1 << LGc) - 1));
    const long long pix = it >> LGc;
    unsigned gx, gy;
    const unsigned tq = fastdiv_dev((unsigned)pix, (unsigned)p.GW, p.mGW, &gx);
    const int n = (int)fastdiv_dev(tq, (unsigned)p.GH, p.mGH, &gy);
    float g[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int iy = (int)gy + p.tdy[j], ix = (int)gx + p.tdx[j];
      const bool ok = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      g[j] = ok ? G.p[(long long)n * G.sn + (long long)iy * G.sh + (long long)ix * G.sw] : 0.f;
    }
    float* op = R.p + (long long)n * R.sn + (long long)gy * R.sh + (long long)gx * R.sw + 16 * grp;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = R.accumulate ? *reinterpret_cast<const f32x4*>(op + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += g[j] * *reinterpret_cast<const f32x4*>(wsm + j * C + 16 * grp + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(op + 4 * q) = acc[q];
  }
}

constexpr int kH2MaxCG = 8;                           // C <= 128
constexpr int kH2DLD = 68;                            // LDS row of the staged gradient rows (66 used)

// A block takes ONE 16-channel group (blockIdx % cgs) and every (gridDim / cgs)-th set of four row segments: C / 16 times the blocks of
// a kernel that walks all groups per segment (the 16 x 52 head: 128 -> 1024 blocks); its slab holds that group's columns only.
__global__ void __launch_bounds__(256) head_wgrad2_kernel(const IgemmParams p, int cgs, float* __restrict__ slabs) {
  __shared__ float xs[4][64 * 16];                    // per wave: 64 pixels x 16 channels
  __shared__ float ds[4][4 * kH2DLD];                 // per wave: gradient rows y+1, y, y-1 (columns x0-1 .. x0+64) and a zero row
  __shared__ float red[4][9][16];
  const KOperand& S = p.in[0];
  const int C = S.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, j = lane & 15;
  const int q = (int)blockIdx.x % cgs, bq = (int)blockIdx.x / cgs, nbq = (int)gridDim.x / cgs;
  float* xw = xs[wave];
  float* dw_ = ds[wave];
  for (int i = lane; i < kH2DLD; i += 64) dw_[3 * kH2DLD + i] = 0.f;
  // A operand (rows = taps): lane (tap = j, pixel sub-index k) reads dy[y - tdy][x - tdx] = staged row (1 - tdy... see below), col 1 + x - tdx
  const bool tap_ok = j < 9;
  const int tdy = tap_ok ? p.tdy[j] : 0, tdx = tap_ok ? p.tdx[j] : 0;
  // staged rows: index 0 <-> gradient row y-1, 1 <-> y, 2 <-> y+1; dy row needed = y - tdy -> index 1 - tdy; invalid taps -> zero row 3
  const int a_off = (tap_ok ? (1 - tdy) : 3) * kH2DLD + 1 + k - (tap_ok ? tdx : 0);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int segX = (p.IW + 63) / 64;
  const long long nseg = (long long)p.N * p.IH * segX;
  for (long long sg = (long long)bq * 4 + wave; sg < nseg; sg += (long long)nbq * 4) {
    const int sx = (int)(sg % segX);
    const long long ty = sg / segX;
    const int y = (int)(ty % p.IH), n = (int)(ty / p.IH);
    const int x0 = sx * 64;
    // ---- the three gradient rows (1-channel map [N][GH][GW]), columns x0-1 .. x0+64
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int gy = y - 1 + r;
      const bool rok = (unsigned)gy < (unsigned)p.GH;
      const float* row = p.g + ((long long)n * p.GH + (rok ? gy : 0)) * p.GW;
      const int gx = x0 - 1 + lane;
      dw_[r * kH2DLD + lane] = (rok && (unsigned)gx < (unsigned)p.GW) ? row[gx] : 0.f;
      if (lane < 2) {
        const int gx2 = x0 + 63 + lane;
        dw_[r * kH2DLD + 64 + lane] = (rok && (unsigned)gx2 < (unsigned)p.GW) ? row[gx2] : 0.f;
      }
    }
    const float* xrow = S.p + (long long)n * S.sn + (long long)y * S.sh;
    // ---- 64 pixels x 16 channels of x into LDS: lane = (pixel 16 i + lane / 4, channel quad lane % 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = 16 * i + (lane >> 2);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (x0 + px < p.IW) v = *reinterpret_cast<const f32x4*>(xrow + (long long)(x0 + px) * S.sw + q * 16 + 4 * (lane & 3));
      *reinterpret_cast<f32x4*>(xw + px * 16 + 4 * (lane & 3)) = v;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = dw_[a_off + 4 * s];
      const float b = xw[(4 * s + k) * 16 + j];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
  }
  // ---- fold the four waves (fixed order), one slab [9][C] per block (columns 16 q .. 16 q + 15)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tap = 4 * k + r;
    if (tap < 9) red[wave][tap][j] = acc[r];
  }
  __syncthreads();
  float* slab = slabs + (long long)blockIdx.x * 9 * C;
  for (int i = threadIdx.x; i < 9 * 16; i += 256) {
    const int tap = i >> 4, c = i & 15;
    slab[tap * C + 16 * q + c] = (red[0][tap][c] + red[1][tap][c]) + (red[2][tap][c] + red[3][tap][c]);
  }
}

// ------------------------------------------------------------------------------------------------------- dispatch
static int log2_exact(int v) {
  for (int l = 0; l <= 6; ++l)
    if ((1 << l) == v) return l;
  return -1;
}

static bool plain_vec_operand(const KOperand& o) { return o.vec && o.up == 0 && o.scale == nullptr && o.C % 4 == 0 && log2_exact(o.C / 4) >= 0; }

constexpr int kHeadSlabs = 1024;

bool head_fwd_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  return d->kind == DN_CONV_FWD && p.Ntot == 1 && p.n_in == 1 && p.n_out == 1 && p.nphases == 1 && !p.reflect && p.bn_partial == nullptr &&
         plain_vec_operand(p.in[0]) && p.ph[0].ntaps * p.in[0].C * sizeof(float) <= 48 * 1024;
}

// the r02 kernels: 3x3 taps within +-1, stride 1, 16-aligned channels
static bool head2_geometry(const IgemmParams& p, int C) {
  if (false || p.ph[0].ntaps != 9 || C % 16 != 0 || C > 16 * kH2MaxCG || p.sy != 1 || p.sx != 1) return false;
  for (int j = 0; j < 9; ++j)
    if (p.tdy[j] < -1 || p.tdy[j] > 1 || p.tdx[j] < -1 || p.tdx[j] > 1) return false;
  return true;
}

int launch_head_fwd(const IgemmParams& p, hipStream_t stream) {
  if (head2_geometry(p, p.in[0].C) && p.GH == p.IH && p.GW == p.IW) {
    const int LGc = log2_exact(p.in[0].C / 16);
    const int big = p.N * ((p.GW + 63) / 64) * ((p.GH + 7) / 8);
    if (big >= 1024) {
      const int tilesX = (p.GW + 63) / 64, tilesY = (p.GH + 7) / 8;
      const size_t lds = (size_t)(9 * p.in[0].C + 9 * 10 * 68) * sizeof(float);
      DN_LAUNCH((head_fwd2_kernel<8, 64>), dim3(p.N * tilesX * tilesY), dim3(256), lds, stream, p, tilesX, tilesY, LGc);
    } else {
      const int tilesX = (p.GW + 15) / 16, tilesY = (p.GH + 3) / 4;
      const size_t lds = (size_t)(9 * p.in[0].C + 9 * 6 * 20) * sizeof(float);
      DN_LAUNCH((head_fwd2_kernel<4, 16>), dim3(p.N * tilesX * tilesY), dim3(256), lds, stream, p, tilesX, tilesY, LGc);
    }
    set_last_kernel("dn::head_fwd2_kernel");
    return check_launch("head_fwd2_kernel");
  }
  const int LG = log2_exact(p.in[0].C / 4);
  const int px = 256 >> LG;
  int hblocks = (p.M + px - 1) / px;
  if (hblocks > 2048) hblocks = 2048;          // 8 blocks per CU, grid-stride
  DN_LAUNCH(head_fwd_kernel, dim3(hblocks), dim3(256), p.ph[0].ntaps * p.in[0].C * sizeof(float), stream, p, LG);
  set_last_kernel("dn::head_fwd_kernel");
  return check_launch("head_fwd_kernel");
}

bool head_fwd_fuses_reciprocal(const dn_conv_desc* d, const IgemmParams& p) {
  return !knobs().no_direct && head_fwd_eligible(d, p) && head2_geometry(p, p.in[0].C) && p.GH == p.IH && p.GW == p.IW && !p.out[0].accumulate;
}

bool head_dgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  return d->kind == DN_CONV_DGRAD && p.n_in == 1 && p.in[0].C == 1 && p.in[0].up == 0 && p.in[0].scale == nullptr && p.n_out == 1 &&
         p.nphases == 1 && !p.reflect && p.Ntot % 4 == 0 && log2_exact(p.Ntot / 4) >= 0 && p.act == DN_ACT_NONE && p.bias == nullptr &&
         (reinterpret_cast<uintptr_t>(p.out[0].p) & 15) == 0 && p.out[0].sw % 4 == 0 && p.out[0].sh % 4 == 0 && p.out[0].sn % 4 == 0 &&
         p.ph[0].ntaps * p.Ntot * sizeof(float) <= 48 * 1024 && p.osy == 1;
}

int launch_head_dgrad(const IgemmParams& p, hipStream_t stream) {
  if (head2_geometry(p, p.Ntot)) {
    const int LGc = log2_exact(p.Ntot / 16);
    long long hb = (((long long)p.M << LGc) + 255) / 256;
    if (hb > 4096) hb = 4096;
    DN_LAUNCH(head_dgrad2_kernel, dim3((int)hb), dim3(256), 9 * p.Ntot * sizeof(float), stream, p, p.ph[0].nchunks * kChunk, LGc);
    set_last_kernel("dn::head_dgrad2_kernel");
    return check_launch("head_dgrad2_kernel");
  }
  const int LG = log2_exact(p.Ntot / 4);
  const int px = 256 >> LG;
  int hblocks = (p.M + px - 1) / px;
  if (hblocks > 2048) hblocks = 2048;
  DN_LAUNCH(head_dgrad_kernel, dim3(hblocks), dim3(256), p.ph[0].ntaps * p.Ntot * sizeof(float), stream, p, LG,
                     p.ph[0].nchunks * kChunk);
  set_last_kernel("dn::head_dgrad_kernel");
  return check_launch("head_dgrad_kernel");
}

bool head_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p) {
  return fwd->kind == DN_CONV_FWD && p.Ntot == 1 && p.n_in == 1 && !p.reflect && plain_vec_operand(p.in[0]) && p.in[0].C <= 256 &&
         p.ph[0].ntaps <= kHeadMaxTaps && p.sy == 1 && p.sx == 1;
}

size_t head_wgrad_workspace_bytes(const IgemmParams& p) { return (size_t)kHeadSlabs * p.ph[0].ntaps * p.in[0].C * sizeof(float); }

int launch_head_wgrad(const IgemmParams& p, float* dw, float* workspace, hipStream_t stream) {
  const KOperand& S0 = p.in[0];
  if (head2_geometry(p, S0.C) && p.GH == p.IH && p.GW == p.IW && S0.sw == S0.C) {
    const long long nseg = (long long)p.N * p.IH * ((p.IW + 63) / 64);
    const int cgs = S0.C / 16;
    long long want = (nseg + 3) / 4 * cgs;
    if (want > kHeadSlabs) want = kHeadSlabs;
    const int blocks = (int)(want / cgs) * cgs;                       // a whole number of blocks per channel group
    if (blocks > 0) {
      DN_LAUNCH(head_wgrad2_kernel, dim3(blocks), dim3(256), 0, stream, p, cgs, workspace);
      set_last_kernel("dn::head_wgrad2_kernel");
      int rc = check_launch("head_wgrad2_kernel");
      if (rc != DN_OK) return rc;
      DN_LAUNCH(head_wgrad_reduce_kernel, dim3(9 * S0.C), dim3(256), 0, stream, p, workspace, blocks, cgs, dw);
      return check_launch("head_wgrad_reduce_kernel");
    }
  }
  const int LG = log2_exact(p.in[0].C / 4);
  const int px = 256 >> LG;
  const long long npix = (long long)p.N * p.IH * p.IW;
  int blocks = (int)((npix + px - 1) / px);
  if (blocks > kHeadSlabs) blocks = kHeadSlabs;
  DN_LAUNCH(head_wgrad_kernel, dim3(blocks), dim3(256), 0, stream, p, LG, workspace);
  set_last_kernel("dn::head_wgrad_kernel");
  int rc = check_launch("head_wgrad_kernel");
  if (rc != DN_OK) return rc;
  const int tot = p.ph[0].ntaps * p.in[0].C;
  DN_LAUNCH(head_wgrad_reduce_kernel, dim3(tot), dim3(256), 0, stream, p, workspace, blocks, 1, dw);
  return check_launch("head_wgrad_reduce_kernel");
}

}  // namespace dn
