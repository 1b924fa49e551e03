// 7x7 / stride-2 first layers on <= 3-channel NCHW images, LDS-resident, three-piece arithmetic (round 4):
//   * ResNet conv1 (3 -> 64 + BatchNorm statistics; torchvision resnet via reference models/Disp_res_50.py -- BASELINE configs[3]);
//   * PoseExpNet conv1 (target + two reference images = three 3-channel operands -> 16, ReLU; reference models/PoseExpNet.py:31,72-73).
// 49 taps x 3..9 channels are more than the scheduled kernels' tap tables take at once (they ran as tap windows / on the unscheduled
// kernel: 0.46-0.70 ms for 6-11 GFLOP).  Here, in the stem3 style (dn_lds3.hip):
//   * a block of 8 waves owns 16 x TW output pixels; its (37 x (2 TW + 8)) x planes input tile is fetched ONCE with aligned 16-byte loads
//     along x (the image rows are contiguous in NCHW) and kept in LDS as fp32 planes;
//   * K is laid out for the gather, not for the weights: K slot = (plane, tap row r) x 8 with slot e <-> tap column e - 1 (e = 0 is a
//     dead slot whose weight is zero), so the eight values of a lane's slot are EIGHT CONSECUTIVE floats of one plane row starting at an
//     even column: four ds_read_b64, split into the three bf16 pieces in registers; a K-step of 32 = four (plane, r) combinations;
//   * the weights are re-ordered to that K layout, split and kept in LDS in fragment order (one ds_read_b128 per piece and M tile);
//   * D = W . X with the output channels as M: float4 stores from the C/D layout; BatchNorm partial statistics as in stem3_conv_kernel
//     (per wave about a pivot, DPP row sums, two waves merged per 128-pixel row).
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sk_split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 x = f32x2{v[e], v[e + 1]};
    const bf16x2 h2 = __builtin_convertvector(x, bf16x2);
    const f32x2 r = x - __builtin_convertvector(h2, f32x2);
    const bf16x2 m2 = __builtin_convertvector(r, bf16x2);
    const f32x2 q = r - __builtin_convertvector(m2, f32x2);
    const bf16x2 l2 = __builtin_convertvector(q, bf16x2);
    h[e] = h2[0]; h[e + 1] = h2[1];
    m[e] = m2[0]; m[e + 1] = m2[1];
    l[e] = l2[0]; l[e + 1] = l2[1];
  }
}

__device__ __forceinline__ f32x4 sk_act4(f32x4 v, int act, float p0, float p1) {
  f32x4 r = v;
  switch (act) {
    case DN_ACT_RELU:
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = v[e] > 0.f ? v[e] : 0.f;
      break;
    case DN_ACT_LEAKY:
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = v[e] > 0.f ? v[e] : v[e] * p0;
      break;
    case DN_ACT_ELU:
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = v[e] > 0.f ? v[e] : (expf(v[e]) - 1.f);
      break;
    case DN_ACT_SIGMOID_AFFINE:
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = p0 / (1.f + expf(-v[e])) + p1;
      break;
    default: break;
  }
  return r;
}

template <int ROT>
__device__ __forceinline__ float sk_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROT, 0xf, 0xf, false));
}
__device__ __forceinline__ float sk_row16_sum(float v) {
  v += sk_row_ror<8>(v);
  v += sk_row_ror<4>(v);
  v += sk_row_ror<2>(v);
  v += sk_row_ror<1>(v);
  return v;
}

struct SkGeo {
  int tilesX, tilesY, ntiles, per_xcd;
  int plane_op[9], plane_c[9], kbase[3];       // input plane -> (operand, channel); packed-weight K offset of an operand
};

constexpr int SK_TH = 16, SK_ROWS = 2 * (SK_TH - 1) + 7;       // 37 input rows

// MT: 16-channel M tiles (Ntot = 16 MT); TW: tile width; NPL: input planes (all operands' channels)
template <int MT, int TW, int NPL>
struct SkCfg {
  static constexpr int COLS4 = (2 * TW + 8) / 4;               // float4 groups per plane row: columns 2 gx0 - 4 .. 2 gx0 + 2 TW + 3
  static constexpr int COLSP = 4 * COLS4;
  static constexpr int PLANE = SK_ROWS * COLSP;                // floats
  static constexpr int NCOMBO = 7 * NPL;
  static constexpr int NKS = (NCOMBO + 3) / 4;
  static constexpr size_t PLANES_B = (size_t)NPL * PLANE * 4;
  static constexpr size_t W_B = (size_t)MT * NKS * 3 * 64 * 16;
  static constexpr size_t STAT_B = (size_t)8 * MT * 16 * 2 * 4;
  static constexpr size_t LDS = PLANES_B + W_B + STAT_B;
  static constexpr int ITEMS = NPL * SK_ROWS * COLS4, ROUNDS = (ITEMS + 511) / 512;
  static constexpr int PT = 2 * TW / 16;                        // 16-pixel tiles per wave (two rows)
  static_assert(LDS <= 160 * 1024, "one block per CU");
  static_assert(PT % 2 == 0, "pixel tiles in pairs");
};

template <int MT, int TW, int NPL>
__global__ void __launch_bounds__(512, 1) stemk_conv_kernel(const IgemmParams p, const SkGeo geo) {
  using Cfg = SkCfg<MT, TW, NPL>;
  constexpr int NKS = Cfg::NKS, COLSP = Cfg::COLSP, PLANE = Cfg::PLANE;
  extern __shared__ __align__(16) char lds[];
  float* pl = reinterpret_cast<float*>(lds);
  bf16x8* wl = reinterpret_cast<bf16x8*>(lds + Cfg::PLANES_B);               // [m][ks][piece][lane]
  float* wstat = reinterpret_cast<float*>(lds + Cfg::PLANES_B + Cfg::W_B);   // [wave][16 MT][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KPhase ph = p.ph[0];
  const int Kp = ph.nchunks * kChunk;

  // ---- the weights in the kernel's K order, split, in fragment order: row = output channel 16 m + j, slot (combo 4 ks + g, e)
  for (int it = tid; it < MT * NKS * 64; it += 512) {
    const int l = it & 63, ks = (it >> 6) % NKS, m = it / (64 * NKS);
    const int jj = l & 15, gg = l >> 4;
    const int combo = 4 * ks + gg, n = 16 * m + jj;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (combo < Cfg::NCOMBO && n < p.Ntot) {
      const int plane = combo / 7, r = combo - 7 * plane;
      const int op = geo.plane_op[plane], c = geo.plane_c[plane];
      const float* wrow = p.w + ph.w_off + (long long)n * Kp + geo.kbase[op];
      const int Cop = p.in[op].C;
#pragma unroll
      for (int s = 0; s < 7; ++s) v[1 + s] = wrow[(7 * r + s) * Cop + c];
    }
    bf16x8 h, mm, lo;
    sk_split3(v, h, mm, lo);
    wl[((m * NKS + ks) * 3 + 0) * 64 + l] = h;
    wl[((m * NKS + ks) * 3 + 1) * 64 + l] = mm;
    wl[((m * NKS + ks) * 3 + 2) * 64 + l] = lo;
  }
  float bias[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[m][e] = p.bias != nullptr ? p.bias[16 * m + 4 * g + e] : 0.f;
  asm volatile("" ::"v"(bias[0][0]), "v"(bias[0][1]), "v"(bias[0][2]), "v"(bias[0][3]));     // (consumed ahead of the loop: dn_lds3.hip)
  const KResult& R = p.out[0];
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(R.p, 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t rin[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) rin[o] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in[o < p.n_in ? o : 0].p), 0, 0x80000000u, 0x00020000);

  // ---- staging items: (plane, row, float4 group) -> one aligned 16-byte load; constants per thread
  f32x4 stg[Cfg::ROUNDS];
  int gofs[Cfg::ROUNDS], rowcol[Cfg::ROUNDS], ldst[Cfg::ROUNDS], opsel[Cfg::ROUNDS];
#pragma unroll
  for (int r = 0; r < Cfg::ROUNDS; ++r) {
    const int it = tid + 512 * r;
    const int q4 = it % Cfg::COLS4, rr = it / Cfg::COLS4;
    const int row = rr % SK_ROWS, plane = rr / SK_ROWS;
    const bool live = it < Cfg::ITEMS;
    const int op = live ? geo.plane_op[plane] : 0, c = live ? geo.plane_c[plane] : 0;
    const KOperand& S = p.in[0];                          // (all operands have the strides of the first: stemk_form)
    gofs[r] = (row * (int)S.sh + 4 * q4 + c * (int)S.sc) * 4;
    rowcol[r] = live ? (row << 16 | (4 * q4)) : -1;
    ldst[r] = (plane * PLANE + row * COLSP + 4 * q4) * 4;
    opsel[r] = op;
  }
  auto issue_loads = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int iy0 = 2 * tyb * SK_TH - 3, ix0 = 2 * txb * TW - 4;
#pragma unroll
    for (int r = 0; r < Cfg::ROUNDS; ++r) {
      const int iy = iy0 + (rowcol[r] >> 16), ix = ix0 + (rowcol[r] & 0xffff);
      const bool ok = rowcol[r] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int op = opsel[r];
      const KOperand& S = p.in[0];
      const int off = (n * (int)S.sn + iy0 * (int)S.sh + ix0) * 4 + gofs[r];
      f32x4 v;
      if (op == 0) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[0], ok ? off : -1, 0, 0));
      else if (op == 1) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[1], ok ? off : -1, 0, 0));
      else v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[2], ok ? off : -1, 0, 0));
      stg[r] = v;
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  // B slot of K-step ks for this lane: combination 4 ks + g (dead combinations read plane 0 / row 0: their weights are zero)
  const int lane_out = (j * (int)R.sw + 4 * g) * 4;
  if (band_lo + local < band_hi) issue_loads(band_lo + local);
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int gy0 = tyb * SK_TH, gx0 = txb * TW;
#pragma unroll
    for (int r = 0; r < Cfg::ROUNDS; ++r)
      if (rowcol[r] >= 0) *reinterpret_cast<f32x4*>(lds + ldst[r]) = stg[r];
    __syncthreads();
    if (t + nlocal < band_hi) issue_loads(t + nlocal);

    float sv[MT][4], qv[MT][4], pv[MT][4];
    const int tbase = (n * (int)R.sn + gy0 * (int)R.sh + gx0 * (int)R.sw) * 4;
#pragma unroll 1
    for (int i = 0; i < Cfg::PT; i += 2) {
      f32x4 acc[2][MT];
      int pbase[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = i + u;
        const int ty = 2 * wave + pt / (TW / 16), tx16 = pt % (TW / 16);
        pbase[u] = (2 * ty) * COLSP + 2 * (tx16 * 16 + j);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 2
      for (int ks = 0; ks < NKS; ++ks) {
        int combo = 4 * ks + g;
        combo = combo < Cfg::NCOMBO ? combo : 0;
        const int plane = combo / 7, r7 = combo - 7 * plane;
        const int boff = plane * PLANE + r7 * COLSP;
        bf16x8 b[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2* src = reinterpret_cast<const f32x2*>(pl + boff + pbase[u]);
          const f32x2 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
          const float v[8] = {x0[0], x0[1], x1[0], x1[1], x2[0], x2[1], x3[0], x3[1]};
          sk_split3(v, b[u][0], b[u][1], b[u][2]);
        }
        bf16x8 a[MT][3];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int P = 0; P < 3; ++P) a[m][P] = wl[((m * NKS + ks) * 3 + P) * 64 + lane];
        constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], b[u][BS[q]], acc[u][m], 0, 0, 0);
      }
      if (p.bn_partial != nullptr) {
        if (i == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              pv[m][e] = __shfl(acc[0][m][e], lane & 48);       // the pivot: this channel's value at the wave's first pixel
              sv[m][e] = 0.f;
              qv[m][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dlt = acc[u][m][e] - pv[m][e];
              sv[m][e] += dlt;
              qv[m][e] = fmaf(dlt, dlt, qv[m][e]);
            }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = i + u;
        const int ty = 2 * wave + pt / (TW / 16), tx16 = pt % (TW / 16);
        const int off = tbase + (ty * (int)R.sh + tx16 * 16 * (int)R.sw) * 4 + lane_out;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 w4 = sk_act4(acc[u][m] + f32x4{bias[m][0], bias[m][1], bias[m][2], bias[m][3]}, p.act, p.act_p0, p.act_p1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rout, off + 64 * m, 0, 0);
        }
      }
    }
    if constexpr (TW == 32) {
      if (p.bn_partial != nullptr) {
        // (sum, M2 about the mean) of the wave's 64 pixels per channel from the pivot-centred sums: S = s + 64 pv, M2 = q - s^2 / 64
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float s1 = sk_row16_sum(sv[m][e]), q1 = sk_row16_sum(qv[m][e]);
            if (j == 0)
              *reinterpret_cast<f32x2*>(&wstat[((wave * MT + m) * 16 + 4 * g + e) * 2]) = f32x2{fmaf(64.f, pv[m][e], s1), q1 - s1 * s1 * (1.f / 64.f)};
          }
      }
    }
    __syncthreads();
    if constexpr (TW == 32) {
      if (p.bn_partial != nullptr && tid < 4 * MT * 16) {
        // waves (2 h, 2 h + 1) make one 128-pixel row of the statistics table (Chan's merge of two 64-pixel halves); the next tile's
        // wstat writes come after the next barrier, so these reads cannot race them
        const int h = tid / (MT * 16), c = tid % (MT * 16);
        const f32x2 a = *reinterpret_cast<const f32x2*>(&wstat[((2 * h) * MT * 16 + c) * 2]);
        const f32x2 b2 = *reinterpret_cast<const f32x2*>(&wstat[((2 * h + 1) * MT * 16 + c) * 2]);
        const float dm = (b2[0] - a[0]) * (1.f / 64.f);
        *reinterpret_cast<f32x2*>(p.bn_partial + ((long long)(4 * t + h) * p.Ntot + c) * 2) = f32x2{a[0] + b2[0], a[1] + b2[1] + dm * dm * 32.f};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// dW[co][k] = sum over pixels of dy[pixel][co] * x[2 pixel + tap][c], k in the kernel's (plane, tap row, tap column + 1) order: 8 x 7 x planes
// columns in N tiles of 16, the contraction over PIXELS in K-steps of 32 (lane group g = eight consecutive output pixels of one row).
//   * dy (NHWC) is transposed by the staging threads into channel-planar three-piece planes (8 pixels per 16-byte word), as in
//     lds3_wgrad_stem_kernel; x is fetched like the forward kernel's tile (aligned 16-byte loads along x) and stored with its columns
//     de-interleaved by PARITY, so that the eight stride-2 samples of a lane are eight consecutive floats (8 ds_read_b32, split in registers);
//   * a wave owns whole N tiles (columns) for ALL pixels: accumulators stay in registers across the persistent block's tiles, no wave meets
//     another; one slab per block in the packed-weight K order, folded by wgrad_reduce_kernel in a fixed order.
template <int MT, int TW, int NPL>
struct SkwCfg {
  static constexpr int TH = 4;
  static constexpr int ROWS = 2 * (TH - 1) + 7;                 // 13 input rows
  static constexpr int COLS4 = (2 * TW + 8) / 4;
  static constexpr int COLSP = 4 * COLS4, HALF = COLSP / 2;
  static constexpr int PLANE = ROWS * COLSP;                    // floats
  static constexpr int NCOMBO = 7 * NPL;
  static constexpr int NT = (8 * NCOMBO + 15) / 16;             // N tiles of 16 columns
  static constexpr int NTW = (NT + 3) / 4;                      // per wave
  static constexpr int PX = TH * TW;                            // pixels per tile
  static constexpr int KSTEPS = PX / 32;
  static constexpr int GSTR = PX * 2 + 16;                      // bytes between the dy planes of two output channels
  static constexpr int GPIECE = 16 * MT * GSTR;
  static constexpr size_t G_B = (size_t)3 * GPIECE;
  static constexpr size_t LDS = G_B + (size_t)NPL * PLANE * 4;
  static constexpr int XITEMS = NPL * ROWS * COLS4, XROUNDS = (XITEMS + 255) / 256;
  static constexpr int GITEMS = 4 * MT * (PX / 8);              // (channel quad, 8-pixel group)
  static_assert(GITEMS <= 256, "one dy item per thread");
  static_assert(2 * LDS <= 160 * 1024, "two blocks per CU");
};

template <int MT, int TW, int NPL>
__global__ void __launch_bounds__(256, 2) stemk_wgrad_kernel(const IgemmParams p, const SkGeo geo, long long slab_floats) {
  using Cfg = SkwCfg<MT, TW, NPL>;
  constexpr int COLSP = Cfg::COLSP, HALF = Cfg::HALF, PLANE = Cfg::PLANE, NTW = Cfg::NTW;
  extern __shared__ __align__(16) char lds[];
  char* Gp = lds;
  float* Ip = reinterpret_cast<float*>(lds + Cfg::G_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KOperand& S = p.in[0];                              // (all operands have the strides of the first: stemk_form)
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t rin[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) rin[o] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in[o < p.n_in ? o : 0].p), 0, 0x80000000u, 0x00020000);

  f32x4 acc[NTW][MT];
#pragma unroll
  for (int u = 0; u < NTW; ++u)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's column of N tile wave + 4 u: (combination, slot e) -> offset of its sample for pixel (row 0, column 0) in the planes
  int ioff[NTW];
  bool ilive[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int col = 16 * (wave + 4 * u) + j, combo = col >> 3, e = col & 7;
    ilive[u] = combo < Cfg::NCOMBO && e != 0 && wave + 4 * u < Cfg::NT;
    const int plane = ilive[u] ? combo / 7 : 0, r = ilive[u] ? combo % 7 : 0;
    ioff[u] = plane * PLANE + r * COLSP + (e & 1) * HALF + (e >> 1);
  }

  // ---- staging roles
  f32x4 vg[8], vx[Cfg::XROUNDS];
  const int gq = tid % (4 * MT), gpx = tid / (4 * MT);            // dy item: channel quad, 8-pixel group (linear pixel index / 8)
  int xofs[Cfg::XROUNDS], xrc[Cfg::XROUNDS], xdst[Cfg::XROUNDS], xop[Cfg::XROUNDS];
#pragma unroll
  for (int r = 0; r < Cfg::XROUNDS; ++r) {
    const int it = tid + 256 * r;
    const int q4 = it % Cfg::COLS4, rr = it / Cfg::COLS4;
    const int row = rr % Cfg::ROWS, plane = rr / Cfg::ROWS;
    const bool live = it < Cfg::XITEMS;
    const int op = live ? geo.plane_op[plane] : 0, c = live ? geo.plane_c[plane] : 0;
    xofs[r] = (row * (int)S.sh + 4 * q4 + c * (int)S.sc) * 4;
    xrc[r] = live ? (row << 16 | (4 * q4)) : -1;
    xdst[r] = plane * PLANE + row * COLSP + 2 * q4;             // parity 0 half; parity 1 at + HALF
    xop[r] = op;
  }
  auto issue_loads = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, q1 = t / geo.tilesX;
    const int tyb = q1 % geo.tilesY, n = q1 / geo.tilesY;
    {
      const int lin = 8 * gpx, row = lin / TW, col = lin - row * TW;
      const int gy = tyb * Cfg::TH + row, gx = txb * TW + col;
      const int base = (((n * p.GH + gy) * p.GW + gx) * p.Ntot + 4 * gq) * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        vg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, tid < Cfg::GITEMS ? base + i * p.Ntot * 4 : -1, 0, 0));
    }
    const int iy0 = 2 * tyb * Cfg::TH - 3, ix0 = 2 * txb * TW - 4;
#pragma unroll
    for (int r = 0; r < Cfg::XROUNDS; ++r) {
      const int iy = iy0 + (xrc[r] >> 16), ix = ix0 + (xrc[r] & 0xffff);
      const bool ok = xrc[r] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int off = (n * (int)S.sn + iy0 * (int)S.sh + ix0) * 4 + xofs[r];
      f32x4 v;
      if (xop[r] == 0) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[0], ok ? off : -1, 0, 0));
      else if (xop[r] == 1) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[1], ok ? off : -1, 0, 0));
      else v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin[2], ok ? off : -1, 0, 0));
      vx[r] = v;
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
    if (tid < Cfg::GITEMS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v[8] = {vg[0][e], vg[1][e], vg[2][e], vg[3][e], vg[4][e], vg[5][e], vg[6][e], vg[7][e]};
        bf16x8 h, m, l;
        sk_split3(v, h, m, l);
        char* dst = Gp + (4 * gq + e) * Cfg::GSTR + gpx * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + Cfg::GPIECE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * Cfg::GPIECE) = l;
      }
    }
#pragma unroll
    for (int r = 0; r < Cfg::XROUNDS; ++r) {
      if (xrc[r] >= 0) {
        *reinterpret_cast<f32x2*>(Ip + xdst[r]) = f32x2{vx[r][0], vx[r][2]};
        *reinterpret_cast<f32x2*>(Ip + xdst[r] + HALF) = f32x2{vx[r][1], vx[r][3]};
      }
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
  if (band_lo + local < band_hi) issue_loads(band_lo + local);
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    store_lds();
    __syncthreads();
    if (t + nlocal < band_hi) issue_loads(t + nlocal);
#pragma unroll 1
    for (int ks = 0; ks < Cfg::KSTEPS; ++ks) {
      const int lin = 32 * ks + 8 * g, orow = lin / TW, ox0 = lin - orow * TW;     // this lane group's eight pixels
      bf16x8 a[MT][3];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int P = 0; P < 3; ++P) a[m][P] = *reinterpret_cast<const bf16x8*>(Gp + P * Cfg::GPIECE + (16 * m + j) * Cfg::GSTR + lin * 2);
      const float* xb = Ip + 2 * orow * COLSP + ox0;
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        if (wave + 4 * u < Cfg::NT) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = xb[ioff[u] + i];
            v[i] = ilive[u] ? x : 0.f;
          }
          bf16x8 b[3];
          sk_split3(v, b[0], b[1], b[2]);
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], b[BS[q]], acc[u][m], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- every wave owns its columns: straight into the block's slab ws[block][co][k] (k = packed-weight order)
  const int Kp = p.ph[0].nchunks * kChunk;
  float* slab = p.ws + (long long)blockIdx.x * slab_floats;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    if (ilive[u]) {
      const int col = 16 * (wave + 4 * u) + j, combo = col >> 3, e = col & 7;
      const int plane = combo / 7, r = combo - 7 * plane;
      const int op = geo.plane_op[plane], c = geo.plane_c[plane];
      const int k = geo.kbase[op] + (7 * r + (e - 1)) * p.in[op].C + c;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = 16 * m + 4 * g + q;
          if (co < p.Ntot) slab[(long long)co * Kp + k] = acc[u][m][q];
        }
    }
  }
}

// 0: none, 1: ResNet conv1 form <4, 32, 3>, 2: PoseExpNet conv1 form <1, 16, 9>
static int stemk_form(const dn_conv_desc* d, const IgemmParams& p, SkGeo* geo) {
  if (knobs().no_lds3 || p.compute != DN_COMPUTE_F32X3) return 0;
  if (d->kind != DN_CONV_FWD || d->R != 7 || d->S != 7 || d->stride != 2 || d->pad != 3 || d->pad_mode != 0 || d->dilation > 1) return 0;
  if (p.n_in < 1 || p.n_in > 3 || p.n_out != 1 || p.nphases != 1 || p.ph[0].ntaps != 49) return 0;
  if ((d->IH & 1) || (d->IW & 3) || p.GH * 2 != d->IH || p.GW * 2 != d->IW) return 0;
  for (int t = 0; t < 49; ++t)
    if (p.tdy[t] != t / 7 - 3 || p.tdx[t] != t % 7 - 3) return 0;
  int npl = 0, kb = 0;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!(o.C >= 1 && o.C <= 3 && o.up == 0 && o.scale == nullptr && o.small && o.sw == 1)) return 0;
    if ((o.sh & 3) || (o.sc & 3) || (o.sn & 3) || (reinterpret_cast<uintptr_t>(o.p) & 15)) return 0;       // aligned float4 loads along x
    if (o.sh != p.in[0].sh || o.sc != p.in[0].sc || o.sn != p.in[0].sn) return 0;                          // one set of strides for all operands
    geo->kbase[i] = kb;
    kb += (49 * o.C + kChunk - 1) / kChunk * kChunk;
    for (int c = 0; c < o.C && npl < 9; ++c) {
      geo->plane_op[npl] = i;
      geo->plane_c[npl] = c;
      ++npl;
    }
  }
  const KResult& r = p.out[0];
  if (!(r.linear && !r.accumulate && r.n_begin == 0 && (r.sw & 3) == 0 && (r.sh & 3) == 0 && (r.sn & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(r.p) & 15) == 0 && (long long)p.N * r.sn * 4 < (1ll << 31)))
    return 0;
  int form = 0, TW = 0;
  if (npl == 3 && p.n_in == 1 && p.Ntot == 64 && p.GW % 32 == 0) { form = 1; TW = 32; }
  else if (npl == 9 && p.n_in == 3 && p.Ntot == 16 && p.GW % 16 == 0 && p.bn_partial == nullptr) { form = 2; TW = 16; }
  if (!form || p.GH % SK_TH != 0) return 0;
  geo->tilesX = p.GW / TW;
  geo->tilesY = p.GH / SK_TH;
  geo->ntiles = p.N * geo->tilesX * geo->tilesY;
  geo->per_xcd = (geo->ntiles + 7) / 8;
  if (geo->ntiles < 128) return 0;
  return form;
}

// the weight-gradient plan of the same layers (p.g = dy, p.in = the images, grid = output pixels)
static int stemk_wgrad_form(const dn_conv_desc* d, const IgemmParams& p, SkGeo* geo) {
  if (knobs().no_lds3 || norm_compute(d->compute) != DN_COMPUTE_F32X3) return 0;
  if (d->kind != DN_CONV_FWD || d->R != 7 || d->S != 7 || d->stride != 2 || d->pad != 3 || d->pad_mode != 0 || d->dilation > 1) return 0;
  if (p.n_in < 1 || p.n_in > 3 || p.ph[0].ntaps != 49) return 0;
  if ((d->IH & 1) || (d->IW & 3) || p.GH * 2 != d->IH || p.GW * 2 != d->IW) return 0;
  if ((long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31) || (reinterpret_cast<uintptr_t>(p.g) & 15) || (p.Ntot & 3)) return 0;
  for (int t = 0; t < 49; ++t)
    if (p.tdy[t] != t / 7 - 3 || p.tdx[t] != t % 7 - 3) return 0;
  int npl = 0, kb = 0;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!(o.C >= 1 && o.C <= 3 && o.up == 0 && o.scale == nullptr && o.small && o.sw == 1)) return 0;
    if ((o.sh & 3) || (o.sc & 3) || (o.sn & 3) || (reinterpret_cast<uintptr_t>(o.p) & 15)) return 0;
    if (o.sh != p.in[0].sh || o.sc != p.in[0].sc || o.sn != p.in[0].sn) return 0;
    geo->kbase[i] = kb;
    kb += (49 * o.C + kChunk - 1) / kChunk * kChunk;
    for (int c = 0; c < o.C && npl < 9; ++c) {
      geo->plane_op[npl] = i;
      geo->plane_c[npl] = c;
      ++npl;
    }
  }
  int form = 0, TW = 0;
  if (npl == 3 && p.n_in == 1 && p.Ntot == 64 && p.GW % 32 == 0) { form = 1; TW = 32; }
  else if (npl == 9 && p.n_in == 3 && p.Ntot == 16 && p.GW % 16 == 0) { form = 2; TW = 16; }
  if (!form || p.GH % 4 != 0) return 0;
  geo->tilesX = p.GW / TW;
  geo->tilesY = p.GH / 4;
  geo->ntiles = p.N * geo->tilesX * geo->tilesY;
  geo->per_xcd = (geo->ntiles + 7) / 8;
  if (geo->ntiles < 512) return 0;
  return form;
}

static int stemk_wgrad_blocks(const SkGeo& geo) {
  int blocks = geo.ntiles < 512 ? geo.ntiles : 512;
  return (blocks + 7) / 8 * 8;
}

bool stemk_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  SkGeo geo;
  return stemk_wgrad_form(d, p, &geo) != 0;
}

size_t stemk_wgrad_workspace_bytes(const dn_conv_desc* d, const IgemmParams& p) {
  SkGeo geo;
  if (!stemk_wgrad_form(d, p, &geo)) return 0;
  return (size_t)stemk_wgrad_blocks(geo) * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
}

template <int MT, int TW, int NPL>
static int stemk_wgrad_launch(IgemmParams& p, const SkGeo& geo, float* dw, hipStream_t stream) {
  using Cfg = SkwCfg<MT, TW, NPL>;
  auto kernel = stemk_wgrad_kernel<MT, TW, NPL>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(stemk_wgrad_kernel, %zu): %s", (size_t)Cfg::LDS, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  const int blocks = stemk_wgrad_blocks(geo);
  const long long slab = (long long)p.Npad * p.ph[0].nchunks * kChunk;
  DN_LAUNCH(kernel, dim3(blocks), dim3(256), (size_t)Cfg::LDS, stream, p, geo, slab);
  set_last_kernel("dn::stemk_wgrad_kernel<%d, %d, %d>", MT, TW, NPL);
  int rc = check_launch("stemk_wgrad_kernel");
  if (rc != DN_OK) return rc;
  p.splits = blocks;
  return launch_wgrad_reduce(p, dw, stream);
}

int launch_stemk_wgrad(const dn_conv_desc* d, IgemmParams& p, float* dw, hipStream_t stream) {
  SkGeo geo;
  switch (stemk_wgrad_form(d, p, &geo)) {
    case 1: return stemk_wgrad_launch<4, 32, 3>(p, geo, dw, stream);
    case 2: return stemk_wgrad_launch<1, 16, 9>(p, geo, dw, stream);
    default: set_error("launch_stemk_wgrad: no form"); return DN_ERR_UNSUPPORTED;
  }
}

bool stemk_conv_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  SkGeo geo;
  return stemk_form(d, p, &geo) != 0;
}

template <int MT, int TW, int NPL>
static int stemk_launch(const IgemmParams& p, const SkGeo& geo, hipStream_t stream) {
  using Cfg = SkCfg<MT, TW, NPL>;
  auto kernel = stemk_conv_kernel<MT, TW, NPL>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(stemk_conv_kernel, %zu): %s", (size_t)Cfg::LDS, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  int blocks = geo.ntiles < 256 ? geo.ntiles : 256;
  blocks = (blocks + 7) / 8 * 8;
  DN_LAUNCH(kernel, dim3(blocks), dim3(512), (size_t)Cfg::LDS, stream, p, geo);
  set_last_kernel("dn::stemk_conv_kernel<%d, %d, %d>", MT, TW, NPL);
  return check_launch("stemk_conv_kernel");
}

int launch_stemk_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream) {
  SkGeo geo;
  switch (stemk_form(d, p, &geo)) {
    case 1: return stemk_launch<4, 32, 3>(p, geo, stream);
    case 2: return stemk_launch<1, 16, 9>(p, geo, stream);
    default: set_error("launch_stemk_conv: no form"); return DN_ERR_UNSUPPORTED;
  }
}

}  // namespace dn
