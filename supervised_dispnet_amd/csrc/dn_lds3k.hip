// The 8-wave form of the LDS-resident three-piece convolutions (dn_lds3.hip) for layers whose weights do not fit one wave's registers:
// iconv1 (32 + 64 + 1 -> 32, 3x3; reference models/Disp_vgg_BN.py:108,182-183), upconv1 (ConvTranspose2d(64, 32, 4, 2, 1), :104,181) forward
// and input gradient.  Same rules -- input tile + halo once in LDS as three exact bf16 pieces, D = W . X on v_mfma_f32_16x16x32_bf16 with
// the output channels as M, weights in registers of persistent blocks, float4 stores from the C/D layout -- with ONE block of eight waves
// per CU whose waves take roles along three axes:
//   phase (the four sub-pixel phases of a stride-2 transposed convolution) x M tile (16 output channels) x K quarter.
// A K-split role holds only its share of the K-steps (iconv1: 28 K-steps of 32 = 336 registers of weights per M tile -> 7 K-steps = 84
// per wave); every wave walks ALL 16-pixel tiles of the block's tile for its role, the partial accumulators of a (phase, M tile) group
// meet in LDS (over the input planes, which are dead by then) and each wave of the group finishes a share of the pixel tiles.
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lk_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

__device__ __forceinline__ f32x4 lk_act4(f32x4 v, int act, float p0, float p1) {           // one wave-uniform branch for four values
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

__device__ __forceinline__ void lk_split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
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

struct LkGeo {
  int tilesX, tilesY, ntiles, per_xcd;
  int dy0, dx0;
  int cg0;                         // 8-channel groups of main operand 0 (operand 1 has CGT - cg0)
  long long* dbg;                  // DN_LDS3_DBG (tools/lds3_timing.py): per-wave phase ticks, 8 per wave; nullptr otherwise
};

// CGT: 8-channel groups of the main operands together; NKS: K-steps per role; HAS1: trailing 1-channel operand; PH x MT x KQ = 8 roles;
// NOPS: main operands (1 or 2).
template <int CGT, int NKS, bool HAS1, int PH, int MT, int KQ, int STRIDE, int TH, int TW, int ROWS, int COLS, int NOPS>
struct LkCfg {
  static_assert(PH * MT * KQ == 8, "eight waves, one role each");
  static constexpr int HALFC = (COLS + 1) / 2;
  static constexpr int COLSP = STRIDE == 2 ? 2 * HALFC : COLS;
  static constexpr int PLANE = ROWS * COLSP * 16;
  // 16-byte bank group of a plane relative to its neighbour: 1 where the staging lanes run over the channel groups of one pixel, 4 where
  // they run over (4 groups x 4 pixels) -- the two-operand form, whose waves each stage ONE operand (see the kernel)
  static constexpr int CGSTRIDE = ((PLANE + 255) / 256) * 256 + (NOPS == 2 ? 64 : 16);
  static constexpr int PSTRIDE = CGT * CGSTRIDE;
  static constexpr int DPLANE = HAS1 ? ROWS * COLS * 4 : 0;
  static constexpr int PTB = TH * TW / 16;
  static constexpr size_t RED = KQ > 1 ? (size_t)2 * 8 * 2 * 64 * 16 : 0;        // partial accumulators [buffer][wave][tile of the pair][lane] float4
  static constexpr size_t PLANES = ((size_t)3 * PSTRIDE + DPLANE + 15) / 16 * 16;
  static constexpr size_t LDS = PLANES + RED;
  static_assert(PTB % 2 == 0, "pixel tiles in pairs");
  static_assert(LDS <= 160 * 1024, "one block per CU");
};

template <int CGT, int NKS, bool HAS1, int PH, int MT, int KQ, int STRIDE, int TH, int TW, int ROWS, int COLS, int NOPS, bool DBG = false>
__global__ void __launch_bounds__(512, 2) lds3k_conv_kernel(const IgemmParams p, const LkGeo geo) {
  using Cfg = LkCfg<CGT, NKS, HAS1, PH, MT, KQ, STRIDE, TH, TW, ROWS, COLS, NOPS>;
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int kq = wave % KQ, mt = (wave / KQ) % MT, z = wave / (KQ * MT);      // K quarter, M tile, phase
  const KPhase ph = p.ph[z];
  const int ntaps = ph.ntaps, Kp = ph.nchunks * kChunk;
  const int nslots = ntaps * CGT;
  const int C0 = 8 * geo.cg0, C1 = 8 * (CGT - geo.cg0);
  const int kb1 = ((ntaps * C0 + kChunk - 1) / kChunk) * kChunk;                                     // packed K offset of main operand 1
  const int kbs = NOPS == 2 ? kb1 + ((ntaps * C1 + kChunk - 1) / kChunk) * kChunk : kb1;             // ... of the 1-channel operand

  // ---- this wave's weights as A fragments: row = output channel 16 * mt + j, K slot 4 * (kq * NKS + ks) + g = tap * CGT + group
  bf16x8 wa[NKS][3];
  int boff[NKS];
  {
    const int n = 16 * mt + j;
    const float* wrow = p.w + ph.w_off + (long long)n * Kp;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int s = 4 * (kq * NKS + ks) + g;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      int off = 0;
      if (s < nslots) {
        const int tap = s / CGT, cg = s - tap * CGT;
        if (n < p.Ntot) {
          const float* src = (NOPS == 2 && cg >= geo.cg0) ? wrow + kb1 + tap * C1 + 8 * (cg - geo.cg0) : wrow + tap * C0 + 8 * cg;
          const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        }
        const int ey = (int)p.tdy[ph.tap0 + tap] - geo.dy0, ex = (int)p.tdx[ph.tap0 + tap] - geo.dx0;
        const int cterm = STRIDE == 2 ? ((ex & 1) * Cfg::HALFC + (ex >> 1)) : ex;
        off = cg * Cfg::CGSTRIDE + (ey * Cfg::COLSP + cterm) * 16;
      } else if (HAS1 && n < p.Ntot && s - nslots < 2) {
        const int t0 = (s - nslots) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (t0 + e < ntaps) v[e] = wrow[kbs + t0 + e];
      }
      boff[ks] = off;
      lk_split3(v, wa[ks][0], wa[ks][1], wa[ks][2]);
    }
  }
  int doff[8];
  if constexpr (HAS1) {
    const int t0 = (g & 1) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = t0 + e < ntaps ? t0 + e : 0;
      doff[e] = ((int)p.tdy[ph.tap0 + t] - geo.dy0) * COLS + ((int)p.tdx[ph.tap0 + t] - geo.dx0);
    }
  }
  const int n0 = 16 * mt + 4 * g;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (p.bias != nullptr && n0 + e < p.Ntot) ? p.bias[n0 + e] : 0.f;
  int seg = 0;
  if (p.n_out > 1 && n0 >= p.out[1].n_begin) seg = 1;
  if (p.n_out > 2 && n0 >= p.out[2].n_begin) seg = 2;
  const KResult R = p.out[seg];
  const int cl = n0 - R.n_begin;
  const bool vec_store = (n0 + 3 < p.Ntot) && (cl + 4 <= R.C) && ((R.sw & 3) == 0) && ((R.sh & 3) == 0) && ((R.sn & 3) == 0) && ((cl & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(R.p) & 15) == 0);

  const KOperand& S0 = p.in[0];
  const KOperand& S1 = p.in[NOPS == 2 ? 1 : 0];
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S0.p), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S1.p), 0, 0x80000000u, 0x00020000);
  const KOperand& SD = p.in[HAS1 ? NOPS : 0];
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SD.p), 0, 0x80000000u, 0x00020000);

  // staging items: (pixel of the input tile, 8-channel group) -> two float4 loads, one split, three 16-byte LDS words.  With two main
  // operands the items of operand 0 come first, padded to whole waves, so that every wave reads ONE operand through ONE descriptor (a
  // per-lane choice took two loads with complementary predicates and an add that waited for both in the issue phase); within an operand
  // sixteen consecutive lanes are 4 groups x 4 pixels: 128 contiguous bytes per pixel on the way in, sixteen bank groups on the way out.
  constexpr int NPX = ROWS * COLS;
  constexpr int ITEMS = NPX * CGT, ROUNDS = (ITEMS + (NOPS == 2 ? 63 : 0) + 511) / 512;
  static_assert(NOPS == 1 || NPX % 4 == 0, "pixels in fours");
  int item[ROUNDS];                                       // bit 31 live, bit 30 operand 1, group << 16, row << 8, column
  {
    const int n0pad = NOPS == 2 ? (NPX * geo.cg0 + 63) / 64 * 64 : 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = tid + 512 * r;
      int cg, px, live, sec = 0;
      if constexpr (NOPS == 2) {
        sec = it >= n0pad;
        const int i2 = sec ? it - n0pad : it, ncg = sec ? CGT - geo.cg0 : geo.cg0, ncgh = ncg >> 2;
        const int hi = i2 >> 4, px_hi = hi / ncgh, cg_hi = hi - px_hi * ncgh;
        cg = (sec ? geo.cg0 : 0) + 4 * cg_hi + (i2 & 3);
        px = 4 * px_hi + ((i2 >> 2) & 3);
        live = i2 < NPX * ncg;
      } else {
        cg = it % CGT;
        px = it / CGT;
        live = it < ITEMS;
      }
      const int row = px / COLS, col = px - row * COLS;
      item[r] = (int)((unsigned)(live ? 1 : 0) << 31 | (unsigned)(sec << 30) | (unsigned)((cg & 0xff) << 16) | (unsigned)((row & 0xff) << 8) | (unsigned)(col & 0xff));
    }
  }
  f32x4 va[ROUNDS], vb[ROUNDS];
  float dv = 0.f;
  auto issue_loads = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int iy0 = tyb * TH * STRIDE + geo.dy0, ix0 = txb * TW * STRIDE + geo.dx0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int cg = (item[r] >> 16) & 0xff, row = (item[r] >> 8) & 0xff, col = item[r] & 0xff;
      const int iy = iy0 + row, ix = ix0 + col;
      const bool ok = item[r] < 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      if (NOPS == 2 && ((__builtin_amdgcn_readfirstlane(item[r]) >> 30) & 1)) {          // (the operand is the wave's, dead lanes included)
        const int off1 = (n * (int)S1.sn + iy * (int)S1.sh + ix * (int)S1.sw + 8 * (cg - geo.cg0)) * 4;
        va[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, ok ? off1 : -1, 0, 0));
        vb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, ok ? off1 + 16 : -1, 0, 0));
      } else {
        const int off0 = (n * (int)S0.sn + iy * (int)S0.sh + ix * (int)S0.sw + 8 * cg) * 4;
        va[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, ok ? off0 : -1, 0, 0));
        vb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, ok ? off0 + 16 : -1, 0, 0));
      }
    }
    if constexpr (HAS1) {
      const int row = tid / COLS, col = tid - row * COLS;
      const int iy = iy0 + row, ix = ix0 + col;
      const bool ok = tid < ROWS * COLS && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int off = (n * (int)SD.sn + (iy >> SD.up) * (int)SD.sh + (ix >> SD.up) * (int)SD.sw) * 4;
      dv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsd, ok ? off : -1, 0, 0));
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (item[r] < 0) {
        const int cg = (item[r] >> 16) & 0xff, row = (item[r] >> 8) & 0xff, col = item[r] & 0xff;
        const int idx = STRIDE == 2 ? ((col & 1) * Cfg::HALFC + (col >> 1)) : col;
        const float v[8] = {va[r][0], va[r][1], va[r][2], va[r][3], vb[r][0], vb[r][1], vb[r][2], vb[r][3]};
        bf16x8 h, m, l;
        lk_split3(v, h, m, l);
        char* dst = lds + cg * Cfg::CGSTRIDE + (row * Cfg::COLSP + idx) * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + Cfg::PSTRIDE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * Cfg::PSTRIDE) = l;
      }
    }
    if constexpr (HAS1) {
      static_assert(!HAS1 || ROWS * COLS <= 512, "one round stages the 1-channel plane");
      float* dpl = reinterpret_cast<float*>(lds + 3 * Cfg::PSTRIDE);
      if (tid < ROWS * COLS) dpl[tid] = dv;
    }
  };
  // Fast result path (dn_lds3.hip): one plain float4-addressable result tensor below 2 GB -> a raw buffer store per pixel tile, lanes
  // out of range dropped through offset -1; and everything the epilogue reads that came through a vector-memory load is consumed HERE,
  // so that no in-order s_waitcnt vmcnt(0) for it lands between the next tile's loads and this tile's matrix instructions.
  const KResult& R0 = p.out[0];
  const bool fast_out = p.n_out == 1 && !R0.accumulate && R0.n_begin == 0 && (p.Ntot & 3) == 0 && ((R0.sw | R0.sh | R0.sn) & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(R0.p) & 15) == 0 && (long long)p.N * R0.sn * 4 < (1ll << 31);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(R0.p, 0, 0x80000000u, 0x00020000);
  const int lane_out = (j * p.osx * (int)R0.sw + n0) * 4;
  asm volatile("" ::"v"(R.p), "v"(R.sn), "v"(R.sh), "v"(R.sw), "v"(R.accumulate), "v"(bias[0]), "v"(bias[1]), "v"(bias[2]), "v"(bias[3]));
  // result of pixel tile `pt`: bias, activation, store (lane (j, g) holds output channels n0 .. n0 + 3 of grid point (gy, gx))
  auto emit = [&](int n, int gy0, int gx0, int pt, const f32x4& a) __attribute__((always_inline)) {
    const int ty = pt / (TW / 16), tx16 = pt - ty * (TW / 16);
    const int gy = gy0 + ty, gx = gx0 + tx16 * 16 + j;
    const int oy = gy * p.osy + ph.ooy, ox = gx * p.osx + ph.oox;
    if (fast_out) {
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      const bool ok = gy < p.GH && gx < p.GW && oy < p.OH && ox < p.OW && n0 < p.Ntot;
      const f32x4 w4 = lk_act4(a + f32x4{bias[0], bias[1], bias[2], bias[3]}, p.act, p.act_p0, p.act_p1);
      const int off = (n * (int)R0.sn + oy * (int)R0.sh + (gx0 + tx16 * 16) * p.osx * (int)R0.sw + ph.oox * (int)R0.sw) * 4 + lane_out;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rout, ok ? off : -1, 0, 0);
      return;
    }
    if (gy < p.GH && gx < p.GW && oy < p.OH && ox < p.OW && n0 < p.Ntot) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = lk_act(a[e] + bias[e], p.act, p.act_p0, p.act_p1);
      if (vec_store) {
        f32x4* o = reinterpret_cast<f32x4*>(R.p + (long long)n * R.sn + (long long)oy * R.sh + (long long)ox * R.sw + cl);
        f32x4 w4 = f32x4{v[0], v[1], v[2], v[3]};
        if (R.accumulate) w4 += *o;
        *o = w4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nn = n0 + e;
          if (nn < p.Ntot) {
            int sg = 0;
            if (p.n_out > 1 && nn >= p.out[1].n_begin) sg = 1;
            if (p.n_out > 2 && nn >= p.out[2].n_begin) sg = 2;
            const KResult& Q = p.out[sg];
            float* o = Q.p + (long long)n * Q.sn + (long long)oy * Q.sh + (long long)ox * Q.sw + (nn - Q.n_begin);
            *o = Q.accumulate ? *o + v[e] : v[e];
          }
        }
      }
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  long long tk[7] = {0, 0, 0, 0, 0, 0, 0}, c0t = 0, c1t = 0;   // DBG: load wait | split + LDS writes | barrier | load issue | matrix | exchange + stores | barrier
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr (DBG) { c1t = clock64(); tk[k] += c1t - c0t; c0t = c1t; }
  };
  if (band_lo + local < band_hi) issue_loads(band_lo + local);
  if constexpr (DBG) c0t = clock64();
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int gy0 = tyb * TH, gx0 = txb * TW;
    if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(0); }
    store_lds();
    stamp(1);
    __syncthreads();
    stamp(2);
    if (t + nlocal < band_hi) issue_loads(t + nlocal);
    stamp(3);

#pragma unroll 1
    for (int i = 0; i < Cfg::PTB; i += 2) {
      f32x4 acc[2];
      int lbase[2], dbase[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = i + u;
        const int ty = pt / (TW / 16), tx16 = pt - ty * (TW / 16);
        lbase[u] = ((ty * STRIDE) * Cfg::COLSP + tx16 * 16 + j) * 16;
        dbase[u] = ty * COLS + tx16 * 16 + j;
        acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 b[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int P = 0; P < 3; ++P) b[u][P] = *reinterpret_cast<const bf16x8*>(lds + P * Cfg::PSTRIDE + lbase[u] + boff[ks]);
        if constexpr (HAS1) {
          if (ks == NKS - 1 && kq == KQ - 1) {            // the 1-channel piece sits in the last K-step of the last K role
            const float* dpl = reinterpret_cast<const float*>(lds + 3 * Cfg::PSTRIDE);
            const int t0 = (g & 1) * 8;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float x = dpl[dbase[u] + doff[e]];
                v[e] = (t0 + e < ntaps) ? x : 0.f;
              }
              bf16x8 h, m, l;
              lk_split3(v, h, m, l);
              if (4 * (kq * NKS + ks) + g >= nslots) { b[u][0] = h; b[u][1] = m; b[u][2] = l; }
            }
          }
        }
        constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][AS[q]], b[u][BS[q]], acc[u], 0, 0, 0);
      }
      stamp(4);
      if constexpr (KQ == 1) {
        emit(n, gy0, gx0, i, acc[0]);
        emit(n, gy0, gx0, i + 1, acc[1]);
      } else {
        // the K roles of a (phase, M tile) group meet per PAIR of pixel tiles in a double-buffered exchange area behind the planes: pair i
        // uses buffer (i / 2) & 1; a buffer is rewritten two pairs later, i.e. after a barrier every reader of it has passed
        f32x4* red = reinterpret_cast<f32x4*>(lds + Cfg::PLANES) + ((i >> 1) & 1) * (8 * 2 * 64);
        red[(wave * 2 + 0) * 64 + lane] = acc[0];
        red[(wave * 2 + 1) * 64 + lane] = acc[1];
        __syncthreads();
        if (kq < 2) {                                     // waves kq = 0 / 1 of the group finish tile 0 / 1 of the pair (fixed order)
          const int w0 = wave - kq;
          f32x4 sum = red[((w0 + 0) * 2 + kq) * 64 + lane];
#pragma unroll
          for (int q = 1; q < KQ; ++q) sum += red[((w0 + q) * 2 + kq) * 64 + lane];
          emit(n, gy0, gx0, i + kq, sum);
        }
      }
      stamp(5);
    }
    __syncthreads();                                    // the next tile's staging overwrites the planes
    stamp(6);
  }
  if constexpr (DBG) {
    if (lane == 0) {
      long long* o = geo.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 7; ++k) o[k] = tk[k];
      o[7] = (band_hi - band_lo - local + nlocal - 1) / nlocal;
    }
  }
}

// ------------------------------------------------------------------------------------------------------ configurations
struct LkPick {
  int cfg;
  LkGeo geo;
};

static LkPick lk_pick(const dn_conv_desc* d, const IgemmParams& p) {
  LkPick r;
  r.cfg = 0;
  if (knobs().no_lds3 || p.compute != DN_COMPUTE_F32X3) return r;
  if (p.reflect || p.bn_partial != nullptr || p.bnb_y != nullptr || d->dilation > 1 || p.n_out < 1) return r;
  if (!(p.nphases == 1 || p.nphases == 4) || p.sy != p.sx || (p.sy != 1 && p.sy != 2)) return r;
  int nmain = p.n_in;
  bool has1 = false;
  if (p.n_in >= 2 && p.in[p.n_in - 1].C == 1) {
    const KOperand& b = p.in[p.n_in - 1];
    if (!(b.small && b.scale == nullptr)) return r;
    has1 = true;
    nmain = p.n_in - 1;
  }
  if (nmain < 1 || nmain > 2) return r;
  int cgt = 0;
  for (int i = 0; i < nmain; ++i) {
    const KOperand& a = p.in[i];
    if (!(a.vec && a.small && a.up == 0 && a.scale == nullptr && a.C % 8 == 0)) return r;
    cgt += a.C / 8;
  }
  int dy0 = 127, dy1 = -127, dx0 = 127, dx1 = -127, maxtaps = 0;
  for (int z = 0; z < p.nphases; ++z) {
    const KPhase& ph = p.ph[z];
    if (ph.ntaps < 1) return r;
    if (ph.ntaps > maxtaps) maxtaps = ph.ntaps;
    for (int t = 0; t < ph.ntaps; ++t) {
      const int dy = p.tdy[ph.tap0 + t], dx = p.tdx[ph.tap0 + t];
      dy0 = dy < dy0 ? dy : dy0; dy1 = dy > dy1 ? dy : dy1;
      dx0 = dx < dx0 ? dx : dx0; dx1 = dx > dx1 ? dx : dx1;
    }
  }
  const int spanY = dy1 - dy0 + 1, spanX = dx1 - dx0 + 1;
  int cfg = 0, TH = 4, TW = 32;
  // (CGT, NKS, HAS1, PH, MT, KQ, STRIDE, TH, TW, ROWS, COLS, NOPS)
  if (cgt == 12 && nmain == 2 && has1 && p.nphases == 1 && p.sy == 1 && maxtaps == 9 && spanY == 3 && spanX == 3 && p.Ntot > 16 && p.Ntot <= 32) cfg = 1;   // iconv1 forward
  else if (cgt == 8 && nmain == 1 && !has1 && p.nphases == 4 && p.sy == 1 && maxtaps <= 4 && spanY == 3 && spanX == 3 && p.Ntot > 16 && p.Ntot <= 32) cfg = 2;   // upconv1 forward
  else if (cgt == 4 && nmain == 1 && !has1 && p.nphases == 1 && p.sy == 2 && maxtaps <= 16 && spanY == 4 && spanX == 4 && p.Ntot > 32 && p.Ntot <= 64) cfg = 3;   // upconv1 input gradient
  if (!cfg) return r;
  r.cfg = cfg;
  r.geo.tilesX = (p.GW + TW - 1) / TW;
  r.geo.tilesY = (p.GH + TH - 1) / TH;
  r.geo.ntiles = p.N * r.geo.tilesX * r.geo.tilesY;
  if (r.geo.ntiles < 192) {             // one block per CU: a grid that leaves a quarter of the chip idle stays on the tiled kernels
    r.cfg = 0;
    return r;
  }
  r.geo.per_xcd = (r.geo.ntiles + 7) / 8;
  r.geo.dy0 = dy0;
  r.geo.dx0 = dx0;
  r.geo.cg0 = p.in[0].C / 8;
  r.geo.dbg = nullptr;
  if (nmain == 2 && ((r.geo.cg0 & 3) || ((cgt - r.geo.cg0) & 3))) r.cfg = 0;      // staging lanes: 4 groups x 4 pixels per operand
  return r;
}

bool lds3k_conv_eligible(const dn_conv_desc* d, const IgemmParams& p) { return lk_pick(d, p).cfg != 0; }

template <int CGT, int NKS, bool HAS1, int PH, int MT, int KQ, int STRIDE, int TH, int TW, int ROWS, int COLS, int NOPS, bool DBG = false>
static int lk_launch(const IgemmParams& p, const LkGeo& geo, hipStream_t stream) {
  using Cfg = LkCfg<CGT, NKS, HAS1, PH, MT, KQ, STRIDE, TH, TW, ROWS, COLS, NOPS>;
  auto kernel = lds3k_conv_kernel<CGT, NKS, HAS1, PH, MT, KQ, STRIDE, TH, TW, ROWS, COLS, NOPS, DBG>;
  const size_t lds = Cfg::LDS;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS %zu): %s", lds, hipGetErrorString(e));
      return DN_ERR_LAUNCH;
    }
  }
  int blocks = geo.ntiles < 256 ? geo.ntiles : 256;            // one resident block of eight waves per CU, persistent over the tiles
  blocks = (blocks + 7) / 8 * 8;
  DN_LAUNCH(kernel, dim3(blocks), dim3(512), lds, stream, p, geo);
  set_last_kernel("dn::lds3k_conv_kernel<%d, %d, %s, %d, %d, %d, %d, %d, %d, %d, %d, %d>", CGT, NKS, HAS1 ? "true" : "false", PH, MT, KQ, STRIDE, TH, TW,
                  ROWS, COLS, NOPS);
  return check_launch("lds3k_conv_kernel");
}

int launch_lds3k_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream) {
  LkPick k = lk_pick(d, p);
  if (k.cfg == 1 && knobs().lds3_dbg) {                  // phase timestamps (tools/lds3_timing.py)
    k.geo.dbg = reinterpret_cast<long long*>(knobs().wino_dbgptr);
    return lk_launch<12, 7, true, 1, 2, 4, 1, 4, 32, 6, 34, 2, true>(p, k.geo, stream);
  }
  switch (k.cfg) {
    case 1: return lk_launch<12, 7, true, 1, 2, 4, 1, 4, 32, 6, 34, 2>(p, k.geo, stream);
    case 2: return lk_launch<8, 8, false, 4, 2, 1, 1, 4, 32, 6, 34, 1>(p, k.geo, stream);
    case 3: return lk_launch<4, 8, false, 1, 4, 2, 2, 4, 32, 10, 66, 1>(p, k.geo, stream);
    default: set_error("launch_lds3k_conv: no configuration"); return DN_ERR_UNSUPPORTED;
  }
}

}  // namespace dn
