// LDS-resident weight gradients of the thin full-resolution 3x3 layers (round 4): iconv0 (16 + 1 -> 16) and the first encoder layer
// (3 -> 64 on the NCHW image); reference models/Disp_vgg_BN.py:84,110.  dW[co][tap][ci] = sum over pixels of dy[pixel][co] * x[pixel + tap][ci]:
// the contraction runs over PIXELS, so on v_mfma_f32_16x16x32_bf16 a lane must hold eight consecutive pixels of one channel -- the
// transpose of how NHWC tensors lie in memory.  Here the transposition is done once, by the staging threads:
//   * a thread loads 8 pixels x 4 channels (eight float4), splits every value into its three exact bf16 pieces (DN_COMPUTE_F32X3) and
//     writes, per channel and piece, the eight pixels as ONE 16-byte LDS word into channel-planar planes [piece][channel][row][col];
//   * a K-step is one 32-pixel row of the 8 x 32 tile: A fragment = dy plane of output channel j at pixels 8 g .. 8 g + 7 (aligned
//     ds_read_b128); B fragment of tap (dy, dx) = input plane of channel j, row + dy, the same eight columns shifted by dx: dx = 0 is the
//     aligned word, dx = -1 / +1 are funnel shifts (v_alignbit_b32) of that word with one neighbour dword -- every tap of a row comes out
//     of one b128 + two b32 reads per piece;
//   * the four waves of a block take different rows and keep D[co][ci] of all taps (10 or 8 accumulator tiles) in registers across ALL
//     tiles the persistent block walks; they meet once, at the end, through LDS, and the block writes one slab of the split-sum
//     workspace that wgrad_reduce_kernel (dn_conv.hip) folds in a fixed order -- deterministic.
// The 1-channel concat piece (nearest-x2 disparity) and the 3-channel image have their taps on the lanes (column n = tap, or tap * 3 + ci):
// those fragments are gathered from small fp32 planes (8 ds_read_b32) and split in registers.
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wg_split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
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

struct WgGeo {
  int tilesX, tilesY, ntiles, per_xcd;
  long long slab;                  // floats per block slab of the workspace: Npad * Kp
  int dbgmode;                     // DN_LDS3_DBG value: 2 = no loads, 3 = loads from a 64 KB window (timing ablations, wrong results)
  long long* dbg;                  // DN_LDS3_DBG=1 (tools/lds3_timing.py): per-wave phase ticks, 8 per wave; nullptr otherwise
};

constexpr int WG_TH = 8, WG_TW = 32;
// ---- iconv0 form: 16-channel NHWC input (+ optional 1-channel nearest-x2 piece), 16 output channels
constexpr int WG_GSTR = 528;                      // bytes between the dy planes of two output channels: 256 pixels x 2 + 16 (bank stagger)
constexpr int WG_GPIECE = 16 * WG_GSTR;
constexpr int WG_XROWB = 96;                      // one row of an input plane: 48 bf16, column c at element c + 8
constexpr int WG_XSTR = 10 * WG_XROWB + 16;       // 976: bytes between the planes of two input channels
constexpr int WG_XPIECE = 16 * WG_XSTR;
constexpr int WG_DCOLS = 36;                      // fp32 plane of the 1-channel piece: [10][36], column c at c + 1
constexpr int WG16_LDS = 3 * WG_GPIECE + 3 * WG_XPIECE + 10 * WG_DCOLS * 4;

template <bool HAS1, bool DBG = false>
__global__ void __launch_bounds__(256, 2) lds3_wgrad16_kernel(const IgemmParams p, const WgGeo geo) {
  extern __shared__ __align__(16) char lds[];
  char* Gp = lds;
  char* Xp = lds + 3 * WG_GPIECE;
  float* Dp = reinterpret_cast<float*>(lds + 3 * WG_GPIECE + 3 * WG_XPIECE);
  constexpr int NT = HAS1 ? 10 : 9;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KOperand& S = p.in[0];
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t r1 = rx;
  if constexpr (HAS1) r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in[1].p), 0, 0x80000000u, 0x00020000);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the 1-channel piece: lane n < 9 is tap n, offsets into the fp32 plane in floats
  const int dtap = j < 9 ? j : 0;
  const int doff = (dtap / 3) * WG_DCOLS + (dtap % 3);          // (dy + 1) * cols + (dx + 1), column c at c + 1

  f32x4 vg[8], vx[8];
  float dv[2] = {0.f, 0.f};
  auto issue_loads = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, q1 = t / geo.tilesX;
    const int tyb = q1 % geo.tilesY, n = q1 / geo.tilesY;
    const int gy0 = tyb * WG_TH, gx0 = txb * WG_TW;
    {   // dy: threads 0..127 take (row, 8-column group, channel quad)
      const int cq = tid & 3, pxg = tid >> 2, row = pxg >> 2, cgp = pxg & 3;
      const int gy = gy0 + row;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int gx = gx0 + 8 * cgp + i;
        const bool ok = tid < 128 && gy < p.GH && gx < p.GW;
        const int off = (((n * p.GH + gy) * p.GW + gx) * p.Ntot + 4 * cq) * 4;
        vg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : -1, 0, 0));
      }
    }
    {   // input rows gy0 - 1 .. gy0 + 8, columns gx0 - 8 .. gx0 + 39 in 8-column groups (only columns -1 .. 32 are fetched)
      const int cq = tid & 3, grp = tid >> 2, row = grp / 6, cgp = grp - row * 6;
      const int iy = gy0 - 1 + row;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ix = gx0 + 8 * cgp - 8 + i;
        const bool need = cgp == 0 ? i == 7 : (cgp == 5 ? i == 0 : true);
        const bool ok = tid < 240 && need && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int off = (n * (int)S.sn + iy * (int)S.sh + ix * (int)S.sw + 4 * cq) * 4;
        vx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : -1, 0, 0));
      }
    }
    if constexpr (HAS1) {
      const KOperand& S1 = p.in[1];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int it = tid + 256 * r;
        const int row = it / 34, col = it - row * 34;
        const int iy = gy0 - 1 + row, ix = gx0 - 1 + col;
        const bool ok = it < 340 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int off = (n * (int)S1.sn + (iy >> S1.up) * (int)S1.sh + (ix >> S1.up) * (int)S1.sw) * 4;
        dv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, ok ? off : -1, 0, 0));
      }
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
    if (tid < 128) {
      const int cq = tid & 3, pxg = tid >> 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v[8] = {vg[0][e], vg[1][e], vg[2][e], vg[3][e], vg[4][e], vg[5][e], vg[6][e], vg[7][e]};
        bf16x8 h, m, l;
        wg_split3(v, h, m, l);
        char* dst = Gp + (4 * cq + e) * WG_GSTR + pxg * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + WG_GPIECE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * WG_GPIECE) = l;
      }
    }
    if (tid < 240) {
      const int cq = tid & 3, grp = tid >> 2, row = grp / 6, cgp = grp - row * 6;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v[8] = {vx[0][e], vx[1][e], vx[2][e], vx[3][e], vx[4][e], vx[5][e], vx[6][e], vx[7][e]};
        bf16x8 h, m, l;
        wg_split3(v, h, m, l);
        char* dst = Xp + (4 * cq + e) * WG_XSTR + row * WG_XROWB + cgp * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + WG_XPIECE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * WG_XPIECE) = l;
      }
    }
    if constexpr (HAS1) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int it = tid + 256 * r;
        const int row = it / 34, col = it - row * 34;
        if (it < 340) Dp[row * WG_DCOLS + col] = dv[r];
      }
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
  long long tk[6] = {0, 0, 0, 0, 0, 0}, c0t = 0, c1t = 0;             // DBG: load wait | split + LDS writes | barrier | load issue | matrix loop | barrier
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr (DBG) { c1t = clock64(); tk[k] += c1t - c0t; c0t = c1t; }
  };
  if (band_lo + local < band_hi) issue_loads(band_lo + local);
  if constexpr (DBG) c0t = clock64();
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(0); }
    store_lds();
    stamp(1);
    __syncthreads();
    stamp(2);
    if (t + nlocal < band_hi) issue_loads(t + nlocal);
    stamp(3);
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;                                   // this wave's K-step: tile row r, pixels 8 g .. 8 g + 7 per lane group
      bf16x8 a[3];
#pragma unroll
      for (int P = 0; P < 3; ++P) a[P] = *reinterpret_cast<const bf16x8*>(Gp + P * WG_GPIECE + j * WG_GSTR + (r * 32 + 8 * g) * 2);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const char* rowb = Xp + j * WG_XSTR + (r + ty) * WG_XROWB + (8 * g + 8) * 2;
        u32x4 c[3];
        unsigned lf[3], rt[3];
#pragma unroll
        for (int P = 0; P < 3; ++P) {
          c[P] = *reinterpret_cast<const u32x4*>(rowb + P * WG_XPIECE);
          lf[P] = *reinterpret_cast<const unsigned*>(rowb + P * WG_XPIECE - 4);
          rt[P] = *reinterpret_cast<const unsigned*>(rowb + P * WG_XPIECE + 16);
        }
        bf16x8 bm[3], bz[3], bp[3];
#pragma unroll
        for (int P = 0; P < 3; ++P) {
          const u32x4 m4 = u32x4{__builtin_amdgcn_alignbit(c[P][0], lf[P], 16), __builtin_amdgcn_alignbit(c[P][1], c[P][0], 16),
                                 __builtin_amdgcn_alignbit(c[P][2], c[P][1], 16), __builtin_amdgcn_alignbit(c[P][3], c[P][2], 16)};
          const u32x4 p4 = u32x4{__builtin_amdgcn_alignbit(c[P][1], c[P][0], 16), __builtin_amdgcn_alignbit(c[P][2], c[P][1], 16),
                                 __builtin_amdgcn_alignbit(c[P][3], c[P][2], 16), __builtin_amdgcn_alignbit(rt[P], c[P][3], 16)};
          bm[P] = __builtin_bit_cast(bf16x8, m4);
          bz[P] = __builtin_bit_cast(bf16x8, c[P]);
          bp[P] = __builtin_bit_cast(bf16x8, p4);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          acc[3 * ty + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS[q]], bm[BS[q]], acc[3 * ty + 0], 0, 0, 0);
          acc[3 * ty + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS[q]], bz[BS[q]], acc[3 * ty + 1], 0, 0, 0);
          acc[3 * ty + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS[q]], bp[BS[q]], acc[3 * ty + 2], 0, 0, 0);
        }
      }
      if constexpr (HAS1) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = Dp[r * WG_DCOLS + 8 * g + e + doff];
          v[e] = j < 9 ? x : 0.f;
        }
        bf16x8 b[3];
        wg_split3(v, b[0], b[1], b[2]);
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[9] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS[q]], b[BS[q]], acc[9], 0, 0, 0);
      }
    }
    stamp(4);
    __syncthreads();
    stamp(5);
  }
  if constexpr (DBG) {
    if (lane == 0) {
      long long* o = geo.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 6; ++k) o[k] = tk[k];
      o[6] = (band_hi - band_lo - local + nlocal - 1) / nlocal;
    }
  }

  // ---- the four waves meet (fixed order) and the block writes its slab ws[block][co][k], k = tap * 16 + ci (operand 0), kbase1 + tap
  f32x4* red = reinterpret_cast<f32x4*>(lds);          // [wave][tile][lane]: 4 * 10 * 64 * 16 B = 40 KB (the planes are free now)
#pragma unroll
  for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[t];
  __syncthreads();
  const int Kp = p.ph[0].nchunks * kChunk;
  const int kbase1 = ((9 * 16 + kChunk - 1) / kChunk) * kChunk;
  float* slab = p.ws + (long long)blockIdx.x * geo.slab;
  for (int it = tid; it < NT * 64; it += 256) {
    const int t = it >> 6, l = it & 63;
    const f32x4 s = (red[(0 * NT + t) * 64 + l] + red[(1 * NT + t) * 64 + l]) + (red[(2 * NT + t) * 64 + l] + red[(3 * NT + t) * 64 + l]);
    const int jj = l & 15, gg = l >> 4;
    const int k = t < 9 ? t * 16 + jj : kbase1 + jj;
    if (t < 9 || jj < 9) {
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(long long)(4 * gg + e) * Kp + k] = s[e];
    }
  }
}

// ---- stem form: <= 3-channel image through its strides (NCHW), 64 output channels.  dy is staged half a tile (4 rows) at a time.
constexpr int WS_GSTR = 272;                      // 128 pixels x 2 + 16
constexpr int WS_GPIECE = 64 * WS_GSTR;
constexpr int WS_ICOLS = 36;                      // fp32 image planes [3][10][36], column c at c + 1
constexpr int WS_IPLANE = 10 * WS_ICOLS;
constexpr int WGS_LDS = 3 * WS_GPIECE + 3 * WS_IPLANE * 4;

__global__ void __launch_bounds__(256, 2) lds3_wgrad_stem_kernel(const IgemmParams p, const WgGeo geo) {
  extern __shared__ __align__(16) char lds[];
  char* Gp = lds;
  float* Ip = reinterpret_cast<float*>(lds + 3 * WS_GPIECE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KOperand& S = p.in[0];
  const int C = S.C, K = 9 * C;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);

  f32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // column 16 nt + j = tap * C + ci: offset of its pixel (0, 0) value in the image planes
  int ioff[2];
  bool ilive[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = 16 * nt + j;
    ilive[nt] = col < K;
    const int tap = ilive[nt] ? col / C : 0, ci = ilive[nt] ? col - tap * C : 0;
    ioff[nt] = ci * WS_IPLANE + (tap / 3) * WS_ICOLS + (tap % 3);
  }

  f32x4 vg[8];
  float vi[4];
  auto issue_g = [&](int t, int h) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, q1 = t / geo.tilesX;
    const int tyb = q1 % geo.tilesY, n = q1 / geo.tilesY;
    const int pxg = tid & 15, cq = tid >> 4, row = pxg >> 2, cgp = pxg & 3;
    const int gy = tyb * WG_TH + 4 * h + row;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gx = txb * WG_TW + 8 * cgp + i;
      const bool ok = gy < p.GH && gx < p.GW;
      const int off = (((n * p.GH + gy) * p.GW + gx) * p.Ntot + 4 * cq) * 4;
      vg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : -1, 0, 0));
    }
  };
  auto issue_img = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, q1 = t / geo.tilesX;
    const int tyb = q1 % geo.tilesY, n = q1 / geo.tilesY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int it = tid + 256 * r;
      const int c = it / 340, px = it - c * 340;
      const int row = px / 34, col = px - row * 34;
      const int iy = tyb * WG_TH - 1 + row, ix = txb * WG_TW - 1 + col;
      const bool ok = c < C && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int off = (n * (int)S.sn + iy * (int)S.sh + ix * (int)S.sw + c * (int)S.sc) * 4;
      vi[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? off : -1, 0, 0));
    }
  };
  auto store_g = [&]() __attribute__((always_inline)) {
    const int pxg = tid & 15, cq = tid >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v[8] = {vg[0][e], vg[1][e], vg[2][e], vg[3][e], vg[4][e], vg[5][e], vg[6][e], vg[7][e]};
      bf16x8 h, m, l;
      wg_split3(v, h, m, l);
      char* dst = Gp + (4 * cq + e) * WS_GSTR + pxg * 16;
      *reinterpret_cast<bf16x8*>(dst) = h;
      *reinterpret_cast<bf16x8*>(dst + WS_GPIECE) = m;
      *reinterpret_cast<bf16x8*>(dst + 2 * WS_GPIECE) = l;
    }
  };
  auto store_img = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int it = tid + 256 * r;
      const int c = it / 340, px = it - c * 340;
      const int row = px / 34, col = px - row * 34;
      if (c < 3) Ip[c * WS_IPLANE + row * WS_ICOLS + col] = vi[r];
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
  if (band_lo + local < band_hi) {
    issue_g(band_lo + local, 0);
    issue_img(band_lo + local);
  }
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      store_g();
      if (h == 0) store_img();
      __syncthreads();
      if (h == 0) issue_g(t, 1);
      else if (t + nlocal < band_hi) {
        issue_g(t + nlocal, 0);
        issue_img(t + nlocal);
      }
      // this wave's K-step: row 4 h + wave of the tile = row `wave` of the staged half
      const int r = 4 * h + wave;
      bf16x8 b[2][3];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = Ip[r * WS_ICOLS + 8 * g + e + ioff[nt]];
          v[e] = ilive[nt] ? x : 0.f;
        }
        wg_split3(v, b[nt][0], b[nt][1], b[nt][2]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        bf16x8 a[3];
#pragma unroll
        for (int P = 0; P < 3; ++P) a[P] = *reinterpret_cast<const bf16x8*>(Gp + P * WS_GPIECE + (16 * m + j) * WS_GSTR + (wave * 32 + 8 * g) * 2);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS[q]], b[nt][BS[q]], acc[m][nt], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  f32x4* red = reinterpret_cast<f32x4*>(lds);          // [wave][m][nt][lane]: 32 KB
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) red[((wave * 4 + m) * 2 + nt) * 64 + lane] = acc[m][nt];
  __syncthreads();
  const int Kp = p.ph[0].nchunks * kChunk;
  float* slab = p.ws + (long long)blockIdx.x * geo.slab;
  for (int it = tid; it < 8 * 64; it += 256) {
    const int mn = it >> 6, l = it & 63, m = mn >> 1, nt = mn & 1;
    const f32x4 s = (red[((0 * 4 + m) * 2 + nt) * 64 + l] + red[((1 * 4 + m) * 2 + nt) * 64 + l]) +
                    (red[((2 * 4 + m) * 2 + nt) * 64 + l] + red[((3 * 4 + m) * 2 + nt) * 64 + l]);
    const int jj = l & 15, gg = l >> 4, k = 16 * nt + jj;
    if (k < K) {
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(long long)(16 * m + 4 * gg + e) * Kp + k] = s[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------------ dispatch
static int lds3_wgrad_form(const dn_conv_desc* d, const IgemmParams& p) {     // 0 none, 1 iconv0 (16 -> 16), 2 iconv0 + 1-channel piece, 3 stem
  if (knobs().no_lds3 || norm_compute(d->compute) != DN_COMPUTE_F32X3) return 0;
  if (d->kind != DN_CONV_FWD || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return 0;
  if (d->IH != d->OH || d->IW != d->OW) return 0;
  if ((long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31) || (reinterpret_cast<uintptr_t>(p.g) & 15)) return 0;
  for (int t = 0; t < 9; ++t)
    if (p.tdy[t] != t / 3 - 1 || p.tdx[t] != t % 3 - 1) return 0;
  const KOperand& a = p.in[0];
  if (!a.small || a.scale != nullptr || a.up != 0) return 0;
  if (p.Ntot == 16 && a.C == 16 && a.vec) {
    if (p.n_in == 1) return 1;
    const KOperand& b = p.in[1];
    if (p.n_in == 2 && b.C == 1 && b.small && b.scale == nullptr) return 2;
    return 0;
  }
  if (p.Ntot == 64 && p.n_in == 1 && a.C >= 1 && a.C <= 3) return 3;
  return 0;
}

static int lds3_wgrad_blocks(const IgemmParams& p) {
  const int ntiles = p.N * ((p.GH + WG_TH - 1) / WG_TH) * ((p.GW + WG_TW - 1) / WG_TW);
  int blocks = ntiles < 512 ? ntiles : 512;
  return (blocks + 7) / 8 * 8;
}

bool lds3_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) { return lds3_wgrad_form(d, p) != 0; }

size_t lds3_wgrad_workspace_bytes(const IgemmParams& p) {
  return (size_t)lds3_wgrad_blocks(p) * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
}

int launch_lds3_wgrad(const dn_conv_desc* d, IgemmParams& p, float* dw, hipStream_t stream) {
  const int form = lds3_wgrad_form(d, p);
  WgGeo geo;
  geo.tilesX = (p.GW + WG_TW - 1) / WG_TW;
  geo.tilesY = (p.GH + WG_TH - 1) / WG_TH;
  geo.ntiles = p.N * geo.tilesX * geo.tilesY;
  geo.per_xcd = (geo.ntiles + 7) / 8;
  geo.slab = (long long)p.Npad * p.ph[0].nchunks * kChunk;
  geo.dbg = knobs().lds3_dbg ? reinterpret_cast<long long*>(knobs().wino_dbgptr) : nullptr;
  geo.dbgmode = knobs().lds3_dbg;
  const int blocks = lds3_wgrad_blocks(p);
  hipError_t e = hipSuccess;
  if (geo.dbg != nullptr && form == 2) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3_wgrad16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WG16_LDS);
    if (e == hipSuccess) DN_LAUNCH((lds3_wgrad16_kernel<true, true>), dim3(blocks), dim3(256), (size_t)WG16_LDS, stream, p, geo);
    set_last_kernel("dn::lds3_wgrad16_kernel<true>");
  } else if (form == 3) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3_wgrad_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WGS_LDS);
    if (e == hipSuccess) DN_LAUNCH(lds3_wgrad_stem_kernel, dim3(blocks), dim3(256), (size_t)WGS_LDS, stream, p, geo);
    set_last_kernel("dn::lds3_wgrad_stem_kernel");
  } else if (form == 2) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3_wgrad16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WG16_LDS);
    if (e == hipSuccess) DN_LAUNCH(lds3_wgrad16_kernel<true>, dim3(blocks), dim3(256), (size_t)WG16_LDS, stream, p, geo);
    set_last_kernel("dn::lds3_wgrad16_kernel<true>");
  } else if (form == 1) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3_wgrad16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WG16_LDS);
    if (e == hipSuccess) DN_LAUNCH(lds3_wgrad16_kernel<false>, dim3(blocks), dim3(256), (size_t)WG16_LDS, stream, p, geo);
    set_last_kernel("dn::lds3_wgrad16_kernel<false>");
  } else {
    set_error("launch_lds3_wgrad: no form");
    return DN_ERR_UNSUPPORTED;
  }
  if (e != hipSuccess) {
    set_error("lds3 wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  int rc = check_launch("lds3_wgrad_kernel");
  if (rc != DN_OK) return rc;
  p.splits = blocks;                                 // wgrad_reduce_kernel: fixed-order sum of the block slabs + scatter into dw's layout
  return launch_wgrad_reduce(p, dw, stream);
}

}  // namespace dn

// ---- iconv1 form (second half of round 4): 96 input channels in one or two NHWC operands (+ the 1-channel nearest-x2 piece), <= 32 output
// channels.  55 accumulator tiles (9 taps x 6 sixteen-channel groups + the 1-channel piece) x 2 M tiles do not fit one wave: ONE block of
// eight waves per CU, each wave owning whole (channel group, tap row) UNITS of three tap columns -- its own output columns for ALL pixels,
// so no wave ever meets another.  18 units over 8 waves as (3, 3, 2, 2, 2, 2, 2, 2) puts 5 / 5 / 4 / 4 units on the four SIMDs.  A tile is
// 2 rows x 32 pixels (input planes of 4 rows x 96 channels x three pieces = 113 KB of LDS); staging, funnel-shifted taps and the
// fixed-order slab fold are the 16-channel form's.
namespace dn {

constexpr int WK_TH = 2;
constexpr int WK_GSTR = WK_TH * 32 * 2 + 16;      // 144: bytes between the dy planes of two output channels
constexpr int WK_GPIECE = 32 * WK_GSTR;
constexpr int WK_XROWS = WK_TH + 2;
constexpr int WK_XSTR = WK_XROWS * WG_XROWB + 16; // 400: bytes between the planes of two input channels
constexpr int WK_NCI = 96;
constexpr int WK_XPIECE = WK_NCI * WK_XSTR;
constexpr int WK_LDS = 3 * WK_GPIECE + 3 * WK_XPIECE + WK_XROWS * WG_DCOLS * 4;

template <bool HAS1, bool DBG = false>
__global__ void __launch_bounds__(512, 2) lds3k_wgrad_kernel(const IgemmParams p, const WgGeo geo, int c0) {
  extern __shared__ __align__(16) char lds[];
  char* Gp = lds;
  char* Xp = lds + 3 * WK_GPIECE;
  float* Dp = reinterpret_cast<float*>(lds + 3 * WK_GPIECE + 3 * WK_XPIECE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KOperand& S0 = p.in[0];
  const bool two = p.n_in - (HAS1 ? 1 : 0) == 2;
  const KOperand& S1 = p.in[two ? 1 : 0];
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S0.p), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S1.p), 0, 0x80000000u, 0x00020000);
  const KOperand& SD = p.in[HAS1 ? p.n_in - 1 : 0];
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SD.p), 0, 0x80000000u, 0x00020000);

  // units of this wave: u0 = wave, u1 = wave + 8, u2 = wave + 16 (waves 0 and 1 only); unit = tap row * 6 + channel group.  (The extra
  // staging jobs and the 1-channel piece are waves 2 and 3's, the ones with two units.)
  const int nunits = wave < 2 ? 3 : 2;
  f32x4 acc[3][3][2];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[u][x][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // the 1-channel piece's tile: wave 2
  const int dtap = j < 9 ? j : 0;
  const int doff = (dtap / 3) * WG_DCOLS + (dtap % 3);

  f32x4 va[8], vb[8];
  float dv = 0.f;
  // staging jobs: x job = (row 0..3, column group 0..5, channel quad 0..23) -> 576: round A = jobs tid, round B = jobs 512 + tid (tid < 64);
  // dy job = (row 0..1, column group 0..3, channel quad 0..7) -> 64: round B, threads 64..127.  The jobs of the first operand come first
  // (24 x c0 / 4 of them, a whole number of waves: lds3k_wgrad_form), quad fastest within an operand: every wave reads ONE operand through
  // ONE descriptor (a per-lane choice needs two loads with complementary predicates and an add that waits for both: 8 k of the tile's 21 k
  // ticks went there, tools/lds3_timing.py) and a pixel's channels stay contiguous across its lanes.
  // Within an operand a job index reads (rc / 4, quad / 4, quad % 4, rc % 4) from the top: sixteen consecutive lanes write four channel
  // quads x four neighbouring 16-byte columns, which are sixteen different bank groups (planes 400 bytes apart: the quad moves the
  // column by 4 of 16, rc by one); quad-fastest lanes collided four deep (3.2-3.7 k ticks of staging against 2.1 k, tools/lds3_timing.py).
  const int nq0 = two ? c0 >> 2 : 24, n0jobs = 24 * nq0;
  auto decode = [&](int job, int& quad, int& rc, bool& second) __attribute__((always_inline)) {
    second = job >= n0jobs;
    const int jj = second ? job - n0jobs : job, nqh = (second ? 24 - nq0 : nq0) >> 2;
    const int rc_lo = jj & 3, q_lo = (jj >> 2) & 3, hi = jj >> 4;
    const int rc_hi = hi / nqh, q_hi = hi - rc_hi * nqh;
    rc = 4 * rc_hi + rc_lo;
    quad = (second ? nq0 : 0) + 4 * q_hi + q_lo;
  };
  int quadA, rcA, quadB, rcB;
  bool secA, secB;
  decode(tid, quadA, rcA, secA);
  decode(512 + (tid & 63), quadB, rcB, secB);
  // A job is eight pixel loads of one float4 through one (wave-uniform) descriptor: byte offset of pixel 0, byte step, 8 validity bits.
  // Job A = this thread's x job; job B = the 64 remaining x jobs (wave 2) and the 64 dy jobs (wave 3).  The offsets of the NEXT tile are
  // prepared once per tile and the sixteen loads are issued ONE AT A TIME between the groups of the matrix loop (load_site): the texture
  // path takes a tile's 82 KB in ~2-3 k cycles, and a wave that issues its loads in one piece stands in that queue with its matrix
  // instructions behind it (tools/lds3_timing.py: 3 k of 14.8 k ticks per tile, whether ahead of the loop or at the top of a K-step).
  const bool secAu = __builtin_amdgcn_readfirstlane((int)secA) != 0, secBu = __builtin_amdgcn_readfirstlane((int)secB) != 0;
  const int waveu = __builtin_amdgcn_readfirstlane(wave);
  const bool hasB = waveu == 2 || waveu == 3;
  const __amdgpu_buffer_rsrc_t rA = secAu ? r1 : r0;
  const __amdgpu_buffer_rsrc_t rB = waveu == 3 ? rg : (secBu ? r1 : r0);
  int offA = 0, stepA = 0, offB = 0, stepB = 0, offD = -1;
  unsigned maskA = 0, maskB = 0;
  auto x_prep = [&](int quad, int rc, bool second, int gy0, int gx0, int n, int& off, int& step, unsigned& mask) __attribute__((always_inline)) {
    const int cgp = rc % 6, row = rc / 6;
    const int iy = gy0 - 1 + row, ix0 = gx0 + 8 * cgp - 8;
    const int sw = second ? (int)S1.sw : (int)S0.sw;
    const int base = second ? (n * (int)S1.sn + iy * (int)S1.sh + 4 * quad - c0) : (n * (int)S0.sn + iy * (int)S0.sh + 4 * quad);
    off = (base + ix0 * sw) * 4;
    step = sw * 4;
    unsigned m = 0;
    if ((unsigned)iy < (unsigned)p.IH) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if ((unsigned)(ix0 + i) < (unsigned)p.IW) m |= 1u << i;
      m &= cgp == 0 ? 0x80u : (cgp == 5 ? 0x01u : 0xffu);             // (the halo groups: one column each)
    }
    mask = m;
  };
  auto prep_loads = [&](int t, bool live) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, q1 = t / geo.tilesX;
    const int tyb = q1 % geo.tilesY, n = q1 / geo.tilesY;
    const int gy0 = tyb * WK_TH, gx0 = txb * WG_TW;
    x_prep(quadA, rcA, secA, gy0, gx0, n, offA, stepA, maskA);
    if (waveu == 2) {
      x_prep(quadB, rcB, secB, gy0, gx0, n, offB, stepB, maskB);
    } else if (waveu == 3) {
      const int cgp = lane & 3, cq = ((lane >> 2) & 3) + 4 * ((lane >> 4) & 1), row = lane >> 5;      // (bank groups: as the x jobs)
      const int gy = gy0 + row, gx = gx0 + 8 * cgp;
      offB = (((n * p.GH + gy) * p.GW + gx) * p.Ntot + 4 * cq) * 4;
      stepB = p.Ntot * 4;
      unsigned m = 0;
      if (gy < p.GH && 4 * cq < p.Ntot) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (gx + i < p.GW) m |= 1u << i;
      }
      maskB = m;
    }
    if constexpr (HAS1) {
      const int row = tid / 34, col = tid - row * 34;
      const int iy = gy0 - 1 + row, ix = gx0 - 1 + col;
      const bool ok = tid < WK_XROWS * 34 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      offD = ok ? (n * (int)SD.sn + (iy >> SD.up) * (int)SD.sh + (ix >> SD.up) * (int)SD.sw) * 4 : -1;
    }
    if (!live) { maskA = 0; maskB = 0; offD = -1; }
    if constexpr (DBG) {
      if (geo.dbgmode == 2) { maskA = 0; maskB = 0; }
      if (geo.dbgmode == 3) { offA &= 0xfff0; offB &= 0xfff0; stepA = 16; stepB = 16; }
    }
  };
  auto load_site = [&](int i) __attribute__((always_inline)) {          // pixel i of both jobs (i is a compile-time constant at every site)
    va[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (maskA >> i) & 1 ? offA + i * stepA : -1, 0, 0));
    if (hasB) vb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, (maskB >> i) & 1 ? offB + i * stepB : -1, 0, 0));
  };
  auto load_d = [&]() __attribute__((always_inline)) {
    if constexpr (HAS1) dv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, offD, 0, 0));
  };
  auto store_x = [&](int quad, int rc, const f32x4 (&v)[8]) __attribute__((always_inline)) {
    const int cgp = rc % 6, row = rc / 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f[8] = {v[0][e], v[1][e], v[2][e], v[3][e], v[4][e], v[5][e], v[6][e], v[7][e]};
      bf16x8 h, m, l;
      wg_split3(f, h, m, l);
      char* dst = Xp + (4 * quad + e) * WK_XSTR + row * WG_XROWB + cgp * 16;
      *reinterpret_cast<bf16x8*>(dst) = h;
      *reinterpret_cast<bf16x8*>(dst + WK_XPIECE) = m;
      *reinterpret_cast<bf16x8*>(dst + 2 * WK_XPIECE) = l;
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
    store_x(quadA, rcA, va);
    if (wave == 2) {
      store_x(quadB, rcB, vb);
    } else if (wave == 3) {
      const int cq = ((lane >> 2) & 3) + 4 * ((lane >> 4) & 1), pxg = (lane & 3) + 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f[8] = {vb[0][e], vb[1][e], vb[2][e], vb[3][e], vb[4][e], vb[5][e], vb[6][e], vb[7][e]};
        bf16x8 h, m, l;
        wg_split3(f, h, m, l);
        char* dst = Gp + (4 * cq + e) * WK_GSTR + pxg * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + WK_GPIECE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * WK_GPIECE) = l;
      }
    }
    if constexpr (HAS1) {
      const int row = tid / 34, col = tid - row * 34;
      if (tid < WK_XROWS * 34) Dp[row * WG_DCOLS + col] = dv;
    }
  };

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
  long long tk[6] = {0, 0, 0, 0, 0, 0}, c0t = 0, c1t = 0;             // DBG: load wait | split + LDS writes | barrier | load issue | matrix loop | barrier
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr (DBG) { c1t = clock64(); tk[k] += c1t - c0t; c0t = c1t; }
  };
  if (band_lo + local < band_hi) {
    prep_loads(band_lo + local, true);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_site(i);
    load_d();
  }
  if constexpr (DBG) c0t = clock64();
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(0); }
    store_lds();
    stamp(1);
    __syncthreads();
    stamp(2);
    const bool more = t + nlocal < band_hi;
    if (more) {
      prep_loads(t + nlocal, true);
#pragma unroll
      for (int i = 0; i < 8; ++i) load_site(i);
      load_d();
    }
    stamp(3);
#pragma unroll
    for (int r = 0; r < WK_TH; ++r) {                                  // K-step: tile row r, pixels 8 g .. 8 g + 7 per lane group
      bf16x8 a[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int P = 0; P < 3; ++P) a[m][P] = *reinterpret_cast<const bf16x8*>(Gp + P * WK_GPIECE + (16 * m + j) * WK_GSTR + (r * 32 + 8 * g) * 2);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < nunits) {
          const int unit = wave + 8 * u, c16 = unit % 6, ty = unit / 6;
          const char* rowb = Xp + (16 * c16 + j) * WK_XSTR + (r + ty) * WG_XROWB + (8 * g + 8) * 2;
          u32x4 c[3];
          unsigned lf[3], rt[3];
#pragma unroll
          for (int P = 0; P < 3; ++P) {
            c[P] = *reinterpret_cast<const u32x4*>(rowb + P * WK_XPIECE);
            lf[P] = *reinterpret_cast<const unsigned*>(rowb + P * WK_XPIECE - 4);
            rt[P] = *reinterpret_cast<const unsigned*>(rowb + P * WK_XPIECE + 16);
          }
          bf16x8 bm[3], bz[3], bp[3];
#pragma unroll
          for (int P = 0; P < 3; ++P) {
            const u32x4 m4 = u32x4{__builtin_amdgcn_alignbit(c[P][0], lf[P], 16), __builtin_amdgcn_alignbit(c[P][1], c[P][0], 16),
                                   __builtin_amdgcn_alignbit(c[P][2], c[P][1], 16), __builtin_amdgcn_alignbit(c[P][3], c[P][2], 16)};
            const u32x4 p4 = u32x4{__builtin_amdgcn_alignbit(c[P][1], c[P][0], 16), __builtin_amdgcn_alignbit(c[P][2], c[P][1], 16),
                                   __builtin_amdgcn_alignbit(c[P][3], c[P][2], 16), __builtin_amdgcn_alignbit(rt[P], c[P][3], 16)};
            bm[P] = __builtin_bit_cast(bf16x8, m4);
            bz[P] = __builtin_bit_cast(bf16x8, c[P]);
            bp[P] = __builtin_bit_cast(bf16x8, p4);
          }
#pragma unroll
          for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              acc[u][0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], bm[BS[q]], acc[u][0][m], 0, 0, 0);
              acc[u][1][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], bz[BS[q]], acc[u][1][m], 0, 0, 0);
              acc[u][2][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], bp[BS[q]], acc[u][2][m], 0, 0, 0);
            }
          }
        }
      }
      if constexpr (HAS1) {
        if (wave == 2) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = Dp[r * WG_DCOLS + 8 * g + e + doff];
            v[e] = j < 9 ? x : 0.f;
          }
          bf16x8 b[3];
          wg_split3(v, b[0], b[1], b[2]);
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < 2; ++m) accd[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][AS[q]], b[BS[q]], accd[m], 0, 0, 0);
        }
      }
    }
    stamp(4);
    __syncthreads();
    stamp(5);
  }
  if constexpr (DBG) {
    if (lane == 0) {
      long long* o = geo.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 6; ++k) o[k] = tk[k];
      o[6] = (band_hi - band_lo - local + nlocal - 1) / nlocal;
    }
  }

  // ---- every wave owns its columns: straight into the block's slab ws[block][co][k]
  const int Kp = p.ph[0].nchunks * kChunk;
  const int C1 = WK_NCI - c0;
  const int kb1 = ((9 * c0 + kChunk - 1) / kChunk) * kChunk;
  const int kbs = two ? kb1 + ((9 * C1 + kChunk - 1) / kChunk) * kChunk : ((9 * WK_NCI + kChunk - 1) / kChunk) * kChunk;
  float* slab = p.ws + (long long)blockIdx.x * geo.slab;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (u < nunits) {
      const int unit = wave + 8 * u, c16 = unit % 6, ty = unit / 6;
      const int ci = 16 * c16 + j;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const int tap = 3 * ty + x;
        const int k = (two && ci >= c0) ? kb1 + tap * C1 + (ci - c0) : tap * (two ? c0 : WK_NCI) + ci;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = 16 * m + 4 * g + e;
            if (co < p.Ntot) slab[(long long)co * Kp + k] = acc[u][x][m][e];
          }
      }
    }
  }
  if constexpr (HAS1) {
    if (wave == 2 && j < 9) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = 16 * m + 4 * g + e;
          if (co < p.Ntot) slab[(long long)co * Kp + kbs + j] = accd[m][e];
        }
    }
  }
}

// 0 none, 1 without / 2 with the trailing 1-channel piece
static int lds3k_wgrad_form(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_lds3 || norm_compute(d->compute) != DN_COMPUTE_F32X3) return 0;
  if (d->kind != DN_CONV_FWD || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return 0;
  if (d->IH != d->OH || d->IW != d->OW || p.Ntot > 32 || p.Ntot < 17 || (p.Ntot & 3)) return 0;
  if ((long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31)) return 0;
  for (int t = 0; t < 9; ++t)
    if (p.tdy[t] != t / 3 - 1 || p.tdx[t] != t % 3 - 1) return 0;
  int nmain = p.n_in;
  bool has1 = false;
  if (p.n_in >= 2 && p.in[p.n_in - 1].C == 1) {
    const KOperand& b = p.in[p.n_in - 1];
    if (!(b.small && b.scale == nullptr)) return 0;
    has1 = true;
    nmain = p.n_in - 1;
  }
  if (nmain < 1 || nmain > 2) return 0;
  int ctot = 0;
  for (int i = 0; i < nmain; ++i) {
    const KOperand& a = p.in[i];
    if (!(a.vec && a.small && a.up == 0 && a.scale == nullptr && a.C % 4 == 0)) return 0;
    ctot += a.C;
  }
  if (ctot != WK_NCI) return 0;
  if (nmain == 2 && (p.in[0].C & 31)) return 0;             // the first operand's staging jobs fill whole waves (24 x C / 4 a multiple of 64)
  const int ntiles = p.N * ((p.GH + WK_TH - 1) / WK_TH) * ((p.GW + WG_TW - 1) / WG_TW);
  if (ntiles < 192) return 0;
  return has1 ? 2 : 1;
}

static int lds3k_wgrad_blocks(const IgemmParams& p) {
  const int ntiles = p.N * ((p.GH + WK_TH - 1) / WK_TH) * ((p.GW + WG_TW - 1) / WG_TW);
  int blocks = ntiles < 256 ? ntiles : 256;
  return (blocks + 7) / 8 * 8;
}

bool lds3k_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) { return lds3k_wgrad_form(d, p) != 0; }

size_t lds3k_wgrad_workspace_bytes(const IgemmParams& p) {
  return (size_t)lds3k_wgrad_blocks(p) * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
}

int launch_lds3k_wgrad(const dn_conv_desc* d, IgemmParams& p, float* dw, hipStream_t stream) {
  const int form = lds3k_wgrad_form(d, p);
  WgGeo geo;
  geo.tilesX = (p.GW + WG_TW - 1) / WG_TW;
  geo.tilesY = (p.GH + WK_TH - 1) / WK_TH;
  geo.ntiles = p.N * geo.tilesX * geo.tilesY;
  geo.per_xcd = (geo.ntiles + 7) / 8;
  geo.slab = (long long)p.Npad * p.ph[0].nchunks * kChunk;
  const int blocks = lds3k_wgrad_blocks(p);
  const int c0 = p.in[0].C;
  hipError_t e = hipSuccess;
  geo.dbg = knobs().lds3_dbg ? reinterpret_cast<long long*>(knobs().wino_dbgptr) : nullptr;
  geo.dbgmode = knobs().lds3_dbg;
  if (geo.dbg != nullptr && form == 2) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3k_wgrad_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS);
    if (e == hipSuccess) DN_LAUNCH((lds3k_wgrad_kernel<true, true>), dim3(blocks), dim3(512), (size_t)WK_LDS, stream, p, geo, c0);
    set_last_kernel("dn::lds3k_wgrad_kernel<true>");
  } else if (form == 2) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3k_wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS);
    if (e == hipSuccess) DN_LAUNCH(lds3k_wgrad_kernel<true>, dim3(blocks), dim3(512), (size_t)WK_LDS, stream, p, geo, c0);
    set_last_kernel("dn::lds3k_wgrad_kernel<true>");
  } else if (form == 1) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds3k_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS);
    if (e == hipSuccess) DN_LAUNCH(lds3k_wgrad_kernel<false>, dim3(blocks), dim3(512), (size_t)WK_LDS, stream, p, geo, c0);
    set_last_kernel("dn::lds3k_wgrad_kernel<false>");
  } else {
    set_error("launch_lds3k_wgrad: no form");
    return DN_ERR_UNSUPPORTED;
  }
  if (e != hipSuccess) {
    set_error("lds3k wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  int rc = check_launch("lds3k_wgrad_kernel");
  if (rc != DN_OK) return rc;
  p.splits = blocks;
  return launch_wgrad_reduce(p, dw, stream);
}

}  // namespace dn
