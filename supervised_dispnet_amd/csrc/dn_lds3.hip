// LDS-resident direct convolutions for the thin full-resolution layers of the decoder (round 4; reference models/Disp_vgg_BN.py:101-110,
// 177-186: upconv0 = ConvTranspose2d(32, 16, 4, 2, 1) + LeakyReLU, iconv0 = Conv2d(16 + 1, 16, 3, 1, 1) + LeakyReLU, and their input
// gradients).  These layers move 160-220 MB for 7-8 GFLOP: their floor is HBM (20-27 us), yet the tiled kernels re-fetched every 3x3 tap of
// a 128-pixel row tile through L1 / L2 (1.6-4.5x the algorithmic bytes) and ran the fp32 matrix instruction, which alone needs 44-53 us.
//
// Here a block owns a 2-D tile of the output grid (TH x TW grid points of one image):
//   * its input tile + halo is read from global memory ONCE, split ONCE into the three exact bf16 pieces of DN_COMPUTE_F32X3 (DESIGN.md
//     section 3: x = x0 + x1 + x2 exactly) and kept in LDS as [piece][8-channel group][row][col] planes of 16-byte units, so the
//     operand of any tap is one conflict-free ds_read_b128 per lane (stride-2 gathers: columns de-interleaved by parity);
//   * the product is D = W . X on v_mfma_f32_16x16x32_bf16 with the OUTPUT CHANNELS as the M dimension: the weights of a wave's 16
//     output channels -- three pieces, <= 8 K-steps -- live in registers for the whole kernel (persistent blocks), the pixels are the N
//     dimension, and the C/D layout hands every lane four consecutive output channels of ONE pixel: bias + activation + one float4
//     store per lane, whole 64-byte pixels per quarter wave, no transposition through LDS;
//   * six partial products per K-step (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0: fp32-level accuracy, same scheme as the Winograd kernels);
//   * a trailing 1-channel piece (the nearest-x2 up-sampled disparity of torch.cat((upconv0, disp1up), 1)) rides in the two spare
//     8-wide K slots of the last K-step: its nine taps are gathered from a small fp32 plane and split in registers;
//   * waves take ROLES: one role (16 output channels), two (17..32 output channels: M tiles 0 / 1 on wave pairs), or four (the four
//     sub-pixel phases of a stride-2 transposed convolution, one per wave, all fed from the same input tile).
// The XCD a block runs on gets a contiguous band of tiles, so the halo rows two neighbouring tiles share are fetched by one L2.
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lds3_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

// four values at once: ONE (wave-uniform) branch on the activation instead of one per element
__device__ __forceinline__ f32x4 lds3_act4(f32x4 v, int act, float p0, float p1) {
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

// x = h + m + l exactly (round to bf16, subtract, round, subtract: the last residual has <= 8 significant bits)
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
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

struct Lds3Geo {
  int tilesX, tilesY, ntiles;      // tiles per image along x / y, total (N * tilesY * tilesX)
  int dy0, dx0;                    // smallest tap offsets over all phases: the input tile starts at grid * stride + (dy0, dx0)
  int per_xcd;                     // tiles per XCD band (ceil(ntiles / 8))
  long long* dbg;                  // DN_LDS3_DBG (tools/lds3_timing.py): per-wave phase ticks, 8 per wave; nullptr otherwise
  int dbgmode;                     // its value: 2 = no prefetch loads, 3 = no result stores (timing ablations, wrong results)
};

// CG: 8-channel groups of the main operand (2: 16 channels, 4: 32); NKS: K-steps of 32 per role; HAS1: trailing 1-channel operand;
// ROLES: 1 / 2 (M tiles) / 4 (phases); STRIDE: input step per grid point; TH x TW grid points per tile; ROWS x COLS input tile.
template <int CG, int NKS, bool HAS1, int ROLES, int STRIDE, int TH, int TW, int ROWS, int COLS>
struct Lds3Cfg {
  static constexpr int HALFC = (COLS + 1) / 2;
  static constexpr int COLSP = STRIDE == 2 ? 2 * HALFC : COLS;
  static constexpr int PLANE = ROWS * COLSP * 16;                                   // one (piece, channel group) plane, bytes
  // staging writes: a 16-lane group covers 16 / CG pixels x CG groups -> group planes 256 / CG bytes apart (mod 256) are conflict free
  static constexpr int CGSTRIDE = ((PLANE + 255) / 256) * 256 + 256 / CG;
  static constexpr int PSTRIDE = CG * CGSTRIDE;                                     // one piece
  static constexpr int DPLANE = HAS1 ? ROWS * COLS * 4 : 0;                         // fp32 plane of the 1-channel operand
  static constexpr size_t LDS = (size_t)3 * PSTRIDE + DPLANE;
  static constexpr int PTB = TH * TW / 16;                                          // 16-pixel tiles per block tile
  static constexpr int WAVES_PER_ROLE = 4 / ROLES;
  static constexpr int PT_PER_WAVE = PTB / WAVES_PER_ROLE;
  static_assert(TW % 16 == 0 && PT_PER_WAVE % 2 == 0, "pixel tiles are processed in pairs");
};

template <int CG, int NKS, bool HAS1, int ROLES, int STRIDE, int TH, int TW, int ROWS, int COLS, int PIPE>       // PIPE 2: 1 + phase timestamps
__global__ void __launch_bounds__(256, 2) lds3_conv_kernel(const IgemmParams p, const Lds3Geo geo) {
  using Cfg = Lds3Cfg<CG, NKS, HAS1, ROLES, STRIDE, TH, TW, ROWS, COLS>;
  constexpr int C = 8 * CG;
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int role = ROLES == 1 ? 0 : (ROLES == 2 ? (wave >> 1) : wave);
  const int sub = ROLES == 1 ? wave : (ROLES == 2 ? (wave & 1) : 0);
  const int z = ROLES == 4 ? role : 0;                 // phase
  const int mt = ROLES == 2 ? role : 0;                // M tile (16 output channels)
  const KPhase ph = p.ph[z];
  const int ntaps = ph.ntaps, Kp = ph.nchunks * kChunk;
  const int nslots = ntaps * CG;                       // 8-wide K slots of the main operand
  const int kbase1 = ((ntaps * C + kChunk - 1) / kChunk) * kChunk;

  // ---- this wave's weights, as A fragments in registers: row = output channel 16 * mt + j, K slot 4 * ks + g
  bf16x8 wa[NKS][3];
  int boff[NKS];                                       // LDS byte offset of the lane's K slot relative to its pixel (piece 0)
  {
    const int n = 16 * mt + j;
    const float* wrow = p.w + ph.w_off + (long long)n * Kp;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int s = 4 * ks + g;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      int off = 0;
      if (s < nslots) {
        const int tap = s / CG, cg = s - tap * CG;
        if (n < p.Ntot) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + tap * C + 8 * cg), b = *reinterpret_cast<const f32x4*>(wrow + tap * C + 8 * cg + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        }
        const int ey = (int)p.tdy[ph.tap0 + tap] - geo.dy0, ex = (int)p.tdx[ph.tap0 + tap] - geo.dx0;
        const int cterm = STRIDE == 2 ? ((ex & 1) * Cfg::HALFC + (ex >> 1)) : ex;
        off = cg * Cfg::CGSTRIDE + (ey * Cfg::COLSP + cterm) * 16;
      } else if (HAS1 && n < p.Ntot) {
        const int t0 = (s - nslots) * 8;                // taps t0 .. t0 + 7 of the 1-channel operand (k = kbase1 + tap)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (s - nslots < 2 && t0 + e < ntaps) v[e] = wrow[kbase1 + t0 + e];
      }
      boff[ks] = off;
      split3(v, wa[ks][0], wa[ks][1], wa[ks][2]);
    }
  }
  // the 1-channel operand's eight taps of this lane's slot (lanes g >= 2 of the last K-step): offsets into the fp32 plane, in floats
  int doff[8];
  if constexpr (HAS1) {
    const int t0 = (g & 1) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = t0 + e < ntaps ? t0 + e : 0;
      doff[e] = ((int)p.tdy[ph.tap0 + t] - geo.dy0) * COLS + ((int)p.tdx[ph.tap0 + t] - geo.dx0);
    }
  }
  // bias of this lane's four output channels
  const int n0 = 16 * mt + 4 * g;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (p.bias != nullptr && n0 + e < p.Ntot) ? p.bias[n0 + e] : 0.f;
  // result segment of channel n0 (a lane's four channels never straddle two segments on the float4 path: checked per store)
  int seg = 0;
  if (p.n_out > 1 && n0 >= p.out[1].n_begin) seg = 1;
  if (p.n_out > 2 && n0 >= p.out[2].n_begin) seg = 2;
  const KResult R = p.out[seg];
  const int cl = n0 - R.n_begin;
  const bool vec_store = (n0 + 3 < p.Ntot) && (cl + 4 <= R.C) && ((R.sw & 3) == 0) && ((R.sh & 3) == 0) && ((R.sn & 3) == 0) && ((cl & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(R.p) & 15) == 0);

  const KOperand& S = p.in[0];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc1 = rsrc;
  if constexpr (HAS1) rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in[1].p), 0, 0x80000000u, 0x00020000);

  // XCD-aware tile walk: blocks b, b + 8, ... share an XCD (observed placement; only speed depends on it) and walk one contiguous band
  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  // The global loads of tile t + 1 are issued before tile t is computed and land in registers under its matrix instructions; split +
  // LDS writes follow the compute phase (one barrier each side).  Without the prefetch a block sat through a full memory latency per tile.
  // The staging items of a thread never change: their tile-relative global offset, (row, column) for the bounds test and LDS address are
  // computed once (per tile that leaves one add, two compares and a select per load; re-deriving them cost as many vector instructions
  // as the split itself: tools/lds3_timing.py, 1.4 k of a tile's 12.7 k ticks in the load issue alone).
  constexpr int ITEMS = ROWS * COLS * CG, ROUNDS = (ITEMS + 255) / 256;
  f32x4 va[ROUNDS], vb[ROUNDS];
  float dv[2] = {0.f, 0.f};
  int goff[ROUNDS], rowcol[ROUNDS], ldst[ROUNDS];          // rowcol: row << 16 | column, or -1 for a dead slot
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int it = tid + 256 * r;
    const int cg = it % CG, px = it / CG;
    const int row = px / COLS, col = px - row * COLS;
    const int idx = STRIDE == 2 ? ((col & 1) * Cfg::HALFC + (col >> 1)) : col;
    goff[r] = (row * (int)S.sh + col * (int)S.sw + 8 * cg) * 4;
    rowcol[r] = it < ITEMS ? (row << 16 | col) : -1;
    ldst[r] = cg * Cfg::CGSTRIDE + (row * Cfg::COLSP + idx) * 16;
  }
  int rowcol1[2] = {-1, -1};
  int sh1 = 0, up1 = 0;
  if constexpr (HAS1) {
    const KOperand& S1 = p.in[1];
    sh1 = (int)S1.sh;
    up1 = S1.up;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int it = tid + 256 * r;
      const int row = it / COLS, col = it - row * COLS;
      rowcol1[r] = it < ROWS * COLS ? (row << 16 | col) : -1;
    }
  }
  auto issue_loads = [&](int t) __attribute__((always_inline)) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int iy0 = tyb * TH * STRIDE + geo.dy0, ix0 = txb * TW * STRIDE + geo.dx0;
    const int tbase = (n * (int)S.sn + iy0 * (int)S.sh + ix0 * (int)S.sw) * 4;                // (scalar)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int iy = iy0 + (rowcol[r] >> 16), ix = ix0 + (rowcol[r] & 0xffff);
      const bool ok = rowcol[r] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int off = tbase + goff[r];
      va[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : -1, 0, 0));
      vb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off + 16 : -1, 0, 0));
    }
    if constexpr (HAS1) {
      const KOperand& S1 = p.in[1];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int iy = iy0 + (rowcol1[r] >> 16), ix = ix0 + (rowcol1[r] & 0xffff);
        const bool ok = rowcol1[r] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int off = (n * (int)S1.sn + (iy >> up1) * sh1 + (ix >> up1) * (int)S1.sw) * 4;
        dv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc1, ok ? off : -1, 0, 0));
      }
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (rowcol[r] >= 0) {
        const float v[8] = {va[r][0], va[r][1], va[r][2], va[r][3], vb[r][0], vb[r][1], vb[r][2], vb[r][3]};
        bf16x8 h, m, l;
        split3(v, h, m, l);
        char* dst = lds + ldst[r];
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + Cfg::PSTRIDE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * Cfg::PSTRIDE) = l;
      }
    }
    if constexpr (HAS1) {
      float* dpl = reinterpret_cast<float*>(lds + 3 * Cfg::PSTRIDE);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (tid + 256 * r < ROWS * COLS) dpl[tid + 256 * r] = dv[r];
    }
  };
  constexpr bool DBG = PIPE == 2;
  long long tk[6] = {0, 0, 0, 0, 0, 0}, c0t = 0, c1t = 0;             // DBG: load wait | split + LDS writes | barrier | load issue | matrix + stores | barrier
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr (DBG) { c1t = clock64(); tk[k] += c1t - c0t; c0t = c1t; }
  };
  // Everything the epilogue reads that came through a vector-memory load (the per-lane result descriptor, the bias) is CONSUMED here,
  // ahead of the loop: the compiler waits for a loaded register at its first use, the counter is in order, and a first use inside the
  // loop put an s_waitcnt vmcnt(0) between the issue of the next tile's loads and this tile's matrix instructions -- the prefetch never
  // ran under the compute (found in the ISA, round 4; tools/lds3_timing.py).
  asm volatile("" ::"v"(R.p), "v"(R.sn), "v"(R.sh), "v"(R.sw), "v"(R.accumulate), "v"(bias[0]), "v"(bias[1]), "v"(bias[2]), "v"(bias[3]));
  // Fast result path (ONE plain result tensor, float4-addressable, below 2 GB -- every layer of the metric's nets): a raw buffer store per
  // 16-pixel tile whose out-of-range lanes carry offset -1 (dropped by the hardware), the tile's scalar base + a per-lane constant as
  // the address.  The general path below walks the result segments per element under exec masks: ~150 instructions and ten branches
  // per pixel tile, 2.5 k of a tile's 9 k matrix-phase ticks (tools/lds3_timing.py).
  const KResult& R0 = p.out[0];
  const bool fast_out = p.n_out == 1 && !R0.accumulate && R0.n_begin == 0 && (p.Ntot & 3) == 0 && ((R0.sw | R0.sh | R0.sn) & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(R0.p) & 15) == 0 && (long long)p.N * R0.sn * 4 < (1ll << 31);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(R0.p, 0, 0x80000000u, 0x00020000);
  const int lane_out = (j * p.osx * (int)R0.sw + n0) * 4;
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
    if (t + nlocal < band_hi && !(DBG && geo.dbgmode == 2)) issue_loads(t + nlocal);
    stamp(3);

    // ---- the wave's pixel tiles, two at a time (independent accumulators)
#pragma unroll 1
    for (int i = 0; i < Cfg::PT_PER_WAVE; i += 2) {
      f32x4 acc[2];
      int lbase[2], dbase[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = sub + Cfg::WAVES_PER_ROLE * (i + u);
        const int ty = pt / (TW / 16), tx16 = pt - ty * (TW / 16);
        lbase[u] = ((ty * STRIDE) * Cfg::COLSP + tx16 * 16 + j) * 16;
        dbase[u] = ty * COLS + tx16 * 16 + j;
        acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // B fragments of K-step ks for both pixel tiles (the 1-channel piece: gathered + split for the lanes whose slot it is)
      auto load_b = [&](int ks, bf16x8 (&b)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int P = 0; P < 3; ++P) b[u][P] = *reinterpret_cast<const bf16x8*>(lds + P * Cfg::PSTRIDE + lbase[u] + boff[ks]);
        if constexpr (HAS1) {
          if (ks == NKS - 1 && !(DBG && geo.dbgmode == 4)) {
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
              split3(v, h, m, l);
              if (4 * ks + g >= nslots) { b[u][0] = h; b[u][1] = m; b[u][2] = l; }
            }
          }
        }
      };
      // x . w partial products, smallest first: (w0 x2) (w0 x1) (w1 x1) (w0 x0) (w1 x0) (w2 x0)
      constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
      if constexpr (PIPE) {
        // software pipeline: the LDS reads of K-step ks + 1 are ISSUED before the matrix instructions of K-step ks (the compiler's own
        // order waited for each K-step's reads right in front of its first matrix instruction)
        bf16x8 bq[2][2][3];
        load_b(0, bq[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks + 1 < NKS) load_b(ks + 1, bq[(ks + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][AS[q]], bq[ks & 1][u][BS[q]], acc[u], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          bf16x8 b[2][3];
          load_b(ks, b);
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][AS[q]], b[u][BS[q]], acc[u], 0, 0, 0);
        }
      }
      // ---- epilogue: lane (j, g) holds output channels n0 .. n0 + 3 of grid point (gy, gx0 + 16 * tx16 + j)
      if (fast_out) {
        const int tbase = (n * (int)R0.sn + (gy0 * p.osy + ph.ooy) * (int)R0.sh + (gx0 * p.osx + ph.oox) * (int)R0.sw) * 4;      // (scalar)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pt = sub + Cfg::WAVES_PER_ROLE * (i + u);
          const int ty = pt / (TW / 16), tx16 = pt - ty * (TW / 16);
          const int gy = gy0 + ty, gx = gx0 + tx16 * 16 + j;
          const int oy = gy * p.osy + ph.ooy, ox = gx * p.osx + ph.oox;
          const bool ok = gy < p.GH && gx < p.GW && oy < p.OH && ox < p.OW && n0 < p.Ntot && !(DBG && geo.dbgmode == 3 && acc[u][0] != 12345.678f);
          const f32x4 w4 = lds3_act4(acc[u] + f32x4{bias[0], bias[1], bias[2], bias[3]}, p.act, p.act_p0, p.act_p1);
          const int off = tbase + (ty * p.osy * (int)R0.sh + tx16 * 16 * p.osx * (int)R0.sw) * 4 + lane_out;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rout, ok ? off : -1, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = sub + Cfg::WAVES_PER_ROLE * (i + u);
        const int ty = pt / (TW / 16), tx16 = pt - ty * (TW / 16);
        const int gy = gy0 + ty, gx = gx0 + tx16 * 16 + j;
        const int oy = gy * p.osy + ph.ooy, ox = gx * p.osx + ph.oox;
        if (gy < p.GH && gx < p.GW && oy < p.OH && ox < p.OW && n0 < p.Ntot && !(DBG && geo.dbgmode == 3 && acc[u][0] != 12345.678f)) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = lds3_act(acc[u][e] + bias[e], p.act, p.act_p0, p.act_p1);
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
      }
    }
    stamp(4);
    __syncthreads();                                   // the next tile's staging overwrites the planes
    stamp(5);
  }
  if constexpr (DBG) {
    if (lane == 0) {
      long long* o = geo.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 6; ++k) o[k] = tk[k];
      o[6] = (band_hi - band_lo - local + nlocal - 1) / nlocal;
    }
  }
}

// ------------------------------------------------------------------------------------------------------ first layer (stem)
// conv3x3 of the <= 3-channel image (the user's NCHW tensor through its strides) to 64 channels + BatchNorm partial statistics
// (torchvision vgg16_bn features[0..1]; reference models/Disp_vgg_BN.py:84,137).  436 MB of output for 5.9 GFLOP: a store-bound kernel.
// Same scheme as above with K = 9 * C <= 27 padded to ONE 32-wide K-step: the 18 x 34 x C input tile sits in LDS as fp32 planes, a lane
// gathers the eight K values of its slot for a 16-pixel tile (8 ds_read_b32), splits them in registers, and 4 M tiles x 6 partial products
// produce 64 output channels of 16 pixels; lane (j, g) then holds channels 16 m + 4 g .. + 3 of pixel j: four float4 stores.  A wave owns
// 2 rows x 32 columns = 64 pixels; two waves make one 128-pixel row of the BatchNorm partial-statistics table (sum, M2 about the tile
// mean; dn_bn_finalize): taken in one pass about a pivot (the wave's first pixel), reduced over the 16 pixel lanes with row-rotate DPP
// adds, the two halves merged through LDS (Chan).  8 x 32 tiles: 6656 of them at 32 x 128 x 416 = 13 per resident block, no tail.
template <int ROT>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROT, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row, in every lane
  v += dpp_row_ror<8>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<1>(v);
  return v;
}

constexpr int STEM_TH = 8, STEM_TW = 32, STEM_ROWS = 10, STEM_COLS = 34, STEM_COLSP = 36, STEM_PLANE = STEM_ROWS * STEM_COLSP;

__global__ void __launch_bounds__(256, 2) stem3_conv_kernel(const IgemmParams p, const Lds3Geo geo) {
  __shared__ float pl[3 * STEM_PLANE];
  __shared__ __align__(16) float wstat[4][64][2];       // per wave: (sum, M2 about the wave's mean) of its 64 pixels, per channel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const KOperand& S = p.in[0];
  const int C = S.C, K = 9 * C;
  const KPhase ph = p.ph[0];
  const int Kp = ph.nchunks * kChunk;

  bf16x8 wa[4][3];
  int goff[8];
  bool glive[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e;
    glive[e] = k < K;
    const int tap = glive[e] ? k / C : 0, c = glive[e] ? k - tap * C : 0;
    goff[e] = c * STEM_PLANE + ((int)p.tdy[tap] - geo.dy0) * STEM_COLSP + ((int)p.tdx[tap] - geo.dx0);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float* wrow = p.w + ph.w_off + (long long)(16 * m + j) * Kp;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = glive[e] ? wrow[8 * g + e] : 0.f;
    split3(v, wa[m][0], wa[m][1], wa[m][2]);
  }
  float bias[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[m][e] = p.bias != nullptr ? p.bias[16 * m + 4 * g + e] : 0.f;
  const KResult R = p.out[0];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);

  const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, nlocal = (int)gridDim.x >> 3;
  const int band_lo = xcd * geo.per_xcd, band_hi = min(band_lo + geo.per_xcd, geo.ntiles);
  constexpr int PIX = STEM_ROWS * STEM_COLS, ROUNDS = (3 * PIX + 255) / 256;
  float stg[ROUNDS];
  auto issue_loads = [&](int t) __attribute__((always_inline)) {       // the image tile of tile t (zero halo) -> registers
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int iy0 = tyb * STEM_TH + geo.dy0, ix0 = txb * STEM_TW + geo.dx0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = tid + 256 * r;
      const int c = it / PIX, px = it - c * PIX;
      const int row = px / STEM_COLS, col = px - row * STEM_COLS;
      const int iy = iy0 + row, ix = ix0 + col;
      const bool ok = c < C && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const int off = (n * (int)S.sn + iy * (int)S.sh + ix * (int)S.sw + c * (int)S.sc) * 4;
      stg[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? off : -1, 0, 0));
    }
  };
  if (band_lo + local < band_hi) issue_loads(band_lo + local);
  for (int t = band_lo + local; t < band_hi; t += nlocal) {
    const int txb = t % geo.tilesX, r1 = t / geo.tilesX;
    const int tyb = r1 % geo.tilesY, n = r1 / geo.tilesY;
    const int gy0 = tyb * STEM_TH, gx0 = txb * STEM_TW;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = tid + 256 * r;
      const int c = it / PIX, px = it - c * PIX;
      const int row = px / STEM_COLS, col = px - row * STEM_COLS;
      if (c < 3) pl[c * STEM_PLANE + row * STEM_COLSP + col] = stg[r];
    }
    __syncthreads();
    if (t + nlocal < band_hi) issue_loads(t + nlocal);      // lands under this tile's matrix instructions

    float sv[4][4], qv[4][4], pv[4][4];
#pragma unroll 1
    for (int i = 0; i < 4; i += 2) {                        // the wave's rows 2 * wave, 2 * wave + 1: two 16-pixel tiles each
      f32x4 acc[2][4];
      bf16x8 b[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = i + u;
        const int ty = 2 * wave + (pt >> 1), tx16 = pt & 1;
        const int base = ty * STEM_COLSP + tx16 * 16 + j;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = pl[base + goff[e]];
          v[e] = glive[e] ? x : 0.f;
        }
        split3(v, b[u][0], b[u][1], b[u][2]);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][AS[q]], b[u][BS[q]], acc[u][m], 0, 0, 0);
      if (p.bn_partial != nullptr) {
        if (i == 0) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
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
          for (int m = 0; m < 4; ++m)
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
        const int gy = gy0 + 2 * wave + (pt >> 1), gx = gx0 + (pt & 1) * 16 + j;
        float* o = R.p + (long long)n * R.sn + (long long)gy * R.sh + (long long)gx * R.sw + 4 * g;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          f32x4 w4;
#pragma unroll
          for (int e = 0; e < 4; ++e) w4[e] = lds3_act(acc[u][m][e] + bias[m][e], p.act, p.act_p0, p.act_p1);
          *reinterpret_cast<f32x4*>(o + 16 * m) = w4;
        }
      }
    }
    if (p.bn_partial != nullptr) {
      // (sum, M2 about the mean) of the wave's 64 pixels per channel from the pivot-centred sums: S = s + 64 pv, M2 = q - s^2 / 64
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float s1 = row16_sum(sv[m][e]), q1 = row16_sum(qv[m][e]);
          if (j == 0) *reinterpret_cast<f32x2*>(&wstat[wave][16 * m + 4 * g + e][0]) = f32x2{fmaf(64.f, pv[m][e], s1), q1 - s1 * s1 * (1.f / 64.f)};
        }
    }
    __syncthreads();
    if (p.bn_partial != nullptr && tid < 128) {
      // waves (0, 1) and (2, 3) make one 128-pixel row of the statistics table each (Chan's merge of two 64-pixel halves); the next
      // tile's wstat writes come after the next barrier, so these reads cannot race them
      const int h = tid >> 6, c = tid & 63;
      const f32x2 a = *reinterpret_cast<const f32x2*>(&wstat[2 * h][c][0]), b2 = *reinterpret_cast<const f32x2*>(&wstat[2 * h + 1][c][0]);
      const float dm = (b2[0] - a[0]) * (1.f / 64.f);
      *reinterpret_cast<f32x2*>(p.bn_partial + ((long long)(2 * t + h) * p.Ntot + c) * 2) = f32x2{a[0] + b2[0], a[1] + b2[1] + dm * dm * 32.f};
    }
  }
}

static bool stem3_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_lds3 || p.compute != DN_COMPUTE_F32X3) return false;
  if (d->kind != DN_CONV_FWD || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return false;
  if (p.n_in != 1 || p.n_out != 1 || p.Ntot != 64 || p.nphases != 1) return false;
  const KOperand& o = p.in[0];
  const KResult& r = p.out[0];
  if (!(o.C >= 1 && o.C <= 3 && o.up == 0 && o.scale == nullptr && o.small)) return false;
  if (!(r.linear && !r.accumulate && (r.sw & 3) == 0 && (reinterpret_cast<uintptr_t>(r.p) & 15) == 0)) return false;
  return p.GH % STEM_TH == 0 && p.GW % STEM_TW == 0;          // full tiles: two waves' 128 pixels are one row of the statistics table
}

bool stem3_conv_eligible(const dn_conv_desc* d, const IgemmParams& p) { return stem3_eligible(d, p); }

int launch_stem3_conv(const IgemmParams& p, hipStream_t stream) {
  Lds3Geo geo;
  geo.tilesX = p.GW / STEM_TW;
  geo.tilesY = p.GH / STEM_TH;
  geo.ntiles = p.N * geo.tilesX * geo.tilesY;
  geo.dy0 = -1;
  geo.dx0 = -1;
  geo.per_xcd = (geo.ntiles + 7) / 8;
  int blocks = geo.ntiles < 512 ? geo.ntiles : 512;
  blocks = (blocks + 7) / 8 * 8;
  DN_LAUNCH(stem3_conv_kernel, dim3(blocks), dim3(256), 0, stream, p, geo);
  set_last_kernel("dn::stem3_conv_kernel");
  return check_launch("stem3_conv_kernel");
}

// ------------------------------------------------------------------------------------------------------ configurations
struct Lds3Pick {
  int cfg;          // 0: none
  Lds3Geo geo;
};

static Lds3Pick lds3_pick(const dn_conv_desc* d, const IgemmParams& p) {
  Lds3Pick r;
  r.cfg = 0;
  if (knobs().no_lds3 || p.compute != DN_COMPUTE_F32X3) return r;             // three-piece arithmetic only (DN_COMPUTE=f32 keeps the fp32 instruction)
  if (p.reflect || p.bn_partial != nullptr || p.bnb_y != nullptr || p.Ntot > 32 || p.n_in > 2 || p.n_out < 1) return r;
  if (d->dilation > 1) return r;
  if (!(p.nphases == 1 || (p.nphases == 4 && p.Ntot <= 16))) return r;
  const KOperand& a = p.in[0];
  if (!(a.vec && a.small && a.up == 0 && a.scale == nullptr && (a.C == 16 || a.C == 32))) return r;
  const bool has1 = p.n_in == 2;
  if (has1) {
    const KOperand& b = p.in[1];
    if (!(b.C == 1 && b.small && b.scale == nullptr)) return r;
  }
  if (p.sy != p.sx || (p.sy != 1 && p.sy != 2)) return r;
  if (has1 && (p.sy != 1 || p.nphases != 1 || p.Ntot > 16)) return r;
  for (int i = 0; i < p.n_out; ++i)
    if (p.out[i].C < 1) return r;
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
  const int cg = a.C / 8, slots = maxtaps * cg + (has1 ? 2 : 0);
  if (has1 && maxtaps > 16) return r;
  const int roles = p.nphases == 4 ? 4 : (p.Ntot > 16 ? 2 : 1);
  // the compiled forms (CG, NKS, HAS1, ROLES, STRIDE, TH, TW, ROWS, COLS)
  int cfg = 0, TH = 8, TW = 32;
  if (cg == 2 && has1 && roles == 1 && p.sy == 1 && maxtaps == 9 && spanY == 3 && spanX == 3) cfg = 1;            // iconv0 forward
  else if (cg == 2 && !has1 && roles == 2 && p.sy == 1 && slots <= 20 && spanY == 3 && spanX == 3) cfg = 2;      // iconv0 input gradient
  else if (cg == 2 && !has1 && roles == 1 && p.sy == 1 && slots <= 20 && spanY == 3 && spanX == 3) cfg = 5;      // 16 -> <= 16, 3x3
  else if (cg == 4 && !has1 && roles == 4 && p.sy == 1 && slots <= 16 && spanY == 3 && spanX == 3) cfg = 3;      // upconv0 forward
  else if (cg == 2 && !has1 && roles == 2 && p.sy == 2 && slots <= 32 && spanY == 4 && spanX == 4) { cfg = 4; TH = 4; }   // upconv0 input gradient
  if (!cfg) return r;
  r.cfg = cfg;
  r.geo.tilesX = (p.GW + TW - 1) / TW;
  r.geo.tilesY = (p.GH + TH - 1) / TH;
  r.geo.ntiles = p.N * r.geo.tilesX * r.geo.tilesY;
  r.geo.dy0 = dy0;
  r.geo.dx0 = dx0;
  r.geo.per_xcd = (r.geo.ntiles + 7) / 8;
  r.geo.dbg = knobs().lds3_dbg ? reinterpret_cast<long long*>(knobs().wino_dbgptr) : nullptr;
  r.geo.dbgmode = knobs().lds3_dbg;
  return r;
}

bool lds3_conv_eligible(const dn_conv_desc* d, const IgemmParams& p) { return lds3_pick(d, p).cfg != 0; }

template <int CG, int NKS, bool HAS1, int ROLES, int STRIDE, int TH, int TW, int ROWS, int COLS>
static int lds3_launch(const IgemmParams& p, const Lds3Geo& geo, hipStream_t stream) {
  using Cfg = Lds3Cfg<CG, NKS, HAS1, ROLES, STRIDE, TH, TW, ROWS, COLS>;
  auto kernel = lds3_conv_kernel<CG, NKS, HAS1, ROLES, STRIDE, TH, TW, ROWS, COLS, 1>;
  if constexpr (CG == 2 && HAS1) {                                  // phase timestamps of the iconv0 forward form (tools/lds3_timing.py)
    if (geo.dbg != nullptr) kernel = lds3_conv_kernel<CG, NKS, HAS1, ROLES, STRIDE, TH, TW, ROWS, COLS, 2>;
  }
  const size_t lds = Cfg::LDS;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS %zu): %s", lds, hipGetErrorString(e));
      return DN_ERR_LAUNCH;
    }
  }
  int blocks = geo.ntiles < 512 ? geo.ntiles : 512;            // two resident blocks per CU, persistent over the tiles
  if (geo.dbg != nullptr && geo.dbgmode == 5) blocks = 256;     // (timing: one block per CU)
  blocks = (blocks + 7) / 8 * 8;
  DN_LAUNCH(kernel, dim3(blocks), dim3(256), lds, stream, p, geo);
  set_last_kernel("dn::lds3_conv_kernel<%d, %d, %s, %d, %d, %d, %d, %d, %d, %s>", CG, NKS, HAS1 ? "true" : "false", ROLES, STRIDE, TH, TW, ROWS, COLS,
                  "true");
  return check_launch("lds3_conv_kernel");
}

int launch_lds3_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream) {
  const Lds3Pick k = lds3_pick(d, p);
  switch (k.cfg) {
    case 1: return lds3_launch<2, 5, true, 1, 1, 8, 32, 10, 34>(p, k.geo, stream);
    case 2: return lds3_launch<2, 5, false, 2, 1, 8, 32, 10, 34>(p, k.geo, stream);
    case 5: return lds3_launch<2, 5, false, 1, 1, 8, 32, 10, 34>(p, k.geo, stream);
    case 3: return lds3_launch<4, 4, false, 4, 1, 8, 32, 10, 34>(p, k.geo, stream);
    case 4: return lds3_launch<2, 8, false, 2, 2, 4, 32, 10, 66>(p, k.geo, stream);
    default: set_error("launch_lds3_conv: no configuration"); return DN_ERR_UNSUPPORTED;
  }
}

}  // namespace dn
