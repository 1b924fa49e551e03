// Implicit-GEMM convolution family on the CDNA4 fp32 matrix core (v_mfma_f32_32x32x2_f32), gfx950 only.
//
//   forward / input-gradient / conv-transpose :  igemm_conv_kernel   D[pixel][cout] = sum_k A[pixel][k] * W[cout][k]
//   weight gradient                          :  igemm_wgrad_kernel  D[cout][k]     = sum_pixels G[pixel][cout] * A[pixel][k]
//
// A is never materialised: the loader walks (operand piece, tap, channel) and gathers straight from the NHWC
// activations (virtual concat, on-the-fly nearest x2 upsample, fused BatchNorm-apply + ReLU of the producer), zero-fills
// the halo, and stages 32-wide K chunks through LDS (register-staged, double-buffered, one barrier per chunk).
// LDS rows are padded to 36 floats so the per-lane ds_read_b128 fragment reads are conflict-free; the K index inside a
// group of 8 is permuted between the two half-waves (lanes<32 take k=0..3, lanes>=32 take k=4..7) so one b128 read feeds
// four MFMAs.  Accumulation is an exact fp32 FMA chain (no reduced precision anywhere).
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "dn_internal.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DN_STAGGER
#define DN_STAGGER 0
#endif
constexpr int LDK = 36;  // padded LDS row (floats) of a [rows][32] K-chunk tile

__device__ __forceinline__ float apply_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

// floor(n/d), n < 2^31, with magic M = floor(2^32/d) (0xFFFFFFFF for d == 1): estimate is exact or one low; branch-free fix-up
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned d, unsigned M, unsigned* rem) {
  unsigned q = __umulhi(n, M);
  unsigned r = n - q * d;
  const bool fix = r >= d;
  q += fix ? 1u : 0u;
  r -= fix ? d : 0u;
  *rem = r;
  return q;
}

// Two blocks share a CU, i.e. two waves share each SIMD's matrix pipe.  Launched together they run IN PHASE (both in their MFMA
// phase, then both staging: pipe idle ~25 % -- measured SQ_VALU_MFMA_BUSY 66 %).  A static priority asymmetry between the
// two wave slots of a SIMD breaks the symmetry: the favoured wave keeps the pipe whenever it wants it, the other fills the
// gaps while the first one stages.  HW_REG_HW_ID (id 4) bits [3:0] = wave slot within the SIMD.
__device__ __forceinline__ void stagger_priority() {
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned slot = (lin >> 8) & 1u;
#if DN_STAGGER == 2
  if (slot & 1u) {
    __builtin_amdgcn_s_sleep(64);   // ~4k cycles: start the odd slot half a chunk late
  }
#else
  if (slot & 1u) __builtin_amdgcn_s_setprio(2);
#endif
}

// ReflectionPad2d index map for v in [-(n-1), 2(n-1)]
__device__ __forceinline__ int reflect_idx(int v, int n) {
  const int m = n - 1;
  int a = v < 0 ? -v : v;
  a = m - a;
  a = a < 0 ? -a : a;
  return m - a;
}

// Which operand piece does K-chunk `kc` of a phase with `ntaps` taps fall in?  Uniform across the block.
__device__ __forceinline__ int select_operand(const IgemmParams& p, int ntaps, int kc, int* kc_local) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < DN_MAX_OPERANDS - 1; ++i) {
    if (s == i && i < p.n_in - 1) {
      int nch = (ntaps * p.in[i].C + kChunk - 1) / kChunk;
      if (kc >= nch) {
        kc -= nch;
        s = i + 1;
      }
    }
  }
  *kc_local = kc;
  return s;
}

// One 4-wide K group of one row of the A operand.
struct AGroup {
  f32x4 v;
  bool ok;     // vector path: halo / tail predicate (value must be zeroed after the deferred affine)
};

// Gathers 4 consecutive K elements [kl, kl+4) of operand S for the pixel context (n, by, bx).
// Vector path: one 16-byte load, affine deferred to the caller (returns raw value + predicate).
// Scalar path: element-wise, fully resolved here (affine applied, zeros filled); ok = true, *defer = false.
__device__ __forceinline__ AGroup gather4(const KOperand& S, int kl, int ntaps, const int* taps, int n, int by, int bx,
                                          bool rowvalid, int IH, int IW, int j_vec, int c_vec, int reflect) {
  AGroup r;
  r.v = f32x4{0.f, 0.f, 0.f, 0.f};
  r.ok = false;
  if (S.vec) {
    if (rowvalid && j_vec < ntaps) {
      int t = taps[j_vec];
      int iy = by + (int)(short)(t & 0xffff), ix = bx + (t >> 16);
      if (reflect) {
        iy = reflect_idx(iy, IH);
        ix = reflect_idx(ix, IW);
      }
      if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) {
        const float* a = S.p + n * S.sn + (long long)(iy >> S.up) * S.sh + (long long)(ix >> S.up) * S.sw + c_vec;
        r.v = *reinterpret_cast<const f32x4*>(a);
        r.ok = true;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int k = kl + e;
      int j = k / S.C, c = k - j * S.C;
      float val = 0.f;
      if (rowvalid && j < ntaps) {
        int t = taps[j];
        int iy = by + (int)(short)(t & 0xffff), ix = bx + (t >> 16);
        if (reflect) {
          iy = reflect_idx(iy, IH);
          ix = reflect_idx(ix, IW);
        }
        if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) {
          val = S.p[n * S.sn + (long long)(iy >> S.up) * S.sh + (long long)(ix >> S.up) * S.sw + (long long)c * S.sc];
          if (S.scale) val = fmaxf(0.f, val * S.scale[c] + S.shift[c]);
        }
      }
      r.v[e] = val;
    }
    r.ok = true;
  }
  return r;
}

// p.tile_store: 0 = never, 1 = dense un-phased results only (the round-2 rule), 2 = every pixel-dense result
// (run_conv requires pixel-dense results; the tile path addresses pixels through rowpix[], which is phase-aware)
__device__ __forceinline__ bool knobs_dev_linear_only(const IgemmParams& p) { return p.tile_store == 1; }

// ---- shared epilogue: bias, activation, channel-split store; optional batch-statistic partials
// C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
template <int BM, int BN, int WM, int WN, bool STORE = true>
__device__ __forceinline__ void conv_epilogue(const IgemmParams& p, f32x16 (&acc)[WM / 32][WN / 32], const int* rowpix, float* As,
                                              int m0, int n0) {
  constexpr int MI = WM / 32, NI = WN / 32, WAVES_N = BN / WN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Fast store path (one dense float4-addressable result, no accumulation): bias + activation into an LDS tile [BM][BN + 4], then
  // every thread stores float4s of consecutive channels -- a wave writes whole pixels (BN*4 contiguous bytes each) instead of
  // 32 x 4 bytes of two pixels per instruction.  The thin decoder layers are bound by their vector-memory instruction count.
  bool tile_store = false;
  if constexpr (STORE) {
    const KResult& R0 = p.out[0];
    // (also when the result ACCUMULATES -- the second writer of a gradient, e.g. the input gradient of a ResNet bottleneck's first 1x1
    //  convolution landing on the residual path's: one float4 read-add-write per thread instead of 16-32 scalar ones.  Config 4's
    //  1x1 input gradients ran at 19-35 TFLOP/s through the scalar path, 2-3x slower than their forward.)
    tile_store = p.n_out == 1 && (R0.linear || !knobs_dev_linear_only(p)) && (R0.sw & 3) == 0 && (p.Ntot & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(R0.p) & 15) == 0 && p.tile_store != 0;
    if (tile_store) {
      constexpr int TLD = BN + 4;
      float* Ts = As;                                  // the staging buffers are free after the main loop's last barrier
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wn * WN + j * 32 + (lane & 31);
        const int n = n0 + col;
        const float bias = (p.bias != nullptr && n < p.Ntot) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wm * WM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            Ts[row * TLD + col] = apply_act(acc[i][j][reg] + bias, p.act, p.act_p0, p.act_p1);
          }
      }
      __syncthreads();
      constexpr int C4 = BN / 4;
      for (int it = tid; it < BM * C4; it += 256) {
        const int row = it / C4, c4 = it - row * C4;
        const int pix = rowpix[row];
        if (pix >= 0 && n0 + 4 * c4 < p.Ntot) {
          f32x4* dst = reinterpret_cast<f32x4*>(R0.p + (long long)pix * R0.sw + n0 + 4 * c4);
          f32x4 v = *reinterpret_cast<const f32x4*>(Ts + row * TLD + 4 * c4);
          if (R0.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ((STORE && !tile_store) ? NI : 0); ++j) {
    const int n = n0 + wn * WN + j * 32 + (lane & 31);
    const bool nvalid = n < p.Ntot;
    int seg = 0;
    if (p.n_out > 1 && n >= p.out[1].n_begin) seg = 1;
    if (p.n_out > 2 && n >= p.out[2].n_begin) seg = 2;
    const KResult& R = p.out[seg];
    float* optr = R.p + (n - R.n_begin);
    const long long sw = R.sw;
    const bool accumulate = R.accumulate != 0;
    const float bias = (p.bias != nullptr && nvalid) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wm * WM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int pix = rowpix[row];
        if (pix >= 0 && nvalid) {
          float v = apply_act(acc[i][j][reg] + bias, p.act, p.act_p0, p.act_p1);
          float* o = optr + (long long)pix * sw;
          if (accumulate) v += *o;
          *o = v;
        }
      }
    }
  }

  if (p.bn_partial != nullptr) {
    // Per-tile batch statistics of the PRE-BIAS accumulators, in the numerically stable form (sum, M2 about the TILE mean):
    // dn_bn_finalize merges the tiles with Chan's parallel-variance update.  E[x^2] - mean^2 on raw sums loses the variance
    // to cancellation whenever |mean| >> std (measured on ResNet-50's 12-values-per-channel layer4).
    float* red = As;                   // [WAVES_M][BN]
    float* tmean = As + (BM / WM) * BN;  // [BN]
    const int nvalid = min(BM, p.M - m0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) s1 += acc[i][j][reg];     // rows past M hold exact zeros
      s1 += __shfl_xor(s1, 32);
      if (lane < 32) red[wm * BN + wn * WN + j * 32 + lane] = s1;
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < BN) {
#pragma unroll
      for (int w = 0; w < BM / WM; ++w) tot += red[w * BN + tid];
      tmean[tid] = tot / (float)nvalid;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float mu = tmean[wn * WN + j * 32 + (lane & 31)];
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = wm * WM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          const float dv = acc[i][j][reg] - mu;
          s2 += (row < nvalid) ? dv * dv : 0.f;
        }
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) red[wm * BN + wn * WN + j * 32 + lane] = s2;
    }
    __syncthreads();
    if (tid < BN) {
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < BM / WM; ++w) m2 += red[w * BN + tid];
      const int n = n0 + tid;
      if (n < p.Ntot) {
        float* dst = p.bn_partial + ((long long)(m0 / BM) * p.Ntot + n) * 2;
        dst[0] = tot;
        dst[1] = m2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward family
// ALLVEC: every operand is float4-addressable with int32 offsets.  Then the whole staging code is straight-line (no
// divergent branches, loads always issued with clamped addresses and masked afterwards), so it shares one basic block with
// the MFMAs and the scheduler can interleave address arithmetic / loads with the 64-cycle matrix instructions.
template <int BM, int BN, int WM, int WN, bool ALLVEC>
__global__ void __launch_bounds__(256, 2) igemm_conv_kernel(const IgemmParams p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                                        // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;                         // [2][BN][LDK]
  int* taps = reinterpret_cast<int*>(Bs + 2 * BN * LDK);   // [kMaxTaps]  (dy | dx<<16)
  int* rowpix = taps + kMaxTaps;                           // [BM] output pixel index or -1

  if (DN_STAGGER) stagger_priority();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const KPhase ph = p.ph[blockIdx.z];
  const int ntaps = ph.ntaps;
  const int nchunks = ph.nchunks;
  const int Kp = nchunks * kChunk;

  if (tid < ntaps) taps[tid] = ((int)p.tdy[ph.tap0 + tid] & 0xffff) | ((int)p.tdx[ph.tap0 + tid] << 16);
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r, pix = -1;
    if (m < p.M) {
      unsigned gx, gy;
      const unsigned t = fastdiv((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
      const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      int oy = (int)gy * p.osy + ph.ooy, ox = (int)gx * p.osx + ph.oox;
      if (oy < p.OH && ox < p.OW) pix = (n * p.OH + oy) * p.OW + ox;
    }
    rowpix[r] = pix;
  }

  // per-thread staging assignment: K group g (4 floats) of rows r0 + 32*i
  const int g = tid & 7, r0 = tid >> 3;
  int rn[AR], rby[AR], rbx[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      unsigned gx, gy;
      const unsigned t = fastdiv((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
      rn[i] = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      rby[i] = (int)gy * p.sy;
      rbx[i] = (int)gx * p.sx;
    } else {
      rn[i] = -1;
      rby[i] = rbx[i] = 0;
    }
  }
  __syncthreads();

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  AGroup av[AR];
  f32x4 bv[BR];
  f32x4 sc4, sh4;
  bool aff = false;
  float relu_floor = 0.f;
  const float* wbase = p.w + ph.w_off;
  int boff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) boff[i] = (n0 + r0 + 32 * i) * Kp + g * 4;

  auto issue_loads = [&](int kc) {
    int kcl;
    const int s = select_operand(p, ntaps, kc, &kcl);
    const KOperand& S = p.in[s];
    const int kl = kcl * kChunk + g * 4;
    if constexpr (ALLVEC) {
      unsigned c;
      const int j = (int)fastdiv((unsigned)kl, (unsigned)S.C, S.mC, &c);
      const bool kvalid = j < ntaps;
      const int t = taps[kvalid ? j : 0];
      const int dy = (int)(short)(t & 0xffff), dx = t >> 16;
      const float* base = S.p;
      const int sn = (int)S.sn, sh = (int)S.sh, sw = (int)S.sw, up = S.up;
      const bool has_aff = S.scale != nullptr;
      // identity affine + floor of -inf when the operand has no pending BN/ReLU: keeps the store stage branch-free
      const f32x4 l1 = *reinterpret_cast<const f32x4*>((has_aff ? S.scale : base) + c);
      const f32x4 l2 = *reinterpret_cast<const f32x4*>((has_aff ? S.shift : base) + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc4[e] = has_aff ? l1[e] : 1.f;
        sh4[e] = has_aff ? l2[e] : 0.f;
      }
      relu_floor = has_aff ? 0.f : -__builtin_huge_valf();
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int iy = rby[i] + dy, ix = rbx[i] + dx;
        if (p.reflect) {
          iy = reflect_idx(iy, p.IH);
          ix = reflect_idx(ix, p.IW);
        }
        const bool ok = kvalid && rn[i] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        int off = rn[i] * sn + (iy >> up) * sh + (ix >> up) * sw + (int)c;
        off = ok ? off : 0;
        av[i].v = *reinterpret_cast<const f32x4*>(base + off);
        av[i].ok = ok;
      }
    } else {
      int j = 0, c = 0;
      aff = false;
      if (S.vec) {
        j = kl / S.C;
        c = kl - j * S.C;
        if (S.scale != nullptr && j < ntaps) {
          sc4 = *reinterpret_cast<const f32x4*>(S.scale + c);
          sh4 = *reinterpret_cast<const f32x4*>(S.shift + c);
          aff = true;
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) av[i] = gather4(S, kl, ntaps, taps, rn[i], rby[i], rbx[i], rn[i] >= 0, p.IH, p.IW, j, c, p.reflect);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) bv[i] = *reinterpret_cast<const f32x4*>(wbase + boff[i] + kc * kChunk);
  };

  auto store_stage = [&](int buf) {
    float* a = As + buf * BM * LDK + r0 * LDK + g * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = av[i].v;
      if constexpr (ALLVEC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = fmaxf(relu_floor, fmaf(v[e], sc4[e], sh4[e]));
          v[e] = av[i].ok ? t : 0.f;
        }
      } else {
        if (aff) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * sc4[e] + sh4[e]);
        }
        if (!av[i].ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(a + 32 * i * LDK) = v;
    }
    float* b = Bs + buf * BN * LDK + r0 * LDK + g * 4;
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(b + 32 * i * LDK) = bv[i];
  };

  if (nchunks > 0) {
    issue_loads(0);
    store_stage(0);
  }
  __syncthreads();

  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    const bool more = (kc + 1 < nchunks);
    if constexpr (ALLVEC) {
      issue_loads(more ? kc + 1 : kc);      // the last iteration re-fetches its own chunk into the idle buffer: no branch
      // keep the loads ABOVE the matrix work: hipcc otherwise sinks them below the MFMAs (shorter live ranges) and the
      // store stage then eats the full memory latency right before the barrier
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (more) issue_loads(kc + 1);
    }
    const float* Ab = As + buf * BM * LDK + (wm * WM + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* Bb = Bs + buf * BN * LDK + (wn * WN + (lane & 31)) * LDK + (lane >> 5) * 4;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + kg * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + kg * 8);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    }
    if constexpr (ALLVEC) {
      __builtin_amdgcn_sched_barrier(0);
      store_stage(buf ^ 1);
    } else {
      if (more) store_stage(buf ^ 1);
    }
    __syncthreads();
  }

  conv_epilogue<BM, BN, WM, WN>(p, acc, rowpix, As, m0, n0);
}

// compile-time loop: the body is instantiated once per index, so register arrays indexed by it never become dynamic
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------------------------- forward family, uniform fast path
// uni32 plans (every operand: C % 32 == 0, float4-addressable, no upsample, < 2 GiB; <= 32 taps; zero padding): a K chunk
// never straddles a tap or an operand, so (operand, tap, channel base) are BLOCK-UNIFORM per chunk and live on the scalar
// unit.  Per chunk and row the vector side is left with: one add + one select for the address, one bit test of a
// precomputed per-row tap-validity mask, the load, and the deferred BatchNorm-apply + ReLU of the producer.
//
// The main loop is hand-scheduled: the iteration is cut into one "slot" per MFMA (64 cycles of matrix pipe each) and the
// staging work of the NEXT chunk is dealt into the slots in source order -- global loads + address arithmetic under the
// first MFMAs, the fragment reads of the next K group two slots before they are needed, the LDS store stage under the last
// MFMAs -- with a sched_barrier after every slot so hipcc keeps that order (left alone it clusters the MFMAs and runs
// the staging before/after them, i.e. nothing overlaps within a wave).
template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256, 2) igemm_conv_u32_kernel(const IgemmParams p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                                        // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;                         // [2][BN][LDK]
  int* taps = reinterpret_cast<int*>(Bs + 2 * BN * LDK);   // [32]  (dy | dx<<16)
  int* rowpix = taps + 32;                                 // [BM] output pixel index or -1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware tile order.  Hardware places block b on XCD b % 8 (each XCD has its own 4 MiB L2): XCD x takes the CONTIGUOUS
  // range [x*per, (x+1)*per) of logical tiles, enumerated N-tile fastest, so the N tiles that re-read one A row block and
  // the neighbouring row blocks that share its halo rows are resident on the same L2 at the same time.
  const int MT = (p.M + BM - 1) / BM, NT = p.Npad / BN;
  const int per = (MT * NT + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= MT * NT) return;      // grid is rounded up to a multiple of 8 (block-uniform exit)
  const int m0 = (q / NT) * BM, n0 = (q % NT) * BN;
  const KPhase ph = p.ph[blockIdx.z];
  const int ntaps = ph.ntaps;
  const int Kp = ph.nchunks * kChunk;

  if (tid < 32) taps[tid] = tid < ntaps ? (((int)p.tdy[ph.tap0 + tid] & 0xffff) | ((int)p.tdx[ph.tap0 + tid] << 16)) : 0;
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r, pix = -1;
    if (m < p.M) {
      unsigned gx, gy;
      const unsigned t = fastdiv((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
      const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      int oy = (int)gy * p.osy + ph.ooy, ox = (int)gx * p.osx + ph.oox;
      if (oy < p.OH && ox < p.OW) pix = (n * p.OH + oy) * p.OW + ox;
    }
    rowpix[r] = pix;
  }
  __syncthreads();

  // per-thread staging assignment: K group g (4 floats) of rows r0 + 32*i; per row a bit mask of the taps that land inside.
  // The row coordinates are recomputed at each operand set-up instead of being kept live through the main loops.
  const int g = tid & 7, r0 = tid >> 3;
  auto row_coords = [&](int i, int* n, int* by, int* bx) {
    const int m = m0 + r0 + 32 * i;
    unsigned gx, gy;
    const unsigned t = fastdiv(m < p.M ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
    *n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
    *by = (int)gy * p.sy;
    *bx = (int)gx * p.sx;
  };
  unsigned vmask[AR];
  {
    int rn[AR], rby[AR], rbx[AR];
    unsigned inside[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      row_coords(i, &rn[i], &rby[i], &rbx[i]);
      inside[i] = 0u;
    }
    for (int j = 0; j < ntaps; ++j) {
      const int tp = taps[j];
      const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        inside[i] |= ((unsigned)(rby[i] + dy) < (unsigned)p.IH && (unsigned)(rbx[i] + dx) < (unsigned)p.IW) ? (1u << j) : 0u;
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) vmask[i] = (m0 + r0 + 32 * i) < p.M ? inside[i] : 0u;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const char* wrow = reinterpret_cast<const char*>(p.w + ph.w_off);   // advanced by one chunk (128 B) per iteration
  unsigned boffB[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) boffB[i] = (unsigned)(((n0 + r0 + 32 * i) * Kp + g * 4) * 4);
  const int stA = (r0 * LDK + g * 4) * 4;                                                  // staging store offset (bytes)
  const int frA = ((wm * WM + (lane & 31)) * LDK + (lane >> 5) * 4) * 4;                    // fragment read offsets (bytes)
  const int frB = ((wn * WN + (lane & 31)) * LDK + (lane >> 5) * 4) * 4;
  char* AsB = reinterpret_cast<char*>(As);
  char* BsB = reinterpret_cast<char*>(Bs);
  constexpr int ABUF = BM * LDK * 4, BBUF = BN * LDK * 4;
  constexpr int ROWS32 = 32 * LDK * 4;                       // byte distance of 32 tile rows

  // slot schedule (compile-time)
  constexpr int NM = 16 * MI * NI;                 // MFMAs per chunk
  constexpr int PK = NM / 4;                       // MFMAs per K group of 8
  constexpr int NF = MI + NI;                      // fragment reads per K group
  constexpr int F0 = (PK - NF - 2) > 0 ? (PK - NF - 2) : 0;   // first slot (within a K group) of the next group's reads
  constexpr int NS = AR + BR;                      // store-stage items
  constexpr int SSTEP = (NM >= 4 * NS) ? 2 : 1;    // slots between store-stage items
  constexpr int S0 = NM - SSTEP * NS;              // slot of the first store-stage item

  int buf = 0;
  for (int s = 0; s < p.n_in; ++s) {
    const KOperand& S = p.in[s];
    // UNI  : C % 32 == 0 -- a chunk is (one tap, 32 channels): tap and channel base are block-uniform scalars.
    // !UNI : C in {4, 8, 16} -- a chunk is 32/C whole taps: the thread's K group sits in tap j0 + gt at channel ct, both fixed
    //        per thread up to the uniform chunk base j0, so the tap word / validity bit / offset are per-thread VGPR values.
    auto run_operand = [&](auto aff_tag, auto uni_tag) {
      constexpr bool HA = decltype(aff_tag)::value;
      constexpr bool UNI = decltype(uni_tag)::value;
      // ---- operand set-up (block-uniform scalars + per-row base offsets)
      const char* base = reinterpret_cast<const char*>(S.p);
      const char* scp = reinterpret_cast<const char*>(S.scale);
      const char* shp = reinterpret_cast<const char*>(S.shift);
      const int sh = (int)S.sh, sw = (int)S.sw;
      const int cpt = S.C >> 5;                       // UNI: chunks per tap
      const int tpc = UNI ? 1 : 32 / S.C;             // !UNI: taps per chunk
      const int nch = UNI ? ntaps * cpt : (ntaps + tpc - 1) / tpc;
      const int gt = UNI ? 0 : (g * 4) / S.C;         // !UNI: this thread's tap within the chunk ...
      const int ct = UNI ? g * 4 : (g * 4) % S.C;     //       ... and its channel
      unsigned rowoffB[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int rn, rby, rbx;
        row_coords(i, &rn, &rby, &rbx);
        rowoffB[i] = (unsigned)((rn * (int)S.sn + rby * sh + rbx * sw + ct) * 4);
      }

      f32x4 av[AR], bv[BR], sc4, sh4;
      bool aok[AR];
      // cursor = the chunk whose loads are issued next: index cn = (tap j, chunk-in-tap cc); its tap word is fetched from LDS
      // one iteration ahead and kept in a VGPR until decoded, so the scalar unit never waits inside the MFMA stream
      int cn = 0, j = 0, cc = 0;          // UNI: j = tap, cc = chunk within the tap; !UNI: j = first tap of the chunk
      int tapv = taps[UNI ? 0 : gt];
      unsigned soffB = 0, jbit = 0, coffB = 0;
      const char* wcur = wrow;

      auto cursor_decode = [&]() {
        const int tapword = UNI ? __builtin_amdgcn_readfirstlane(tapv) : tapv;
        const int dy = (int)(short)(tapword & 0xffff), dx = tapword >> 16;
        soffB = (unsigned)((dy * sh + dx * sw + cc * 32) * 4);
        const int jt = j + gt;
        jbit = jt < ntaps ? 1u << jt : 0u;
        coffB = (unsigned)((cc * 32 + ct) * 4);
        wcur = wrow;
      };
      auto cursor_advance = [&]() {      // clamps at the last chunk (the final iteration re-fetches it into the idle buffer: no branch)
        const bool more = cn + 1 < nch;
        cn += more ? 1 : 0;
        if constexpr (UNI) {
          const int cc1 = cc + 1;
          const bool wrap = cc1 == cpt;
          cc = more ? (wrap ? 0 : cc1) : cc;
          j = (more && wrap) ? j + 1 : j;
        } else {
          j = more ? j + tpc : j;
        }
        wrow += more ? kChunk * 4 : 0;
        const int jt = j + gt;
        tapv = taps[jt < 32 ? jt : 31];
      };
      auto load_a = [&](int i) {
        aok[i] = (vmask[i] & jbit) != 0u;
        const unsigned off = aok[i] ? rowoffB[i] + soffB : 0u;
        av[i] = *reinterpret_cast<const f32x4*>(base + off);
      };
      auto load_aff = [&]() {
        sc4 = *reinterpret_cast<const f32x4*>(scp + coffB);
        sh4 = *reinterpret_cast<const f32x4*>(shp + coffB);
      };
      auto load_b = [&](int i) { bv[i] = *reinterpret_cast<const f32x4*>(wcur + boffB[i]); };
      auto store_a = [&](int b, int i) {
        f32x4 v = av[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[e];
          if constexpr (HA) t = fmaxf(0.f, fmaf(t, sc4[e], sh4[e]));
          v[e] = aok[i] ? t : 0.f;
        }
        *reinterpret_cast<f32x4*>(AsB + b * ABUF + stA + i * ROWS32) = v;
      };
      auto store_b = [&](int b, int i) { *reinterpret_cast<f32x4*>(BsB + b * BBUF + stA + i * ROWS32) = bv[i]; };

      // pipeline fill for this operand (one exposed memory latency per operand)
      cursor_decode();
#pragma unroll
      for (int i = 0; i < AR; ++i) load_a(i);
      if constexpr (HA) load_aff();
#pragma unroll
      for (int i = 0; i < BR; ++i) load_b(i);
#pragma unroll
      for (int i = 0; i < AR; ++i) store_a(buf, i);
#pragma unroll
      for (int i = 0; i < BR; ++i) store_b(buf, i);
      cursor_advance();
      __syncthreads();

      // (A variant with the store stage mid-iteration, the barrier right after the chunk's last fragment read and the next
      //  chunk's first K group fetched under the last MFMAs measured the same 127-128 TFLOP/s: with two waves per SIMD the
      //  partner wave already covers the LDS latency behind the barrier.  The simpler order is kept.)
      for (int c = 0; c < nch; ++c) {
        cursor_decode();                       // chunk min(c+1, nch-1); uses the tap word fetched during the previous iteration

        const char* Ab = AsB + buf * ABUF + frA;
        const char* Bb = BsB + buf * BBUF + frB;
        f32x4 fa[2][MI], fb[2][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * ROWS32);
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) fb[0][jn] = *reinterpret_cast<const f32x4*>(Bb + jn * ROWS32);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NM>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          constexpr int kg = m / PK, q = m % PK;
          constexpr int kk = q / (MI * NI), ij = q % (MI * NI);
          constexpr int i = ij / NI, jn = ij % NI;
          constexpr int cur = kg & 1, nxt = cur ^ 1;
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][kk], fb[cur][jn][kk], acc[i][jn], 0, 0, 0);
          // ---- side work of this slot
          if constexpr (m < AR) load_a(m);
          if constexpr (HA && m == AR) load_aff();
          if constexpr (m >= AR + (HA ? 1 : 0) && m < AR + (HA ? 1 : 0) + BR) load_b(m - AR - (HA ? 1 : 0));
          if constexpr (kg < 3 && q >= F0 && q < F0 + NF) {
            constexpr int f = q - F0;
            if constexpr (f < MI) fa[nxt][f] = *reinterpret_cast<const f32x4*>(Ab + f * ROWS32 + (kg + 1) * 32);
            else fb[nxt][f - MI] = *reinterpret_cast<const f32x4*>(Bb + (f - MI) * ROWS32 + (kg + 1) * 32);
          }
          if constexpr (m == (PK + 1 > AR + 1 + BR ? PK + 1 : AR + 1 + BR)) cursor_advance();   // after this chunk's loads are issued
          if constexpr (m >= S0 && (m - S0) % SSTEP == 0) {
            constexpr int it = (m - S0) / SSTEP;
            if constexpr (it < AR) store_a(buf ^ 1, it);
            else if constexpr (it < NS) store_b(buf ^ 1, it - AR);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        buf ^= 1;
      }
      wrow += kChunk * 4;        // the cursor stopped on this operand's last chunk; the next operand's weights follow it
    };
    // Anything else (C = 1 disparity piece, the 3-channel NCHW image, upsampled or odd-width operands): plain gather, one
    // barrier per chunk, no overlap.  These operands contribute one or two chunks to layers that are HBM-bound anyway.
    auto run_operand_generic = [&]() {
      const int nch = (ntaps * S.C + kChunk - 1) / kChunk;
      // scalar operands are gathered PIXEL-major: thread = (row tid % BM, K slice tid / BM), so for one K element the lanes
      // of a wave read neighbouring pixels (coalesced for the NCHW image and for 1-channel maps); float4 operands that the
      // scheduled loaders do not take (odd widths, upsampled) keep the K-group-major assignment.
      constexpr int KPT = kChunk / (256 / BM);             // K elements per thread per chunk (pixel-major)
      const int prow = tid % BM, pk0 = (tid / BM) * KPT;
      int pn, pby, pbx;
      {
        const int m = m0 + prow;
        unsigned gx, gy;
        const unsigned t = fastdiv(m < p.M ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
        pn = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
        pby = (int)gy * p.sy;
        pbx = (int)gx * p.sx;
      }
      const bool plive = (m0 + prow) < p.M;
      int rn[AR], rby[AR], rbx[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) row_coords(i, &rn[i], &rby[i], &rbx[i]);
      for (int cl = 0; cl < nch; ++cl) {
        if (!S.vec) {
          float vals[KPT];
#pragma unroll
          for (int e = 0; e < KPT; ++e) {
            const int k = cl * kChunk + pk0 + e;
            unsigned c;
            const int j = (int)fastdiv((unsigned)k, (unsigned)S.C, S.mC, &c);
            float v = 0.f;
            if (j < ntaps) {
              const int tp = taps[j];
              const int iy = pby + (int)(short)(tp & 0xffff), ix = pbx + (tp >> 16);
              if (plive && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
                v = S.p[(long long)pn * S.sn + (long long)(iy >> S.up) * S.sh + (long long)(ix >> S.up) * S.sw + (long long)c * S.sc];
                if (S.scale) v = fmaxf(0.f, v * S.scale[c] + S.shift[c]);
              }
            }
            vals[e] = v;
          }
#pragma unroll
          for (int e = 0; e < KPT; e += 4)
            *reinterpret_cast<f32x4*>(AsB + buf * ABUF + (prow * LDK + pk0 + e) * 4) = f32x4{vals[e], vals[e + 1], vals[e + 2], vals[e + 3]};
        } else {
          const int kl = cl * kChunk + g * 4;
          const int jv = kl / S.C, cv = kl - jv * S.C;
          f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f}, sh4 = f32x4{0.f, 0.f, 0.f, 0.f};
          bool aff = false;
          if (S.scale != nullptr && jv < ntaps) {
            sc4 = *reinterpret_cast<const f32x4*>(S.scale + cv);
            sh4 = *reinterpret_cast<const f32x4*>(S.shift + cv);
            aff = true;
          }
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            AGroup a = gather4(S, kl, ntaps, taps, rn[i], rby[i], rbx[i], (m0 + r0 + 32 * i) < p.M, p.IH, p.IW, jv, cv, 0);
            f32x4 v = a.v;
            if (aff) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * sc4[e] + sh4[e]);
            }
            if (!a.ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(AsB + buf * ABUF + stA + i * ROWS32) = v;
          }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
          *reinterpret_cast<f32x4*>(BsB + buf * BBUF + stA + i * ROWS32) = *reinterpret_cast<const f32x4*>(wrow + boffB[i]);
        wrow += kChunk * 4;
        __syncthreads();
        const char* Ab = AsB + buf * ABUF + frA;
        const char* Bb = BsB + buf * BBUF + frB;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          f32x4 fa[MI], fb[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * ROWS32 + kg * 32);
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) fb[jn] = *reinterpret_cast<const f32x4*>(Bb + jn * ROWS32 + kg * 32);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[jn][kk], acc[i][jn], 0, 0, 0);
        }
        buf ^= 1;      // the next chunk fills the other buffer; the barrier above orders it against this chunk's readers
      }
      __syncthreads();
    };
    if (ntaps == 0) continue;       // empty phase of a strided scatter (e.g. 1x1 stride 2): the result is bias/activation only
    const bool fast = S.vec && S.small && S.up == 0;
    if (fast && S.C % 32 == 0) {
      if (S.scale != nullptr) run_operand(std::true_type{}, std::true_type{});
      else run_operand(std::false_type{}, std::true_type{});
    } else if (fast && (S.C == 4 || S.C == 8 || S.C == 16)) {
      if (S.scale != nullptr) run_operand(std::true_type{}, std::false_type{});
      else run_operand(std::false_type{}, std::false_type{});
    } else {
      run_operand_generic();
    }
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, rowpix, As, m0, n0);
}

// ------------------------------------------------------------------------------ forward family, fp32 products on the bf16 matrix cores
// igemm_conv_u32_kernel with DN_COMPUTE_F32X3 arithmetic: every fp32 operand value is split EXACTLY into three bf16
// pieces and the six partial products of weight <= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (16x the fp32
// instruction's rate).  Unlike the Winograd kernels, where each transformed value has exactly one consumer wave, a tile row here is read
// by two waves, so the split is done ONCE by the staging thread (of A after the deferred BatchNorm-apply + ReLU, and of the packed fp32
// weights -- the pack layout is unchanged) and LDS holds the pieces: [row][3 pieces][32 bf16] = 192 bytes per row, its 16-byte groups
// rotated by (row >> 2) & 3 so that both the 8-byte staging stores and the per-lane ds_read_b128 fragment reads are bank-conflict free.
// A 32-deep chunk is two 16-deep matrix steps; wave tiles are limited to 2 x 32 x 32 (two steps of pieces live in registers).  Operands
// the scheduled loaders do not take (1-channel pieces, the 3-channel image, upsampled maps) keep the fp32 instruction on the same
// accumulators (both instructions share the 32 x 32 C/D layout).
constexpr int X3ROW = 192;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256, 2) igemm_conv_x3_kernel(const IgemmParams p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  static_assert(MI * NI <= 2, "the three-piece variant keeps two steps of operand pieces in registers: wave tiles of at most 2 x 32 x 32");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                                        // [2][BM][X3ROW bytes]: per row three pieces x 32 bf16, 16-byte groups rotated by (row >> 2) & 3
  char* AsB = reinterpret_cast<char*>(smem);
  char* BsB = AsB + 2 * BM * X3ROW;                        // [2][BN][X3ROW bytes]
  int* taps = reinterpret_cast<int*>(BsB + 2 * BN * X3ROW);   // [32]  (dy | dx<<16)
  int* rowpix = taps + 32;                                 // [BM] output pixel index or -1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware tile order.  Hardware places block b on XCD b % 8 (each XCD has its own 4 MiB L2): XCD x takes the CONTIGUOUS
  // range [x*per, (x+1)*per) of logical tiles, enumerated N-tile fastest, so the N tiles that re-read one A row block and
  // the neighbouring row blocks that share its halo rows are resident on the same L2 at the same time.
  const int MT = (p.M + BM - 1) / BM, NT = p.Npad / BN;
  const int per = (MT * NT + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= MT * NT) return;      // grid is rounded up to a multiple of 8 (block-uniform exit)
  const int m0 = (q / NT) * BM, n0 = (q % NT) * BN;
  const KPhase ph = p.ph[blockIdx.z];
  const int ntaps = ph.ntaps;
  const int Kp = ph.nchunks * kChunk;
  // K split of small grids (p.ksplit > 1: one scheduled operand, launch_conv_x3): blockIdx.y = kz takes the chunks [kz * cps, (kz + 1) * cps)
  // of the phase; the partial accumulators meet in a workspace and the block that arrives last sums them in index order and runs the
  // epilogue.  The 4x13 / 8x26 transposed convolutions of the decoder are 8-104 tiles with 64-256 chunks each.
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int kz = ksplit > 1 ? (int)blockIdx.y : 0;

  if (tid < 32) taps[tid] = tid < ntaps ? (((int)p.tdy[ph.tap0 + tid] & 0xffff) | ((int)p.tdx[ph.tap0 + tid] << 16)) : 0;
  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r, pix = -1;
    if (m < p.M) {
      unsigned gx, gy;
      const unsigned t = fastdiv((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
      const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      int oy = (int)gy * p.osy + ph.ooy, ox = (int)gx * p.osx + ph.oox;
      if (oy < p.OH && ox < p.OW) pix = (n * p.OH + oy) * p.OW + ox;
    }
    rowpix[r] = pix;
  }
  __syncthreads();

  // per-thread staging assignment: K group g (4 floats) of rows r0 + 32*i; per row a bit mask of the taps that land inside.
  // The row coordinates are recomputed at each operand set-up instead of being kept live through the main loops.
  const int g = tid & 7, r0 = tid >> 3;
  auto row_coords = [&](int i, int* n, int* by, int* bx) {
    const int m = m0 + r0 + 32 * i;
    unsigned gx, gy;
    const unsigned t = fastdiv(m < p.M ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
    *n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
    *by = (int)gy * p.sy;
    *bx = (int)gx * p.sx;
  };
  unsigned vmask[AR];
  {
    int rn[AR], rby[AR], rbx[AR];
    unsigned inside[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      row_coords(i, &rn[i], &rby[i], &rbx[i]);
      inside[i] = 0u;
    }
    for (int j = 0; j < ntaps; ++j) {
      const int tp = taps[j];
      const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        inside[i] |= ((unsigned)(rby[i] + dy) < (unsigned)p.IH && (unsigned)(rbx[i] + dx) < (unsigned)p.IW) ? (1u << j) : 0u;
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) vmask[i] = (m0 + r0 + 32 * i) < p.M ? inside[i] : 0u;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const char* wrow = reinterpret_cast<const char*>(p.w + ph.w_off);   // advanced by one chunk (128 B) per iteration
  unsigned boffB[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) boffB[i] = (unsigned)(((n0 + r0 + 32 * i) * Kp + g * 4) * 4);
  // fp32 layout of the generic (unscheduled) operands, kept inside the same buffers: [rows][LDK floats]
  const int stA = (r0 * LDK + g * 4) * 4;
  const int frA = ((wm * WM + (lane & 31)) * LDK + (lane >> 5) * 4) * 4;
  const int frB = ((wn * WN + (lane & 31)) * LDK + (lane >> 5) * 4) * 4;
  constexpr int ABUF = BM * X3ROW, BBUF = BN * X3ROW;
  constexpr int ROWS32 = 32 * LDK * 4;                       // byte distance of 32 tile rows (fp32 layout)
  constexpr int XROWS32 = 32 * X3ROW;                        // ... (three-piece layout)
  // three-piece layout: the thread's K group g (k 4g..4g+3) of row r0 + 32 i lands in 16-byte group (4 P + (g >> 1) + q) mod 12, half g & 1,
  // q = (row >> 2) & 3 (the same for every i): rows 4 apart would otherwise share banks (192-byte rows)
  int stX[3];
  {
    const int q = (r0 >> 2) & 3;
#pragma unroll
    for (int P = 0; P < 3; ++P) stX[P] = r0 * X3ROW + ((4 * P + (g >> 1) + q) % 12) * 16 + (g & 1) * 8;
  }
  // fragment read: lane (row lane & 31, k half lane >> 5) takes group 4 P + 2 s + h of step s
  int frX[2][3];
  {
    const int q = ((lane & 31) >> 2) & 3;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int P = 0; P < 3; ++P) frX[st][P] = (lane & 31) * X3ROW + ((4 * P + 2 * st + (lane >> 5) + q) % 12) * 16;
  }
  const int frXA = wm * WM * X3ROW, frXB = wn * WN * X3ROW;

  // slot schedule (compile-time): one chunk = two 16-deep steps of MI * NI * 6 matrix instructions
  constexpr int NM = 12 * MI * NI;                 // matrix instructions per chunk
  constexpr int NLD = AR + BR + 1;                 // load items (rows of A, scale / shift, rows of B)
  constexpr int NS = AR + BR;                      // store-stage items (split + three 8-byte stores each)
  constexpr int S0 = NM > NS ? NM - NS : 0;        // slot of the first store-stage item

  int buf = 0;
  for (int s = 0; s < p.n_in; ++s) {
    const KOperand& S = p.in[s];
    // UNI  : C % 32 == 0 -- a chunk is (one tap, 32 channels): tap and channel base are block-uniform scalars.
    // !UNI : C in {4, 8, 16} -- a chunk is 32/C whole taps: the thread's K group sits in tap j0 + gt at channel ct, both fixed
    //        per thread up to the uniform chunk base j0, so the tap word / validity bit / offset are per-thread VGPR values.
    auto run_operand = [&](auto aff_tag, auto uni_tag) {
      constexpr bool HA = decltype(aff_tag)::value;
      constexpr bool UNI = decltype(uni_tag)::value;
      // ---- operand set-up (block-uniform scalars + per-row base offsets)
      const char* base = reinterpret_cast<const char*>(S.p);
      const char* scp = reinterpret_cast<const char*>(S.scale);
      const char* shp = reinterpret_cast<const char*>(S.shift);
      const int sh = (int)S.sh, sw = (int)S.sw;
      const int cpt = S.C >> 5;                       // UNI: chunks per tap
      const int tpc = UNI ? 1 : 32 / S.C;             // !UNI: taps per chunk
      const int nch = UNI ? ntaps * cpt : (ntaps + tpc - 1) / tpc;
      const int gt = UNI ? 0 : (g * 4) / S.C;         // !UNI: this thread's tap within the chunk ...
      const int ct = UNI ? g * 4 : (g * 4) % S.C;     //       ... and its channel
      unsigned rowoffB[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int rn, rby, rbx;
        row_coords(i, &rn, &rby, &rbx);
        rowoffB[i] = (unsigned)((rn * (int)S.sn + rby * sh + rbx * sw + ct) * 4);
      }

      f32x4 av[AR], bv[BR], sc4, sh4;
      bool aok[AR];
      // cursor = the chunk whose loads are issued next: index cn = (tap j, chunk-in-tap cc); its tap word is fetched from LDS
      // one iteration ahead and kept in a VGPR until decoded, so the scalar unit never waits inside the MFMA stream
      const int cps = (nch + ksplit - 1) / ksplit;
      const int c_lo = kz * cps, c_hi = (c_lo + cps) < nch ? c_lo + cps : nch;      // this block's chunks (all of them without a split)
      if (c_lo >= c_hi) return;
      int cn = c_lo, j = UNI ? c_lo / cpt : c_lo * tpc, cc = UNI ? c_lo % cpt : 0;   // UNI: j = tap, cc = chunk within the tap; !UNI: j = first tap of the chunk
      wrow += (size_t)c_lo * (kChunk * 4);
      int tapv = taps[UNI ? j : (j + gt < 32 ? j + gt : 31)];
      unsigned soffB = 0, jbit = 0, coffB = 0;
      const char* wcur = wrow;

      auto cursor_decode = [&]() {
        const int tapword = UNI ? __builtin_amdgcn_readfirstlane(tapv) : tapv;
        const int dy = (int)(short)(tapword & 0xffff), dx = tapword >> 16;
        soffB = (unsigned)((dy * sh + dx * sw + cc * 32) * 4);
        const int jt = j + gt;
        jbit = jt < ntaps ? 1u << jt : 0u;
        coffB = (unsigned)((cc * 32 + ct) * 4);
        wcur = wrow;
      };
      auto cursor_advance = [&]() {      // clamps at the last chunk (the final iteration re-fetches it into the idle buffer: no branch)
        const bool more = cn + 1 < c_hi;
        cn += more ? 1 : 0;
        if constexpr (UNI) {
          const int cc1 = cc + 1;
          const bool wrap = cc1 == cpt;
          cc = more ? (wrap ? 0 : cc1) : cc;
          j = (more && wrap) ? j + 1 : j;
        } else {
          j = more ? j + tpc : j;
        }
        wrow += more ? kChunk * 4 : 0;
        const int jt = j + gt;
        tapv = taps[jt < 32 ? jt : 31];
      };
      auto load_a = [&](int i) {
        aok[i] = (vmask[i] & jbit) != 0u;
        const unsigned off = aok[i] ? rowoffB[i] + soffB : 0u;
        av[i] = *reinterpret_cast<const f32x4*>(base + off);
      };
      auto load_aff = [&]() {
        sc4 = *reinterpret_cast<const f32x4*>(scp + coffB);
        sh4 = *reinterpret_cast<const f32x4*>(shp + coffB);
      };
      auto load_b = [&](int i) { bv[i] = *reinterpret_cast<const f32x4*>(wcur + boffB[i]); };
      // x = h + m + l exactly (three bf16 pieces: round, subtract, round, subtract); split ONCE here, by the staging
      // thread -- every value is then read by two waves (the tile is 2 x 2 waves) as ready-made matrix operands
      auto split_store = [&](char* dst, const f32x4& v) {
        const bf16x2 h0 = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), h1 = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
        const f32x2 ra = f32x2{v[0], v[1]} - __builtin_convertvector(h0, f32x2), rb = f32x2{v[2], v[3]} - __builtin_convertvector(h1, f32x2);
        const bf16x2 m0 = __builtin_convertvector(ra, bf16x2), m1 = __builtin_convertvector(rb, bf16x2);
        const f32x2 sa = ra - __builtin_convertvector(m0, f32x2), sb = rb - __builtin_convertvector(m1, f32x2);
        const bf16x2 l0 = __builtin_convertvector(sa, bf16x2), l1 = __builtin_convertvector(sb, bf16x2);
        *reinterpret_cast<bf16x4*>(dst + stX[0]) = bf16x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<bf16x4*>(dst + stX[1]) = bf16x4{m0[0], m0[1], m1[0], m1[1]};
        *reinterpret_cast<bf16x4*>(dst + stX[2]) = bf16x4{l0[0], l0[1], l1[0], l1[1]};
      };
      auto store_a = [&](int b, int i) {
        f32x4 v = av[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[e];
          if constexpr (HA) t = fmaxf(0.f, fmaf(t, sc4[e], sh4[e]));
          v[e] = aok[i] ? t : 0.f;
        }
        split_store(AsB + b * ABUF + i * XROWS32, v);
      };
      auto store_b = [&](int b, int i) { split_store(BsB + b * BBUF + i * XROWS32, bv[i]); };

      // pipeline fill for this operand (one exposed memory latency per operand)
      cursor_decode();
#pragma unroll
      for (int i = 0; i < AR; ++i) load_a(i);
      if constexpr (HA) load_aff();
#pragma unroll
      for (int i = 0; i < BR; ++i) load_b(i);
#pragma unroll
      for (int i = 0; i < AR; ++i) store_a(buf, i);
#pragma unroll
      for (int i = 0; i < BR; ++i) store_b(buf, i);
      cursor_advance();
      __syncthreads();

      // (A variant with the store stage mid-iteration, the barrier right after the chunk's last fragment read and the next
      //  chunk's first K group fetched under the last MFMAs measured the same 127-128 TFLOP/s: with two waves per SIMD the
      //  partner wave already covers the LDS latency behind the barrier.  The simpler order is kept.)
      for (int c = c_lo; c < c_hi; ++c) {
        cursor_decode();                       // chunk min(c+1, nch-1); uses the tap word fetched during the previous iteration

        const char* Ab = AsB + buf * ABUF + frXA;
        const char* Bb = BsB + buf * BBUF + frXB;
        bf16x8 pa[2][MI][3], pb[2][NI][3];
        auto read_step = [&](int st) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int P = 0; P < 3; ++P) pa[st][i][P] = *reinterpret_cast<const bf16x8*>(Ab + i * XROWS32 + frX[st][P]);
#pragma unroll
          for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int P = 0; P < 3; ++P) pb[st][jn][P] = *reinterpret_cast<const bf16x8*>(Bb + jn * XROWS32 + frX[st][P]);
        };
        read_step(0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NM>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          constexpr int st = m / (6 * MI * NI), q = m % (6 * MI * NI);
          constexpr int ij = q / 6, t = q % 6, i = ij / NI, jn = ij % NI;
          // x0y2, x0y1, x1y1, x0y0, x1y0, x2y0 (the six partial products of weight <= 2^-16; smallest first)
          constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[st][i][AS[t]], pb[st][jn][BS[t]], acc[i][jn], 0, 0, 0);
          // ---- side work of this slot
          if constexpr (m == 1) read_step(1);                 // the second step's operands, one step ahead
          if constexpr (m < NLD || (NM < NLD && m == NM - 1)) {
            // load items in order: rows of A, scale / shift, rows of B (all of what is left in the last slot of a short chunk)
            constexpr int k0 = m, k1 = (NM < NLD && m == NM - 1) ? NLD : m + 1;
            static_for<k1 - k0>([&](auto kc) __attribute__((always_inline)) {
              constexpr int k = k0 + decltype(kc)::value;
              if constexpr (k < AR) load_a(k);
              else if constexpr (k == AR) { if constexpr (HA) load_aff(); }
              else load_b(k - AR - 1);
            });
          }
          if constexpr (m == (NM < NLD ? NM - 1 : NLD)) cursor_advance();   // after this chunk's loads are issued
          if constexpr (m >= S0 || NM <= NS) {
            constexpr int it0 = NM > NS ? m - S0 : (m * NS) / NM, it1 = NM > NS ? it0 + 1 : ((m + 1) * NS) / NM;
            static_for<it1 - it0>([&](auto kc) __attribute__((always_inline)) {
              constexpr int it = it0 + decltype(kc)::value;
              if constexpr (it < AR) store_a(buf ^ 1, it);
              else store_b(buf ^ 1, it - AR);
            });
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        buf ^= 1;
      }
      wrow += kChunk * 4;        // the cursor stopped on this operand's last chunk; the next operand's weights follow it
    };
    // Anything else (C = 1 disparity piece, the 3-channel NCHW image, upsampled or odd-width operands): plain gather, one
    // barrier per chunk, no overlap.  These operands contribute one or two chunks to layers that are HBM-bound anyway.
    auto run_operand_generic = [&]() {
      const int nch = (ntaps * S.C + kChunk - 1) / kChunk;
      // scalar operands are gathered PIXEL-major: thread = (row tid % BM, K slice tid / BM), so for one K element the lanes
      // of a wave read neighbouring pixels (coalesced for the NCHW image and for 1-channel maps); float4 operands that the
      // scheduled loaders do not take (odd widths, upsampled) keep the K-group-major assignment.
      constexpr int KPT = kChunk / (256 / BM);             // K elements per thread per chunk (pixel-major)
      const int prow = tid % BM, pk0 = (tid / BM) * KPT;
      int pn, pby, pbx;
      {
        const int m = m0 + prow;
        unsigned gx, gy;
        const unsigned t = fastdiv(m < p.M ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
        pn = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
        pby = (int)gy * p.sy;
        pbx = (int)gx * p.sx;
      }
      const bool plive = (m0 + prow) < p.M;
      int rn[AR], rby[AR], rbx[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) row_coords(i, &rn[i], &rby[i], &rbx[i]);
      for (int cl = 0; cl < nch; ++cl) {
        if (!S.vec) {
          float vals[KPT];
#pragma unroll
          for (int e = 0; e < KPT; ++e) {
            const int k = cl * kChunk + pk0 + e;
            unsigned c;
            const int j = (int)fastdiv((unsigned)k, (unsigned)S.C, S.mC, &c);
            float v = 0.f;
            if (j < ntaps) {
              const int tp = taps[j];
              const int iy = pby + (int)(short)(tp & 0xffff), ix = pbx + (tp >> 16);
              if (plive && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
                v = S.p[(long long)pn * S.sn + (long long)(iy >> S.up) * S.sh + (long long)(ix >> S.up) * S.sw + (long long)c * S.sc];
                if (S.scale) v = fmaxf(0.f, v * S.scale[c] + S.shift[c]);
              }
            }
            vals[e] = v;
          }
#pragma unroll
          for (int e = 0; e < KPT; e += 4)
            *reinterpret_cast<f32x4*>(AsB + buf * ABUF + (prow * LDK + pk0 + e) * 4) = f32x4{vals[e], vals[e + 1], vals[e + 2], vals[e + 3]};
        } else {
          const int kl = cl * kChunk + g * 4;
          const int jv = kl / S.C, cv = kl - jv * S.C;
          f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f}, sh4 = f32x4{0.f, 0.f, 0.f, 0.f};
          bool aff = false;
          if (S.scale != nullptr && jv < ntaps) {
            sc4 = *reinterpret_cast<const f32x4*>(S.scale + cv);
            sh4 = *reinterpret_cast<const f32x4*>(S.shift + cv);
            aff = true;
          }
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            AGroup a = gather4(S, kl, ntaps, taps, rn[i], rby[i], rbx[i], (m0 + r0 + 32 * i) < p.M, p.IH, p.IW, jv, cv, 0);
            f32x4 v = a.v;
            if (aff) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * sc4[e] + sh4[e]);
            }
            if (!a.ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(AsB + buf * ABUF + stA + i * ROWS32) = v;
          }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
          *reinterpret_cast<f32x4*>(BsB + buf * BBUF + stA + i * ROWS32) = *reinterpret_cast<const f32x4*>(wrow + boffB[i]);
        wrow += kChunk * 4;
        __syncthreads();
        const char* Ab = AsB + buf * ABUF + frA;
        const char* Bb = BsB + buf * BBUF + frB;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          f32x4 fa[MI], fb[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * ROWS32 + kg * 32);
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) fb[jn] = *reinterpret_cast<const f32x4*>(Bb + jn * ROWS32 + kg * 32);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[jn][kk], acc[i][jn], 0, 0, 0);
        }
        buf ^= 1;      // the next chunk fills the other buffer; the barrier above orders it against this chunk's readers
      }
      __syncthreads();
    };
    if (ntaps == 0) continue;       // empty phase of a strided scatter (e.g. 1x1 stride 2): the result is bias/activation only
    const bool fast = S.vec && S.small && S.up == 0;
    if (fast && S.C % 32 == 0) {
      if (S.scale != nullptr) run_operand(std::true_type{}, std::true_type{});
      else run_operand(std::false_type{}, std::true_type{});
    } else if (fast && (S.C == 4 || S.C == 8 || S.C == 16)) {
      if (S.scale != nullptr) run_operand(std::true_type{}, std::false_type{});
      else run_operand(std::false_type{}, std::false_type{});
    } else {
      run_operand_generic();
    }
  }
  if (ksplit > 1) {
    // partial accumulators -> lane-private float4 slots [tile][split][MI * NI * 4][thread]; the last arrival sums the splits in index
    // order (deterministic).  Visibility as in wino_conv_kernel: every split of a tile runs on the same XCD (the linear block id is
    // blockIdx.x + gridDim.x * (kz + ...) with gridDim.x a multiple of 8), so the partial tiles only have to reach that XCD's L2 --
    // write-through stores waited for with vmcnt(0), an L2 atomic counter, reader loads that bypass the CU's L1 (glc).
    __shared__ int ks_last;
    constexpr int NQ = MI * NI * 4;
    const int tile = (int)blockIdx.z * (MT * NT) + q;
    int* cnt = reinterpret_cast<int*>(p.ks_ws);
    f32x4* slots = reinterpret_cast<f32x4*>(p.ks_ws + p.ks_cnt_floats);
    f32x4* mine = slots + ((size_t)(tile * ksplit + kz) * NQ) * 256 + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          mine[(size_t)((i * NI + jn) * 4 + e) * 256] = f32x4{acc[i][jn][4 * e], acc[i][jn][4 * e + 1], acc[i][jn][4 * e + 2], acc[i][jn][4 * e + 3]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ks_last = (atomicAdd(cnt + tile, 1) == ksplit - 1) ? 1 : 0;
    __syncthreads();
    if (!ks_last) return;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    const __amdgpu_buffer_rsrc_t rws =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(slots + (size_t)tile * ksplit * NQ * 256), 0, 0x7fffffff, 0x00020000);
    for (int z = 0; z < ksplit; ++z) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 vi = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(((z * NQ + (i * NI + jn) * 4 + e) * 256 + tid) * 16), 0, 1 /* glc */);
            const f32x4 v = __builtin_bit_cast(f32x4, vi);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[i][jn][4 * e + u] += v[u];
          }
    }
    if (tid == 0) cnt[tile] = 0;                          // (self-resetting: the workspace is reusable by the next launch on this stream)
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, rowpix, As, m0, n0);
}

// ------------------------------------------------------------------ forward family, three-piece arithmetic, 128 x 128 tile (round 4)
// igemm_conv_x3_kernel splits (BM + BN) x 32 values per 96 matrix instructions of a 128 x 64 / 64 x 128 block -- with K = C_in only (the
// 1x1 convolutions of ResNet bottlenecks: 13.8 of config 4's 39 ms) it is bound by that split, 85-89 TFLOP/s fp32-equivalent.  This
// variant takes the layers whose ONE operand has C % 32 == 0 (every K chunk lies inside one tap: block-uniform tap and channel base) on
// a 128 x 128 tile: (128 + 128) x 32 values per 192 matrix instructions, wave tile 64 x 64 (2 x 2 tiles of 32 x 32).  LDS rows of
// [3 pieces][32 bf16] + 16 bytes (208: the 16-lane fragment reads touch 16 distinct 16-byte slots), ONE buffer + the next chunk's eight
// float4 loads per thread in registers under the current chunk's matrix instructions (two blocks per CU fill each other's staging
// phase).  Epilogue: the shared one (bias, activation, BatchNorm partials, whole-pixel tile stores).
constexpr int X3B_ROWB = 208;

template <bool HA>
__global__ void __launch_bounds__(256, 2) igemm_conv_x3b_kernel(const IgemmParams p) {
  constexpr int BM = 128, BN = 128, WM = 64, WN = 64, MI = 2, NI = 2;
  constexpr int PIECE = 64;                                   // bytes of one piece of a row (32 bf16)
  extern __shared__ __align__(16) float smem[];
  char* T = reinterpret_cast<char*>(smem);                    // [BM + BN rows][X3B_ROWB]; reused by the epilogue's [BM][BN + 4] float tile
  int* rowpix = reinterpret_cast<int*>(T + (size_t)BM * (BN + 4) * sizeof(float));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int MT = (p.M + BM - 1) / BM, NT = p.Npad / BN;
  const int per = (MT * NT + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= MT * NT) return;
  const int m0 = (q / NT) * BM, n0 = (q % NT) * BN;
  const KPhase ph = p.ph[blockIdx.z];
  const int nchunks = ph.nchunks, Kp = nchunks * kChunk;
  const KOperand& S = p.in[0];
  const int cpt = S.C / kChunk;                               // chunks per tap

  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r, pix = -1;
    if (m < p.M) {
      unsigned gx, gy;
      const unsigned t = fastdiv((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
      const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      const int oy = (int)gy * p.osy + ph.ooy, ox = (int)gx * p.osx + ph.oox;
      if (oy < p.OH && ox < p.OW) pix = (n * p.OH + oy) * p.OW + ox;
    }
    rowpix[r] = pix;
  }
  // staging assignment: K group g (4 floats) of rows r0 + 32 i of A (pixels) and of B (output channels)
  const int g = tid & 7, r0 = tid >> 3;
  int rbase[4], rby[4], rbx[4];
  bool rvalid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    rvalid[i] = m < p.M;
    unsigned gx, gy;
    const unsigned t = fastdiv(rvalid[i] ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
    const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
    rby[i] = (int)gy * p.sy;
    rbx[i] = (int)gx * p.sx;
    rbase[i] = n * (int)S.sn + 4 * g;
  }
  const float* wbase = p.w + ph.w_off + (long long)(n0 + r0) * Kp + 4 * g;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 av[4], bv[4], sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  unsigned okm = 0;
  auto issue_loads = [&](int kc) __attribute__((always_inline)) {
    const int tap = kc / cpt, c0 = (kc - tap * cpt) * kChunk;
    const int dy = p.tdy[ph.tap0 + tap], dx = p.tdx[ph.tap0 + tap];
    okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = rby[i] + dy, ix = rbx[i] + dx;
      const bool ok = rvalid[i] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      int off = rbase[i] + iy * (int)S.sh + ix * (int)S.sw + c0;
      asm volatile("" : "+v"(off));
      off = ok ? off : 0;
      av[i] = *reinterpret_cast<const f32x4*>(S.p + off);
      okm |= ok ? (1u << i) : 0u;
      bv[i] = *reinterpret_cast<const f32x4*>(wbase + (long long)(32 * i) * Kp + kc * kChunk);
    }
    if constexpr (HA) {
      sc4 = *reinterpret_cast<const f32x4*>(S.scale + c0 + 4 * g);
      sh4 = *reinterpret_cast<const f32x4*>(S.shift + c0 + 4 * g);
    }
  };
  auto split_store = [&](char* dst, const f32x4& v) __attribute__((always_inline)) {
    const bf16x2 h0 = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), h1 = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
    const f32x2 ra = f32x2{v[0], v[1]} - __builtin_convertvector(h0, f32x2), rb = f32x2{v[2], v[3]} - __builtin_convertvector(h1, f32x2);
    const bf16x2 m0_ = __builtin_convertvector(ra, bf16x2), m1_ = __builtin_convertvector(rb, bf16x2);
    const f32x2 sa = ra - __builtin_convertvector(m0_, f32x2), sb = rb - __builtin_convertvector(m1_, f32x2);
    const bf16x2 l0 = __builtin_convertvector(sa, bf16x2), l1 = __builtin_convertvector(sb, bf16x2);
    *reinterpret_cast<bf16x4*>(dst) = bf16x4{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<bf16x4*>(dst + PIECE) = bf16x4{m0_[0], m0_[1], m1_[0], m1_[1]};
    *reinterpret_cast<bf16x4*>(dst + 2 * PIECE) = bf16x4{l0[0], l0[1], l1[0], l1[1]};
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = av[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[e];
        if constexpr (HA) t = fmaxf(0.f, fmaf(t, sc4[e], sh4[e]));
        v[e] = (okm >> i) & 1u ? t : 0.f;
      }
      split_store(T + (r0 + 32 * i) * X3B_ROWB + g * 8, v);
      split_store(T + (BM + r0 + 32 * i) * X3B_ROWB + g * 8, bv[i]);
    }
  };
  const int frA = (wm * WM + (lane & 31)) * X3B_ROWB + (lane >> 5) * 16;
  const int frB = (BM + wn * WN + (lane & 31)) * X3B_ROWB + (lane >> 5) * 16;
  constexpr int AS[6] = {2, 1, 1, 0, 0, 0}, BS[6] = {0, 0, 1, 0, 1, 2};      // x2y0, x1y0, x1y1, x0y0, x0y1, x0y2: smallest first
  __syncthreads();
  if (nchunks > 0) issue_loads(0);
  for (int kc = 0; kc < nchunks; ++kc) {
    store_lds();
    __syncthreads();
    if (kc + 1 < nchunks) issue_loads(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MI][3], b[NI][3];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int P = 0; P < 3; ++P) a[i][P] = *reinterpret_cast<const bf16x8*>(T + frA + i * 32 * X3B_ROWB + P * PIECE + ks * 32);
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int P = 0; P < 3; ++P) b[j][P] = *reinterpret_cast<const bf16x8*>(T + frB + j * 32 * X3B_ROWB + P * PIECE + ks * 32);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][AS[t]], b[j][BS[t]], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, rowpix, smem, m0, n0);
}

// ------------------------------------------------------------------------------------------------ first layer (stem)
// conv3x3 of a <= 4-channel image (the NCHW user tensor through its strides) to 64 channels, torchvision vgg16_bn features[0]:
// K = 27 is one MFMA chunk, so on the tiled kernel a block's whole "main loop" is a single barrier-bound iteration and the launch
// is all fixed cost (0.41 ms for 5.9 GFLOP).  Here a wave gathers its 32 pixels' taps straight into A fragments (14 dword loads,
// coalesced along x), keeps the 28 x 64 weights in B-fragment registers for the whole kernel, and blocks walk the 128-pixel
// tiles grid-stride; the epilogue (bias, activation, BatchNorm partial statistics per 128-pixel tile) is the shared one.
__global__ void __launch_bounds__(256) stem_conv_kernel(const IgemmParams p) {
  __shared__ __align__(16) float As[4 * 64 + 64];
  __shared__ __align__(16) float Ts[128 * 68];      // the 128 x 64 result tile (row padded to 68): stored as whole 256-byte pixels
  __shared__ int rowpix[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const KOperand& S = p.in[0];
  const int C = S.C, K = 9 * C, kh = lane >> 5;
  constexpr int NS = 18;                           // k-steps of 2: up to 4 channels x 9 taps = 36
  const int nsteps = (K + 1) / 2;
  const int Kp = p.ph[0].nchunks * kChunk;
  // per-lane k-step constants: this lane's k = 2s + kh -> (tap, channel); weights from the packed rows (k = tap*C + c)
  int kdy[NS], kdx[NS], kco[NS];
  float wreg[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = 2 * s + kh;
    const bool live = s < nsteps && k < K;
    const int tap = live ? k / C : 0, c = live ? k - tap * C : 0;
    kdy[s] = live ? (int)p.tdy[tap] : 127;         // 127: dead step (fails the bounds test)
    kdx[s] = (int)p.tdx[tap];
    kco[s] = c * (int)S.sc;
#pragma unroll
    for (int j = 0; j < 2; ++j) wreg[s][j] = live ? p.w[(long long)((lane & 31) + 32 * j) * Kp + k] : 0.f;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  const int ntiles = (p.M + 127) / 128;
  // the gathers of the NEXT tile are issued before this tile's epilogue (LDS tile, statistics, stores: three barriers during which
  // nothing else of this block was in flight)
  auto gather = [&](int tile, float (&dst)[NS]) __attribute__((always_inline)) {
    const int m = tile * 128 + 32 * wave + (lane & 31);
    const bool live = tile < ntiles && m < p.M;
    unsigned gx, gy;
    const unsigned t = fastdiv_dev(live ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
    const int n = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
    const int base = n * (int)S.sn + (int)gy * (int)S.sh + (int)gx * (int)S.sw;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int iy = (int)gy + kdy[s], ix = (int)gx + kdx[s];
      const bool ok = live && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      int off = (base + kdy[s] * (int)S.sh + kdx[s] * (int)S.sw + kco[s]) * 4;
      off = ok ? off : -1;
      dst[s] = s < nsteps ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0)) : 0.f;
    }
  };
  float a[NS], an[NS];
  gather((int)blockIdx.x, a);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 128;
    if (tid < 128) rowpix[tid] = (m0 + tid) < p.M ? m0 + tid : -1;
    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s < nsteps) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[s][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[s][1], acc[0][1], 0, 0, 0);
      }
    }
    gather(tile + (int)gridDim.x, an);
    __syncthreads();
    // bias + activation into the LDS tile (C/D layout: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)), then
    // float4 stores of whole pixels: a wave writes 1 KiB contiguous per instruction instead of 2 x 128 bytes (this kernel is bound
    // by the number of vector-memory instructions, not by bytes)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 32 * j + (lane & 31);
      const float bias = p.bias != nullptr ? p.bias[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * wave + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        Ts[row * 68 + col] = apply_act(acc[0][j][reg] + bias, p.act, p.act_p0, p.act_p1);
      }
    }
    conv_epilogue<128, 64, 32, 64, false>(p, acc, rowpix, As, m0, 0);      // batch-statistic partials only
    __syncthreads();
    {
      const KResult& R = p.out[0];
      const int c4 = tid & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (tid >> 4) + 16 * i;
        if (m0 + row < p.M)
          *reinterpret_cast<f32x4*>(R.p + (long long)(m0 + row) * R.sw + 4 * c4) = *reinterpret_cast<const f32x4*>(Ts + row * 68 + 4 * c4);
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = an[s];
    __syncthreads();
  }
}

static bool stem_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (false) return false;
  if (d->kind != DN_CONV_FWD || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return false;
  if (p.n_in != 1 || p.n_out != 1 || p.Ntot != 64 || p.nphases != 1) return false;
  const KOperand& o = p.in[0];
  const KResult& r = p.out[0];
  return o.C <= 4 && o.up == 0 && o.scale == nullptr && o.small && r.linear && !r.accumulate && (r.sw & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(r.p) & 15) == 0;
}

static int launch_stem(const IgemmParams& p, hipStream_t stream) {
  int blocks = (p.M + 127) / 128;
  if (blocks > 2048) blocks = 2048;
  DN_LAUNCH(stem_conv_kernel, dim3(blocks), dim3(256), 0, stream, p);
  set_last_kernel("dn::stem_conv_kernel");
  return check_launch("stem_conv_kernel");
}

// ----------------------------------------------------------------------------------------------- weight gradient
// ws[split][n][k] = sum over the split's pixels of G[pixel][n] * A[pixel][k].  Tile: BNW (n) x 128 (k), 32 pixels per step.
// ALLVEC (every gathered operand and G float4-addressable with int32 offsets): straight-line staging, see igemm_conv_kernel.
template <int BNW, int WNn, int WKk, bool ALLVEC>
__global__ void __launch_bounds__(256, 2) igemm_wgrad_kernel(const IgemmParams p) {
  constexpr int BKW = 128;
  constexpr int WAVES_K = BKW / WKk;
  constexpr int NI = WNn / 32, KI = WKk / 32;
  constexpr int GR = BNW / 32;  // float4 groups per thread for the G tile
  static_assert((BNW / WNn) * WAVES_K == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* Gs = smem;                                        // [2][32][BNW]
  float* Xs = smem + 2 * 32 * BNW;                         // [2][32][BKW]
  int* taps = reinterpret_cast<int*>(Xs + 2 * 32 * BKW);   // [kMaxTaps]

  if (DN_STAGGER) stagger_priority();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;
  const int kt = blockIdx.x, n0 = blockIdx.y * BNW;
  const KPhase ph = p.ph[0];
  const int ntaps = ph.ntaps, nchunks = ph.nchunks, Kp = nchunks * kChunk;
  const int m_begin = blockIdx.z * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  if (tid < ntaps) taps[tid] = ((int)p.tdy[tid] & 0xffff) | ((int)p.tdx[tid] << 16);
  __syncthreads();

  const int g = tid & 7, r = tid >> 3;  // staging: row r of the 32-pixel step, 4-float group g
  // fixed per-thread K selections for the 4 chunks of this k tile
  int q_s[4], q_j[4], q_c[4], q_kl[4], q_dy[4], q_dx[4];
  bool q_live[4], q_kvalid[4];
  f32x4 xsc[4], xsh[4];
  float q_floor[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int kc = kt * 4 + q;
    q_live[q] = kc < nchunks;
    int kcl = 0;
    q_s[q] = q_live[q] ? select_operand(p, ntaps, kc, &kcl) : 0;
    q_kl[q] = kcl * kChunk + g * 4;
    const KOperand& S = p.in[q_s[q]];
    q_j[q] = q_kl[q] / S.C;
    q_c[q] = q_kl[q] - q_j[q] * S.C;
    q_kvalid[q] = q_live[q] && q_j[q] < ntaps;
    const int t = taps[q_kvalid[q] ? q_j[q] : 0];
    q_dy[q] = (int)(short)(t & 0xffff);
    q_dx[q] = t >> 16;
    if constexpr (ALLVEC) {
      const bool has_aff = S.scale != nullptr;
      const f32x4 l1 = *reinterpret_cast<const f32x4*>((has_aff ? S.scale : S.p) + q_c[q]);
      const f32x4 l2 = *reinterpret_cast<const f32x4*>((has_aff ? S.shift : S.p) + q_c[q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xsc[q][e] = has_aff ? l1[e] : 1.f;
        xsh[q][e] = has_aff ? l2[e] : 0.f;
      }
      q_floor[q] = has_aff ? 0.f : -__builtin_huge_valf();
    }
  }
  const bool gvec = (p.Ntot % 4 == 0);

  f32x16 acc[NI][KI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < KI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 gv[GR];
  AGroup xv[4];
  bool xaff[4];

  auto issue_loads = [&](int mbase) {
    const int m = mbase + r;
    const bool rowvalid = m < m_end;
    if constexpr (ALLVEC) {
      unsigned gx, gy;
      const unsigned mm = rowvalid ? (unsigned)m : 0u;
      const unsigned t = fastdiv(mm, (unsigned)p.GW, p.mGW, &gx);
      const int n = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
      const int by = (int)gy * p.sy, bx = (int)gx * p.sx;
      const int grow = (int)mm * p.Ntot + n0 + g * 4;
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const bool ok = rowvalid && (n0 + g * 4 + 32 * i) < p.Ntot;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.g + (ok ? grow + 32 * i : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[i][e] = ok ? v[e] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const KOperand& S = p.in[q_s[q]];
        int iy = by + q_dy[q], ix = bx + q_dx[q];
        if (p.reflect) {
          iy = reflect_idx(iy, p.IH);
          ix = reflect_idx(ix, p.IW);
        }
        const bool ok = rowvalid && q_kvalid[q] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        int off = n * (int)S.sn + (iy >> S.up) * (int)S.sh + (ix >> S.up) * (int)S.sw + q_c[q];
        off = ok ? off : 0;
        xv[q].v = *reinterpret_cast<const f32x4*>(S.p + off);
        xv[q].ok = ok;
      }
    } else {
      int n = 0, by = 0, bx = 0;
      if (rowvalid) {
        int gx = m % p.GW, t = m / p.GW;
        int gy = t % p.GH;
        n = t / p.GH;
        by = gy * p.sy;
        bx = gx * p.sx;
      }
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const int col = n0 + g * 4 + 32 * i;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rowvalid) {
          const float* gp = p.g + (long long)m * p.Ntot + col;
          if (gvec) {
            if (col < p.Ntot) v = *reinterpret_cast<const f32x4*>(gp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < p.Ntot) v[e] = gp[e];
          }
        }
        gv[i] = v;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xaff[q] = false;
        if (q_live[q]) {
          const KOperand& S = p.in[q_s[q]];
          if (S.vec && S.scale != nullptr && q_j[q] < ntaps) {
            xsc[q] = *reinterpret_cast<const f32x4*>(S.scale + q_c[q]);
            xsh[q] = *reinterpret_cast<const f32x4*>(S.shift + q_c[q]);
            xaff[q] = true;
          }
          xv[q] = gather4(S, q_kl[q], ntaps, taps, n, by, bx, rowvalid, p.IH, p.IW, q_j[q], q_c[q], p.reflect);
        } else {
          xv[q].v = f32x4{0.f, 0.f, 0.f, 0.f};
          xv[q].ok = false;
        }
      }
    }
  };

  auto store_stage = [&](int buf) {
    float* gs = Gs + buf * 32 * BNW + r * BNW + g * 4;
#pragma unroll
    for (int i = 0; i < GR; ++i) *reinterpret_cast<f32x4*>(gs + 32 * i) = gv[i];
    float* xs = Xs + buf * 32 * BKW + r * BKW + g * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = xv[q].v;
      if constexpr (ALLVEC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = fmaxf(q_floor[q], fmaf(v[e], xsc[q][e], xsh[q][e]));
          v[e] = xv[q].ok ? t : 0.f;
        }
      } else {
        if (xaff[q]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * xsc[q][e] + xsh[q][e]);
        }
        if (!xv[q].ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(xs + 32 * q) = v;
    }
  };

  const int nsteps = (m_end > m_begin) ? (m_end - m_begin + 31) / 32 : 0;
  if (nsteps > 0) {
    issue_loads(m_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    const bool more = st + 1 < nsteps;
    if constexpr (ALLVEC) {
      issue_loads(m_begin + (more ? st + 1 : st) * 32);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (more) issue_loads(m_begin + (st + 1) * 32);
    }
    const float* Gb = Gs + buf * 32 * BNW + (lane >> 5) * BNW + wn * WNn + (lane & 31);
    const float* Xb = Xs + buf * 32 * BKW + (lane >> 5) * BKW + wk * WKk + (lane & 31);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      float a[NI], b[KI];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = Gb[s2 * 2 * BNW + i * 32];
#pragma unroll
      for (int j = 0; j < KI; ++j) b[j] = Xb[s2 * 2 * BKW + j * 32];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < KI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (ALLVEC) {
      __builtin_amdgcn_sched_barrier(0);
      store_stage(buf ^ 1);
    } else {
      if (more) store_stage(buf ^ 1);
    }
    __syncthreads();
  }

  float* ws = p.ws + (long long)blockIdx.z * p.Npad * Kp;
#pragma unroll
  for (int j = 0; j < KI; ++j) {
    const int k = kt * BKW + wk * WKk + j * 32 + (lane & 31);
    if (k >= Kp) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn * WNn + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        ws[(long long)n * Kp + k] = acc[i][j][reg];
      }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, fast path
// The 4 K chunks of this block's k tile are LOOP-INVARIANT (the loop runs over pixels), so everything about them is decided
// once: a chunk of a float4-addressable operand (any C % 4 == 0) gives each thread one (tap, channel) for its K group -- kept
// as per-thread registers, block-uniform when C % 32 == 0; a chunk of a scalar operand (the 1-channel disparity piece, the
// 3-channel NCHW image) gives it four (tap, channel) pairs and is gathered pixel-major so the loads coalesce.  Hand-scheduled
// like igemm_conv_u32_kernel: the next 32-pixel step's loads and address arithmetic are dealt under the first MFMAs of the
// current step, the LDS fragment reads one pixel pair ahead of their MFMAs, the store stage under the last MFMAs.  The chunk
// kind tests are block-uniform branches inside the slots; they do not disturb the slot order.
template <int BNW, int WNn, int WKk, bool AFF>
__global__ void __launch_bounds__(256, 2) igemm_wgrad_u32_kernel(const IgemmParams p) {
  constexpr int BKW = 128;
  constexpr int WAVES_K = BKW / WKk;
  constexpr int NI = WNn / 32, KI = WKk / 32;
  constexpr int GR = BNW / 32;  // float4 groups per thread for the G tile
  static_assert((BNW / WNn) * WAVES_K == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* Gs = smem;                                        // [2][32][BNW]
  float* Xs = smem + 2 * 32 * BNW;                         // [2][32][BKW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;
  const KPhase ph = p.ph[0];
  const int ntaps = ph.ntaps, nchunks = ph.nchunks, Kp = nchunks * kChunk;
  // XCD-aware order (see igemm_conv_u32_kernel): the k tiles and n tiles of ONE pixel split are consecutive logical tiles
  // on one XCD, so they run side by side on one L2 and the split's G / X pixels come over the fabric once, not once per tile.
  const int KT = (nchunks + 3) / 4, NTn = p.Npad / BNW;
  const int total = KT * NTn * p.splits;
  const int per = (total + 7) >> 3;
  const int lq = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || lq >= total) return;
  const int kt = lq % KT, n0 = ((lq / KT) % NTn) * BNW, split = lq / (KT * NTn);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int g = tid & 7, r = tid >> 3;      // float4 staging: row r of the 32-pixel step, K group g
  const int g2 = tid >> 5, r2 = tid & 31;   // scalar-chunk staging: pixel-major (lanes = consecutive pixels), K group g2

  // chunk descriptors (block-uniform part in SGPRs, per-thread tap / channel in VGPRs)
  const char* qbase[4];
  int qsn[4], qsh[4], qsw[4], qsc[4], qup[4];
  bool qvec[4], qscal[4];
  int qtap[4];            // vec chunk: this thread's (dy | dx << 16), or 0x80008000 when its K group is past the last tap
  unsigned qoffB[4];      // vec chunk: this thread's channel byte offset
  int qst[4][4];          // scalar chunk: per element (dy & 0xff) | (dx & 0xff) << 8 | channel << 16, or -1 when dead
  f32x4 xsc[4], xsh[4];
  float qfloor[4];
  bool any_scalar = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kc = kt * 4 + q;
    int kcl = 0;
    const bool live = kc < nchunks;
    const int s = live ? select_operand(p, ntaps, kc, &kcl) : 0;
    const KOperand& S = p.in[s];
    qbase[q] = reinterpret_cast<const char*>(S.p);
    qsn[q] = __builtin_amdgcn_readfirstlane((int)S.sn);
    qsh[q] = __builtin_amdgcn_readfirstlane((int)S.sh);
    qsw[q] = __builtin_amdgcn_readfirstlane((int)S.sw);
    qsc[q] = __builtin_amdgcn_readfirstlane((int)S.sc);
    qup[q] = __builtin_amdgcn_readfirstlane(S.up);
    qvec[q] = live && S.vec;
    qscal[q] = live && !S.vec;
    any_scalar = any_scalar || qscal[q];
    {
      unsigned c;
      const int j = (int)fastdiv((unsigned)(kcl * kChunk + g * 4), (unsigned)S.C, S.mC, &c);
      const bool ok = qvec[q] && j < ntaps;
      const int jj = ok ? j : 0;
      qtap[q] = ok ? (((int)p.tdy[jj] & 0xffff) | ((int)p.tdx[jj] << 16)) : (int)0x80008000;
      qoffB[q] = ok ? c * 4u : 0u;
      if constexpr (AFF) {
        const bool has_aff = ok && S.scale != nullptr;
        const f32x4 l1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(has_aff ? S.scale : S.p) + qoffB[q]);
        const f32x4 l2 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(has_aff ? S.shift : S.p) + qoffB[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xsc[q][e] = has_aff ? l1[e] : 1.f;
          xsh[q][e] = has_aff ? l2[e] : 0.f;
        }
        qfloor[q] = has_aff ? 0.f : -__builtin_huge_valf();
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned c;
      const int j = (int)fastdiv((unsigned)(kcl * kChunk + g2 * 4 + e), (unsigned)S.C, S.mC, &c);
      const bool ok = qscal[q] && j < ntaps;
      const int jj = ok ? j : 0;
      qst[q][e] = ok ? (((int)p.tdy[jj] & 0xff) | (((int)p.tdx[jj] & 0xff) << 8) | ((int)c << 16)) : -1;
    }
  }

  f32x16 acc[NI][KI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < KI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const char* gbase = reinterpret_cast<const char*>(p.g);
  f32x4 gv[GR], xv[4];
  bool xok[4], rowvalid = false, rowvalid2 = false;
  int pn = 0, pby = 0, pbx = 0, pn2 = 0, pby2 = 0, pbx2 = 0;
  unsigned goffB = 0;

  auto decode_pixel = [&](int mbase) {
    const int m = mbase + r;
    rowvalid = m < m_end;
    const unsigned mm = rowvalid ? (unsigned)m : 0u;
    unsigned gx, gy;
    const unsigned t = fastdiv(mm, (unsigned)p.GW, p.mGW, &gx);
    pn = (int)fastdiv(t, (unsigned)p.GH, p.mGH, &gy);
    pby = (int)gy * p.sy;
    pbx = (int)gx * p.sx;
    goffB = (mm * (unsigned)p.Ntot + (unsigned)(n0 + g * 4)) * 4u;
    if (any_scalar) {
      const int m2 = mbase + r2;
      rowvalid2 = m2 < m_end;
      const unsigned t2 = fastdiv(rowvalid2 ? (unsigned)m2 : 0u, (unsigned)p.GW, p.mGW, &gx);
      pn2 = (int)fastdiv(t2, (unsigned)p.GH, p.mGH, &gy);
      pby2 = (int)gy * p.sy;
      pbx2 = (int)gx * p.sx;
    }
  };
  auto load_g = [&](int i) {
    // columns past Ntot only exist in the last n tile of a padded Ntot: clamp the address, zero at the store stage
    const bool ok = (n0 + g * 4 + 32 * i) < p.Ntot;
    gv[i] = *reinterpret_cast<const f32x4*>(gbase + (ok ? goffB + 128u * i : 0u));
  };
  auto load_x = [&](int q) {
    if (qscal[q]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = qst[q][e];
        const int iy = pby2 + (int)(signed char)(d & 0xff), ix = pbx2 + (int)(signed char)((d >> 8) & 0xff);
        const bool ok = (int)rowvalid2 & (int)(d >= 0) & (int)((unsigned)iy < (unsigned)p.IH) & (int)((unsigned)ix < (unsigned)p.IW);
        unsigned off = (unsigned)((pn2 * qsn[q] + (iy >> qup[q]) * qsh[q] + (ix >> qup[q]) * qsw[q] + (d >> 16) * qsc[q]) * 4);
        asm volatile("" : "+v"(off));
        off = ok ? off : 0u;
        const float v = *reinterpret_cast<const float*>(qbase[q] + off);
        xv[q][e] = ok ? v : 0.f;
      }
      xok[q] = true;
    } else {
      const int tp = qtap[q];
      const int iy = pby + (int)(short)(tp & 0xffff), ix = pbx + (tp >> 16);
      xok[q] = (int)rowvalid & (int)qvec[q] & (int)((unsigned)iy < (unsigned)p.IH) & (int)((unsigned)ix < (unsigned)p.IW);
      unsigned off = (unsigned)((pn * qsn[q] + (iy >> qup[q]) * qsh[q] + (ix >> qup[q]) * qsw[q]) * 4) + qoffB[q];
      asm volatile("" : "+v"(off));            // keep the address arithmetic unconditional (no exec-masked region, no branch)
      off = xok[q] ? off : 0u;
      xv[q] = *reinterpret_cast<const f32x4*>(qbase[q] + off);
    }
  };
  char* GsB = reinterpret_cast<char*>(Gs);
  char* XsB = reinterpret_cast<char*>(Xs);
  constexpr int GBUF = 32 * BNW * 4, XBUF = 32 * BKW * 4;
  const int stG = (r * BNW + g * 4) * 4, stX = (r * BKW + g * 4) * 4, stX2 = (r2 * BKW + g2 * 4) * 4;
  bool rowvalid_st = false;      // validity of the row whose data sits in gv/xv (snapshotted at load time)
  auto store_g = [&](int b, int i) {
    f32x4 v = gv[i];
    const bool ok = (int)rowvalid_st & (int)((n0 + g * 4 + 32 * i) < p.Ntot);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
    *reinterpret_cast<f32x4*>(GsB + b * GBUF + stG + i * 128) = v;
  };
  auto store_x = [&](int b, int q) {
    f32x4 v = xv[q];
    if (qscal[q]) {
      *reinterpret_cast<f32x4*>(XsB + b * XBUF + stX2 + q * 128) = v;      // zero fill already applied per element
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[e];
        if constexpr (AFF) t = fmaxf(qfloor[q], fmaf(t, xsc[q][e], xsh[q][e]));
        v[e] = xok[q] ? t : 0.f;
      }
      *reinterpret_cast<f32x4*>(XsB + b * XBUF + stX + q * 128) = v;
    }
  };

  constexpr int NM = 16 * NI * KI;                 // MFMAs per 32-pixel step
  constexpr int PS = NI * KI;                      // MFMAs per pixel pair
  constexpr int NS = GR + 4;                       // store-stage items
  constexpr int SSTEP = (NM >= 4 * NS) ? 2 : 1;
  constexpr int S0 = NM - SSTEP * NS;

  const int nsteps = (m_end > m_begin) ? (m_end - m_begin + 31) / 32 : 0;
  if (nsteps > 0) {
    decode_pixel(m_begin);
    rowvalid_st = rowvalid;
#pragma unroll
    for (int i = 0; i < GR; ++i) load_g(i);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_x(q);
#pragma unroll
    for (int i = 0; i < GR; ++i) store_g(0, i);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_x(0, q);
  }
  __syncthreads();
  const int frG = ((lane >> 5) * BNW + wn * WNn + (lane & 31)) * 4;
  const int frX = ((lane >> 5) * BKW + wk * WKk + (lane & 31)) * 4;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    const int mnext = m_begin + (st + 1 < nsteps ? st + 1 : st) * 32;   // last step re-fetches itself into the idle buffer
    const char* Gb = GsB + buf * GBUF + frG;
    const char* Xb = XsB + buf * XBUF + frX;
    float fa[2][NI], fb[2][KI];
#pragma unroll
    for (int i = 0; i < NI; ++i) fa[0][i] = *reinterpret_cast<const float*>(Gb + i * 128);
#pragma unroll
    for (int j = 0; j < KI; ++j) fb[0][j] = *reinterpret_cast<const float*>(Xb + j * 128);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NM>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int s2 = m / PS, ij = m % PS;
      constexpr int i = ij / KI, j = ij % KI;
      constexpr int cur = s2 & 1, nxt = cur ^ 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      // ---- side work of this slot
      if (m == 0) { decode_pixel(mnext); }
      if (m >= 1 && m < 1 + GR) load_g(m - 1);
      if (m >= 1 + GR && m < 5 + GR) load_x(m - 1 - GR);
      if constexpr (s2 < 15) {     // fragments of the next pixel pair, spread over this pair's slots
        if constexpr (PS >= NI + KI) {
          if constexpr (ij < NI) fa[nxt][ij] = *reinterpret_cast<const float*>(Gb + (s2 + 1) * 2 * BNW * 4 + ij * 128);
          else if constexpr (ij < NI + KI) fb[nxt][ij - NI] = *reinterpret_cast<const float*>(Xb + (s2 + 1) * 2 * BKW * 4 + (ij - NI) * 128);
        } else if constexpr (ij == 0) {
#pragma unroll
          for (int a = 0; a < NI; ++a) fa[nxt][a] = *reinterpret_cast<const float*>(Gb + (s2 + 1) * 2 * BNW * 4 + a * 128);
#pragma unroll
          for (int b = 0; b < KI; ++b) fb[nxt][b] = *reinterpret_cast<const float*>(Xb + (s2 + 1) * 2 * BKW * 4 + b * 128);
        }
      }
      if (m >= S0 && (m - S0) % SSTEP == 0) {
        const int it = (m - S0) / SSTEP;
        if (it == 0) rowvalid_st = rowvalid;
        if (it < GR) store_g(buf ^ 1, it);
        else if (it < NS) store_x(buf ^ 1, it - GR);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  }

  float* ws = p.ws + (long long)split * p.Npad * Kp;
#pragma unroll
  for (int j = 0; j < KI; ++j) {
    const int k = kt * BKW + wk * WKk + j * 32 + (lane & 31);
    if (k >= Kp) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn * WNn + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        ws[(long long)n * Kp + k] = acc[i][j][reg];
      }
  }
}

// packed (n, k) -> framework weight index, or -1 for a padding slot
__device__ __forceinline__ long long packed_to_framework(const IgemmParams& p, const KPhase& ph, int n, int k) {
  if (n >= p.Ntot) return -1;
  int s = 0, kl = k;
  for (int i = 0; i < p.n_in - 1; ++i) {
    int span = ((ph.ntaps * p.in[i].C + kChunk - 1) / kChunk) * kChunk;
    if (s == i && kl >= span) {
      kl -= span;
      s = i + 1;
    }
  }
  const int C = p.in[s].C;
  const int j = kl / C, c = kl - j * C;
  if (j >= ph.ntaps) return -1;
  const int cc = p.in[s].ch_off + c;
  const int r = p.tr[ph.tap0 + j], t = p.ts[ph.tap0 + j];
  const long long rs = (long long)p.R * p.S;
  const long long base = p.n_is_dim0 ? ((long long)n * p.D1 + cc) : ((long long)cc * p.D1 + n);
  return base * rs + r * p.S + t;
}

__device__ __forceinline__ void pack_weights_body(const IgemmParams& p, const float* __restrict__ w, float* __restrict__ wp, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int z = 0;
    while (z + 1 < p.nphases && idx >= p.ph[z + 1].w_off) ++z;
    const KPhase& ph = p.ph[z];
    const long long local = idx - ph.w_off;
    const int Kp = ph.nchunks * kChunk;
    const int n = (int)(local / Kp), k = (int)(local - (long long)n * Kp);
    const long long src = packed_to_framework(p, ph, n, k);
    wp[idx] = src >= 0 ? w[src] : 0.f;
  }
}

__global__ void pack_weights_kernel(const IgemmParams p, const float* __restrict__ w, float* __restrict__ wp, long long total) {
  pack_weights_body(p, w, wp, total);
}

__global__ void pack_weights_many_kernel(const PackEntry* __restrict__ tab) {
  const PackEntry& e = tab[blockIdx.y];
  pack_weights_body(e.p, e.w, e.wp, e.total);
}

// 64 consecutive packed elements per block (coalesced 256-byte rows of every split slab); the four waves take the splits z = w, w + 4,
// ... and meet through LDS in a fixed order (deterministic).  The thin full-resolution layers have a few thousand weights and hundreds
// of splits: one thread per element walking all of them serially ran 10 blocks for up to 90 us.
// NW waves per block: 4 for the tiled kernels' handful of splits, 16 for the hundreds of block slabs of the persistent thin-layer
// kernels (round 4: 40 blocks of 4 waves walking 128 slabs each took 30-60 us per launch, as long as half the kernel they follow); a
// wave keeps four running sums (z = w + NW (4 i + u)) so that its loads are in flight four deep.  The order is fixed by the indices.
template <int NW>
__global__ void __launch_bounds__(64 * NW) wgrad_reduce_kernel(const IgemmParams p, float* __restrict__ dw) {
  __shared__ float part[NW][64];
  const KPhase& ph = p.ph[0];
  const int Kp = ph.nchunks * kChunk;
  const long long total = (long long)p.Ntot * Kp;
  const long long slab = (long long)p.Npad * Kp;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (long long base = blockIdx.x * 64ll; base < total; base += (long long)gridDim.x * 64) {
    const long long idx = base + lane;
    const bool live = idx < total;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      int z = w;
      for (; z + 3 * NW < p.splits; z += 4 * NW) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += p.ws[(z + u * NW) * slab + idx];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (z + u * NW < p.splits) s[u] += p.ws[(z + u * NW) * slab + idx];
    }
    part[w][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (w == 0 && live) {
      float tot = part[0][lane];
#pragma unroll
      for (int q = 1; q < NW; ++q) tot += part[q][lane];
      const int n = (int)(idx / Kp), k = (int)(idx - (long long)n * Kp);
      const long long dst = packed_to_framework(p, ph, n, k);
      if (dst >= 0) dw[dst] = tot;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------ launchers
template <typename K>
static int enable_big_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return DN_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(max dynamic LDS %zu): %s", bytes, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  return DN_OK;
}

template <int BM, int BN, int WM, int WN, bool ALLVEC>
static int launch_conv_v(const IgemmParams& p, hipStream_t stream) {
  size_t lds = (size_t)(2 * BM * LDK + 2 * BN * LDK) * sizeof(float) + (kMaxTaps + BM) * sizeof(int);
  lds += (size_t)0;   // tuning aid: lowers blocks per CU
  auto kernel = igemm_conv_kernel<BM, BN, WM, WN, ALLVEC>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  dim3 grid((p.M + BM - 1) / BM, p.Npad / BN, p.nphases);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::igemm_conv_kernel<%d, %d, %d, %d, %s>", BM, BN, WM, WN, ALLVEC ? "true" : "false");
  return check_launch("igemm_conv_kernel");
}

template <int BM, int BN, int WM, int WN>
static int launch_conv_u32(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * BM * LDK + 2 * BN * LDK) * sizeof(float) + (32 + BM) * sizeof(int);
  auto kernel = igemm_conv_u32_kernel<BM, BN, WM, WN>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.Npad / BN);
  dim3 grid((tiles + 7) / 8 * 8, 1, p.nphases);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::igemm_conv_u32_kernel<%d, %d, %d, %d>", BM, BN, WM, WN);
  return check_launch("igemm_conv_u32_kernel");
}

// K split of the three-piece direct kernel for small grids (see the kernel): how many ways, and the workspace that takes.
constexpr int kX3SplitKCounterBytes = 4096;      // int counters [phase][tile], zero between launches (self-resetting), in front of the partial tiles
static int x3_splitk_choice(const IgemmParams& p, int tiles) {
  if (knobs().no_x3_splitk || p.n_in != 1) return 1;
  const KOperand& S = p.in[0];
  if (!(S.vec && S.small && S.up == 0 && (S.C % 32 == 0 || S.C == 4 || S.C == 8 || S.C == 16))) return 1;   // the scheduled loaders only
  const int blocks = tiles * p.nphases;
  if (blocks > 208 || blocks > (int)(kX3SplitKCounterBytes / sizeof(int))) return 1;
  int nch = 1 << 30;                               // fewest chunks of a (non-empty) phase
  for (int z = 0; z < p.nphases; ++z) {
    const int nt = p.ph[z].ntaps;
    if (nt == 0) continue;
    const int c = S.C % 32 == 0 ? nt * (S.C >> 5) : (nt + 32 / S.C - 1) / (32 / S.C);
    if (c < nch) nch = c;
  }
  if (nch == (1 << 30)) return 1;
  int ks = 512 / blocks;
  if (ks > nch / 8) ks = nch / 8;
  if (ks > 16) ks = 16;
  return ks < 2 ? 1 : ks;
}
static size_t x3_splitk_workspace_bytes(int blocks, int ks, int tiles_per_wave) {
  return kX3SplitKCounterBytes + (size_t)blocks * ks * tiles_per_wave * 4 * 256 * sizeof(float) * 4;
}

// upper bound over the tile shapes run_conv may pick (64-row tiles give the most blocks; two 32 x 32 tiles per wave at most)
size_t conv_x3_splitk_workspace_upper_bytes(const IgemmParams& p) {
  if (p.compute != DN_COMPUTE_F32X3 || !p.uni32 || p.BN < 64) return 0;
  const int tiles = ((p.M + 63) / 64) * (p.Npad / p.BN);
  const int ks = x3_splitk_choice(p, tiles);
  return ks > 1 ? x3_splitk_workspace_bytes(tiles * p.nphases, ks, 2) : 0;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv_x3(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * BM + 2 * BN) * X3ROW + (32 + BM) * sizeof(int);
  auto kernel = igemm_conv_x3_kernel<BM, BN, WM, WN>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.Npad / BN);
  IgemmParams q = p;
  q.ksplit = 1;
  {
    // K split of small grids: one scheduled operand, few blocks, many chunks.  Needs the caller's zeroed workspace (dn_conv_desc.splitk_ws).
    const int ks = x3_splitk_choice(p, tiles);
    if (ks > 1 && p.ks_ws != nullptr && p.ks_ws_bytes >= x3_splitk_workspace_bytes(tiles * p.nphases, ks, (WM / 32) * (WN / 32))) {
      q.ksplit = ks;
      q.ks_cnt_floats = kX3SplitKCounterBytes / 4;
    }
  }
  dim3 grid((tiles + 7) / 8 * 8, q.ksplit, p.nphases);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, q);
  set_last_kernel("dn::igemm_conv_x3_kernel<%d, %d, %d, %d>", BM, BN, WM, WN);
  return check_launch("igemm_conv_x3_kernel");
}

static bool conv_x3b_eligible(const IgemmParams& p) {
  if (p.compute != DN_COMPUTE_F32X3 || !p.uni32 || p.n_in != 1 || p.BN != 128 || p.reflect) return false;
  const KOperand& S = p.in[0];
  if (!(S.vec && S.small && S.up == 0 && S.C % kChunk == 0)) return false;
  if (S.scale != nullptr && ((reinterpret_cast<uintptr_t>(S.scale) | reinterpret_cast<uintptr_t>(S.shift)) & 15)) return false;
  const long long tiles = (long long)((p.M + 127) / 128) * (p.Npad / 128) * p.nphases;
  return tiles >= 192;                                        // smaller grids: the 64-row tiles / K splits of igemm_conv_x3_kernel
}

static int launch_conv_x3b(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)128 * (128 + 4) * sizeof(float) + 128 * sizeof(int);
  const int tiles = ((p.M + 127) / 128) * (p.Npad / 128);
  dim3 grid((tiles + 7) / 8 * 8, 1, p.nphases);
  int rc;
  if (p.in[0].scale != nullptr) {
    rc = enable_big_lds(igemm_conv_x3b_kernel<true>, lds);
    if (rc != DN_OK) return rc;
    DN_LAUNCH(igemm_conv_x3b_kernel<true>, grid, dim3(256), lds, stream, p);
  } else {
    rc = enable_big_lds(igemm_conv_x3b_kernel<false>, lds);
    if (rc != DN_OK) return rc;
    DN_LAUNCH(igemm_conv_x3b_kernel<false>, grid, dim3(256), lds, stream, p);
  }
  set_last_kernel("dn::igemm_conv_x3b_kernel<%s>", p.in[0].scale != nullptr ? "true" : "false");
  return check_launch("igemm_conv_x3b_kernel");
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(const IgemmParams& p, hipStream_t stream) {
  if (p.uni32 && !false)
    return launch_conv_u32<BM, BN, WM, WN>(p, stream);
  return p.allvec ? launch_conv_v<BM, BN, WM, WN, true>(p, stream) : launch_conv_v<BM, BN, WM, WN, false>(p, stream);
}

static int run_conv(const dn_conv_desc* d, int expect_kind, dn_stream_t stream) {
  DN_REQUIRE(d != nullptr && d->kind == expect_kind, DN_ERR_BAD_ARG, "descriptor kind mismatch (want %d)", expect_kind);
  IgemmParams p;
  int rc = build_plan(d, false, &p);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(p.w != nullptr, DN_ERR_BAD_ARG, "w_packed is null");
  for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
  for (int i = 0; i < p.n_out; ++i) {
    const dn_result& o = d->out[i];
    DN_REQUIRE(o.stride_h == (int64_t)d->OW * o.stride_w && o.stride_n == (int64_t)d->OH * o.stride_h, DN_ERR_UNSUPPORTED,
               "result %d must be pixel-dense (NHWC with a channel stride)", i);
  }
  hipStream_t s = as_stream(stream);
  if (!knobs().no_direct) {
    if (head_fwd_eligible(d, p)) return launch_head_fwd(p, s);
    if (head_dgrad_eligible(d, p)) return launch_head_dgrad(p, s);
  }
  if (const int wl = wino_layout(d, p)) {
    p.compute = wl == 2 ? DN_COMPUTE_BF16 : (wl == 3 ? DN_COMPUTE_F32X3 : DN_COMPUTE_F32);
    return launch_wino_conv(p, s);
  }
  if (stem3_conv_eligible(d, p)) return launch_stem3_conv(p, s);
  if (stemk_conv_eligible(d, p)) return launch_stemk_conv(d, p, s);
  if (stem_eligible(d, p)) return launch_stem(p, s);
  if (lds3_conv_eligible(d, p)) return launch_lds3_conv(d, p, s);
  if (lds3k_conv_eligible(d, p)) return launch_lds3k_conv(d, p, s);
  if (thin_conv_eligible(d, p)) return launch_thin_conv(p, s);
  // Few row tiles (the 4x13 / 8x26 decoder levels at b32: 13-52 tiles of 128 rows) leave most of the 256 CUs without a block;
  // 64-row tiles double the block count at the same per-wave MFMA density along N.  Not with batch statistics: the
  // bn_partial layout is per 128-row tile.
  const long long blocks128 = (long long)((p.M + 127) / 128) * (p.Npad / p.BN) * p.nphases;
  const bool small_m = p.uni32 && p.bn_partial == nullptr && blocks128 <= 208 && !false;
  if (p.compute == DN_COMPUTE_F32X3 && p.uni32 && !false && !knobs().no_x3_direct && (p.BN >= 64 || false)) {
    // (the 32-wide N tile -- one 32 x 32 tile per wave, 12 matrix instructions per chunk against five split-and-store items -- measured
    //  4-17 % slower than the fp32 instruction: it stays on that)
    // fp32 products on the bf16 matrix cores (wave tiles of at most 2 x 32 x 32: the 128-wide N tile runs as 64-row blocks)
    if (conv_x3b_eligible(p)) return launch_conv_x3b(p, s);       // 128 x 128 tile: one operand with C % 32 == 0, enough tiles (round 4)
    switch (p.BN) {
      case 128: return p.bn_partial == nullptr ? launch_conv_x3<64, 128, 32, 64>(p, s) : launch_conv_x3<128, 64, 64, 32>(p, s);   // (statistics rows are per 128-row tile)
      case 64: return (small_m || (p.bn_partial == nullptr && blocks128 <= 416)) ? launch_conv_x3<64, 64, 32, 32>(p, s) : launch_conv_x3<128, 64, 64, 32>(p, s);
      default: return launch_conv_x3<128, 32, 32, 32>(p, s);
    }
  }
  switch (p.BN) {
    case 128: return small_m ? launch_conv_u32<64, 128, 32, 64>(p, s) : launch_conv<128, 128, 64, 64>(p, s);
    case 64: return small_m ? launch_conv_u32<64, 64, 32, 32>(p, s) : launch_conv<128, 64, 64, 32>(p, s);
    default: return launch_conv<128, 32, 32, 32>(p, s);
  }
}

template <int BNW, int WNn, int WKk, bool ALLVEC>
static int launch_wgrad_v(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * 32 * BNW + 2 * 32 * 128) * sizeof(float) + kMaxTaps * sizeof(int);
  auto kernel = igemm_wgrad_kernel<BNW, WNn, WKk, ALLVEC>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  dim3 grid((p.ph[0].nchunks + 3) / 4, p.Npad / BNW, p.splits);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::igemm_wgrad_kernel<%d, %d, %d, %s>", BNW, WNn, WKk, ALLVEC ? "true" : "false");
  return check_launch("igemm_wgrad_kernel");
}

template <int BNW, int WNn, int WKk, bool AFF>
static int launch_wgrad_u32(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * 32 * BNW + 2 * 32 * 128) * sizeof(float);
  auto kernel = igemm_wgrad_u32_kernel<BNW, WNn, WKk, AFF>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  const int total = ((p.ph[0].nchunks + 3) / 4) * (p.Npad / BNW) * p.splits;
  dim3 grid((total + 7) / 8 * 8);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::igemm_wgrad_u32_kernel<%d, %d, %d, %s>", BNW, WNn, WKk, AFF ? "true" : "false");
  return check_launch("igemm_wgrad_u32_kernel");
}

template <int BNW, int WNn, int WKk>
static int launch_wgrad(const IgemmParams& p, hipStream_t stream) {
  if (p.wg_uniform && !false)
    return p.any_affine ? launch_wgrad_u32<BNW, WNn, WKk, true>(p, stream) : launch_wgrad_u32<BNW, WNn, WKk, false>(p, stream);
  return p.allvec ? launch_wgrad_v<BNW, WNn, WKk, true>(p, stream) : launch_wgrad_v<BNW, WNn, WKk, false>(p, stream);
}

// Pixel splits of the weight gradient.  Every block does the same amount of work and the chip holds `slots` blocks at once
// (256 CUs x blocks per CU, LDS-limited), so the launch should fill a whole number of rounds from below: tiles * splits just
// under R * slots (1026 blocks on 1024 slots run as THREE rounds, measured 103 vs 135 TFLOP/s).  Fewest rounds that reach
// 92 % slot use wins (fewer splits = fewer partial slabs for wgrad_reduce_kernel).
static void choose_splits(IgemmParams* p) {
  const int tiles = ((p->ph[0].nchunks + 3) / 4) * (p->Npad / p->BN);
  const int per_cu = p->BN >= 128 ? 2 : 3;      // LDS (128-wide) resp. registers (narrower tiles) limit the blocks per CU
  const int slots = 256 * per_cu;
  int max_by_work = (p->M + 255) / 256;  // at least 8 steps of 32 pixels per split
  if (max_by_work < 1) max_by_work = 1;
  int best = 1;
  double best_util = 0.0;
  for (int R = 1; R <= 4; ++R) {
    int sp = (R * slots) / tiles;
    if (sp < 1) continue;
    if (sp > max_by_work) sp = max_by_work;
    const double util = (double)tiles * sp / ((double)((tiles * sp + slots - 1) / slots) * slots);
    if (util > best_util + 1e-9) {
      best_util = util;
      best = sp;
    }
    if (util >= 0.92) break;
  }
  int per = (p->M + best - 1) / best;
  per = (per + 31) / 32 * 32;
  p->m_per_split = per;
  p->splits = (p->M + per - 1) / per;
}

int launch_wgrad_reduce(const IgemmParams& p, float* dw, hipStream_t stream) {
  const long long total = (long long)p.Ntot * p.ph[0].nchunks * kChunk;
  int blocks = (int)((total + 63) / 64);
  if (blocks > 8192) blocks = 8192;
  if (p.splits >= 32) DN_LAUNCH(wgrad_reduce_kernel<16>, dim3(blocks), dim3(1024), 0, stream, p, dw);
  else DN_LAUNCH(wgrad_reduce_kernel<4>, dim3(blocks), dim3(256), 0, stream, p, dw);
  return check_launch("wgrad_reduce_kernel");
}

}  // namespace dn

using namespace dn;

// The tiled weight-gradient kernels + their fixed-order split sum (every layer the Winograd / thin / head kernels do not take).
static int generic_wgrad(const dn_conv_desc* fwd, IgemmParams& p, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                         hipStream_t s) {
  int rc = DN_OK;
  choose_splits(&p);
  const size_t need = (size_t)p.splits * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
  DN_REQUIRE(workspace_bytes >= need, DN_ERR_WORKSPACE, "wgrad workspace too small: %zu < %zu", workspace_bytes, need);
  p.ws = reinterpret_cast<float*>(workspace);
  if (fwd->kind == DN_CONV_FWD) {
    p.g = dy;  // [N*OH*OW][Cout]
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
  } else {
    // conv-transpose: G = forward input x [N*IH*IW][Cin] (dense NHWC), gathered operand = dy [N][OH][OW][Cout]
    const dn_operand& x = fwd->in[0];
    DN_REQUIRE(x.data != nullptr && x.stride_c == 1 && x.stride_w == x.C && x.stride_h == (int64_t)fwd->IW * x.C &&
                   x.stride_n == (int64_t)fwd->IH * fwd->IW * x.C,
               DN_ERR_UNSUPPORTED, "conv-transpose wgrad needs a dense NHWC input");
    p.g = x.data;
    KOperand& o = p.in[0];
    const int co = o.C;
    o.p = dy;
    o.scale = o.shift = nullptr;
    o.sc = 1;
    o.sw = co;
    o.sh = (long long)fwd->OW * co;
    o.sn = (long long)fwd->OH * fwd->OW * co;
    o.up = 0;
    o.vec = (co % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) ? 1 : 0;
    o.mC = fastdiv_magic((unsigned)co);
    o.small = ((long long)fwd->N * o.sn < (1ll << 31)) ? 1 : 0;
    p.allvec = (o.vec && o.small) ? 1 : 0;
    p.any_affine = 0;
    p.wg_uniform = (p.allvec && co % 32 == 0 && p.ph[0].ntaps <= 32 && (long long)fwd->N * o.sn * 4 + 64 < (1ll << 31)) ? 1 : 0;
  }
  // the G operand must be float4-addressable with int32 offsets too
  if (!(p.Ntot % 4 == 0 && (reinterpret_cast<uintptr_t>(p.g) & 15) == 0 && (long long)p.M * p.Ntot < (1ll << 31))) p.allvec = 0;
  // the G operand must be float4-addressable with 32-bit BYTE offsets for the fast kernel
  if (!(p.Ntot % 4 == 0 && (reinterpret_cast<uintptr_t>(p.g) & 15) == 0) || (long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31)) p.wg_uniform = 0;
  if (wgrad_x3_eligible(p)) {
    rc = launch_wgrad_x3(p, s);           // fp32 products on the bf16 matrix cores (dn_wgrad_x3.hip)
  } else {
    switch (p.BN) {
      case 128: rc = launch_wgrad<128, 64, 64>(p, s); break;
      case 64: rc = launch_wgrad<64, 64, 32>(p, s); break;
      default: rc = launch_wgrad<32, 32, 32>(p, s); break;
    }
  }
  if (rc != DN_OK) return rc;
  return launch_wgrad_reduce(p, dw, s);
}

// A concatenated input whose LAST piece is a 1-channel map (the upsampled disparity of the iconv layers: 64 + 128 + 1, 64 + 256 + 1):
// the 1-channel piece is what keeps the layer off the Winograd weight-gradient kernel (operands in multiples of 64).  Split the
// gradient instead: Winograd for the pieces in front (rows of dw written with the full layer's channel stride), the tiled kernel for
// the one trailing channel (9 columns of dw).  iconv2 at 32 images: 0.272 -> 0.15 ms; config 4's 321 -> 64 @120x160: 1.25 -> 0.5 ms.
static bool wgrad_split_plans(const dn_conv_desc* fwd, dn_conv_desc* d1, dn_conv_desc* d2, IgemmParams* p1, IgemmParams* p2, size_t* w1,
                              size_t* w2) {
  if (false || fwd->kind != DN_CONV_FWD || fwd->n_in < 2 || fwd->in[fwd->n_in - 1].C != 1) return false;
  *d1 = *fwd;
  d1->n_in = fwd->n_in - 1;
  *d2 = *fwd;
  d2->n_in = 1;
  d2->in[0] = fwd->in[fwd->n_in - 1];
  if (build_plan(d1, true, p1) != DN_OK || build_plan(d2, true, p2) != DN_OK) return false;
  int cin_total = 0;
  for (int i = 0; i < fwd->n_in; ++i) cin_total += fwd->in[i].C;
  if (wino_wgrad_eligible(d1, *p1)) {
    p1->dw_cin_total = cin_total;
    *w1 = (wino_wgrad_workspace_bytes(*p1) + 255) / 256 * 256;
  } else {
    // (round 4 measured the pieces in front on the three-piece tiled kernel: SLOWER -- iconv1 97 -> 32 @64x208 0.375 -> 0.508 ms, a 32-wide
    //  n tile is bound by its staging; the switch that kept that path alive was removed in round 5)
    return false;
  }
  p2->in[0].ch_off = cin_total - 1;              // (packed_to_framework: the column block of this channel in the full weight tensor)
  p2->D1 = cin_total;
  choose_splits(p2);
  *w2 = (size_t)p2->splits * p2->Npad * p2->ph[0].nchunks * kChunk * sizeof(float);
  return true;
}

// Kernels of more than 32 taps (the 7x7 / stride-2 first layers of ResNet-50 and PoseExpNet: reference models/Disp_res_50.py:65,141,
// models/PoseExpNet.py:28): the scheduled weight-gradient kernels carry one validity bit per tap in a 32-bit word, so such layers fell to
// the unscheduled kernel (1.30 ms at 18 TFLOP/s in config 4, 0.57 ms in config 3).  The taps are independent columns of dW, so the
// gradient is taken as two launches over tap WINDOWS of <= 32 taps each, every one with the full plan's operands and its own rows of the
// tap tables; the split sum of each window writes only its own (r, s) entries of dw.
static int tap_windows(const IgemmParams& p) {
  const int nt = p.ph[0].ntaps;
  if (nt <= 32 || p.reflect || p.nphases != 1 || knobs().no_tap_windows) return 1;
  return (nt + 31) / 32;
}

static void tap_window_plan(const IgemmParams& full, int w, int nw, IgemmParams* q) {
  *q = full;
  const int nt = full.ph[0].ntaps, per = (nt + nw - 1) / nw;
  const int t0 = w * per, t1 = t0 + per < nt ? t0 + per : nt;
  for (int t = t0; t < t1; ++t) {
    q->tdy[t - t0] = full.tdy[full.ph[0].tap0 + t];
    q->tdx[t - t0] = full.tdx[full.ph[0].tap0 + t];
    q->tr[t - t0] = full.tr[full.ph[0].tap0 + t];
    q->ts[t - t0] = full.ts[full.ph[0].tap0 + t];
  }
  q->ph[0].tap0 = 0;
  q->ph[0].ntaps = t1 - t0;
  int nch = 0;
  for (int i = 0; i < q->n_in; ++i) nch += (q->ph[0].ntaps * q->in[i].C + kChunk - 1) / kChunk;
  q->ph[0].nchunks = nch;
  q->uni32 = 1;
  q->wg_uniform = 1;
  for (int i = 0; i < q->n_in; ++i) {
    const KOperand& o = q->in[i];
    if (!o.small) q->uni32 = 0;
    if (!o.small || (!o.vec && o.scale != nullptr) || o.C >= 32768) q->wg_uniform = 0;
  }
}

extern "C" {

int64_t dn_pack_entry_bytes(void) { return (int64_t)sizeof(PackEntry); }

int dn_pack_entry_fill(const dn_conv_desc* d, const float* w, float* w_packed, void* entry_host) {
  DN_REQUIRE(d && w && w_packed && entry_host, DN_ERR_BAD_ARG, "dn_pack_entry_fill: null pointer");
  PackEntry* e = reinterpret_cast<PackEntry*>(entry_host);
  int rc = build_plan(d, false, &e->p);
  if (rc != DN_OK) return rc;
  e->w = w;
  e->wp = w_packed;
  if (const int wl = wino_layout(d, e->p)) {
    e->wino = wl;
    e->total = wino_packed_elems(e->p);
    e->NS = ((e->p.Ntot + 63) / 64 * 64) / 32;
  } else {
    e->wino = 0;
    const KPhase& last = e->p.ph[e->p.nphases - 1];
    e->total = last.w_off + (long long)e->p.Npad * last.nchunks * kChunk;
    e->NS = 0;
  }
  return e->wino;
}

int dn_pack_many(const void* entries_dev, int32_t n_direct, int32_t n_wino, int32_t n_wino16, int32_t n_wino_x3, dn_stream_t stream) {
  DN_REQUIRE(entries_dev && n_direct >= 0 && n_wino >= 0 && n_wino16 >= 0 && n_wino_x3 >= 0, DN_ERR_BAD_ARG, "dn_pack_many: bad argument");
  const PackEntry* tab = reinterpret_cast<const PackEntry*>(entries_dev);
  hipStream_t s = as_stream(stream);
  if (n_direct > 0) {
    DN_LAUNCH(pack_weights_many_kernel, dim3(knobs().pack_blocks, n_direct), dim3(256), 0, s, tab);
    int rc = check_launch("pack_weights_many_kernel");
    if (rc != DN_OK) return rc;
  }
  if (n_wino > 0) {
    int rc = launch_wino_pack_many(tab, n_direct, n_wino, s);
    if (rc != DN_OK) return rc;
  }
  if (n_wino16 > 0) {
    int rc = launch_wino_pack16_many(tab, n_direct + n_wino, n_wino16, 1, s);
    if (rc != DN_OK) return rc;
  }
  if (n_wino_x3 > 0) return launch_wino_pack16_many(tab, n_direct + n_wino + n_wino16, n_wino_x3, 3, s);
  return DN_OK;
}

int dn_conv_pack_weights(const dn_conv_desc* d, const float* w, float* w_packed, dn_stream_t stream) {
  IgemmParams p;
  int rc = build_plan(d, false, &p);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(w != nullptr && w_packed != nullptr, DN_ERR_BAD_ARG, "null weight pointer");
  if (const int wl = wino_layout(d, p))
    return wl == 1 ? launch_wino_pack(p, w, w_packed, as_stream(stream)) : launch_wino_pack16(p, w, w_packed, wl == 3 ? 3 : 1, as_stream(stream));
  const KPhase& last = p.ph[p.nphases - 1];
  const long long total = last.w_off + (long long)p.Npad * last.nchunks * kChunk;
  if (total == 0) return DN_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  DN_LAUNCH(pack_weights_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, w, w_packed, total);
  return check_launch("pack_weights_kernel");
}

int dn_conv2d_fwd(const dn_conv_desc* d, dn_stream_t stream) { return run_conv(d, DN_CONV_FWD, stream); }
int dn_conv2d_dgrad(const dn_conv_desc* d, dn_stream_t stream) { return run_conv(d, DN_CONV_DGRAD, stream); }
int dn_convT2d_fwd(const dn_conv_desc* d, dn_stream_t stream) { return run_conv(d, DN_CONVT_FWD, stream); }
int dn_convT2d_dgrad(const dn_conv_desc* d, dn_stream_t stream) { return run_conv(d, DN_CONVT_DGRAD, stream); }

size_t dn_conv_wgrad_workspace_bytes(const dn_conv_desc* fwd) {
  IgemmParams p;
  if (build_plan(fwd, true, &p) != DN_OK) return 0;
  choose_splits(&p);
  size_t need = (size_t)p.splits * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
  if (head_wgrad_eligible(fwd, p) && head_wgrad_workspace_bytes(p) > need) need = head_wgrad_workspace_bytes(p);
  if (wino_wgrad_eligible(fwd, p) && wino_wgrad_workspace_bytes(p) > need) need = wino_wgrad_workspace_bytes(p);
  if (thin_wgrad_eligible(fwd, p) && thin_wgrad_workspace_bytes(p) > need) need = thin_wgrad_workspace_bytes(p);
  if (lds3_wgrad_eligible(fwd, p) && lds3_wgrad_workspace_bytes(p) > need) need = lds3_wgrad_workspace_bytes(p);
  if (lds3k_wgrad_eligible(fwd, p) && lds3k_wgrad_workspace_bytes(p) > need) need = lds3k_wgrad_workspace_bytes(p);
  if (stemk_wgrad_workspace_bytes(fwd, p) > need) need = stemk_wgrad_workspace_bytes(fwd, p);
  {
    dn_conv_desc d1, d2;
    IgemmParams p1, p2;
    size_t w1 = 0, w2 = 0;
    if (wgrad_split_plans(fwd, &d1, &d2, &p1, &p2, &w1, &w2) && w1 + w2 > need) need = w1 + w2;
  }
  const int nw = tap_windows(p);
  for (int w = 0; w < nw && nw > 1; ++w) {
    IgemmParams q;
    tap_window_plan(p, w, nw, &q);
    choose_splits(&q);
    const size_t wneed = (size_t)q.splits * q.Npad * q.ph[0].nchunks * kChunk * sizeof(float);
    if (wneed > need) need = wneed;
  }
  return need;
}

int dn_conv2d_wgrad(const dn_conv_desc* fwd, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                    dn_stream_t stream) {
  IgemmParams p;
  int rc = build_plan(fwd, true, &p);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(dy != nullptr && dw != nullptr && workspace != nullptr, DN_ERR_BAD_ARG, "null pointer");
  if (head_wgrad_eligible(fwd, p) && workspace_bytes >= head_wgrad_workspace_bytes(p) && !knobs().no_direct) {
    DN_REQUIRE(p.in[0].p != nullptr, DN_ERR_BAD_ARG, "operand 0 has no data");
    p.g = dy;
    return launch_head_wgrad(p, dw, reinterpret_cast<float*>(workspace), as_stream(stream));
  }
  if (wino_wgrad_eligible(fwd, p) && workspace_bytes >= wino_wgrad_workspace_bytes(p)) {
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
    p.g = dy;
    p.ws = reinterpret_cast<float*>(workspace);
    return launch_wino_wgrad(p, dw, as_stream(stream));
  }
  if (lds3_wgrad_eligible(fwd, p) && workspace_bytes >= lds3_wgrad_workspace_bytes(p) && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
    p.g = dy;
    p.ws = reinterpret_cast<float*>(workspace);
    return launch_lds3_wgrad(fwd, p, dw, as_stream(stream));
  }
  if (lds3k_wgrad_eligible(fwd, p) && workspace_bytes >= lds3k_wgrad_workspace_bytes(p) && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
    p.g = dy;
    p.ws = reinterpret_cast<float*>(workspace);
    return launch_lds3k_wgrad(fwd, p, dw, as_stream(stream));
  }
  if (stemk_wgrad_eligible(fwd, p) && workspace_bytes >= stemk_wgrad_workspace_bytes(fwd, p) && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
    p.g = dy;
    p.ws = reinterpret_cast<float*>(workspace);
    return launch_stemk_wgrad(fwd, p, dw, as_stream(stream));        // 7x7 / stride-2 first layers on NCHW images (dn_stemk.hip)
  }
  if (thin_wgrad_eligible(fwd, p) && workspace_bytes >= thin_wgrad_workspace_bytes(p)) {
    for (int i = 0; i < p.n_in; ++i) DN_REQUIRE(p.in[i].p != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
    p.g = dy;
    p.ws = reinterpret_cast<float*>(workspace);
    return launch_thin_wgrad(p, dw, as_stream(stream));
  }
  {
    dn_conv_desc d1, d2;
    IgemmParams p1, p2;
    size_t w1 = 0, w2 = 0;
    if (wgrad_split_plans(fwd, &d1, &d2, &p1, &p2, &w1, &w2) && workspace_bytes >= w1 + w2) {
      for (int i = 0; i < fwd->n_in; ++i) DN_REQUIRE(fwd->in[i].data != nullptr, DN_ERR_BAD_ARG, "operand %d has no data", i);
      if (wino_wgrad_eligible(&d1, p1)) {
        p1.g = dy;
        p1.ws = reinterpret_cast<float*>(workspace);
        rc = launch_wino_wgrad(p1, dw, as_stream(stream));
      } else {
        rc = generic_wgrad(&d1, p1, dy, dw, workspace, w1, as_stream(stream));
      }
      if (rc != DN_OK) return rc;
      return generic_wgrad(&d2, p2, dy, dw, reinterpret_cast<char*>(workspace) + w1, w2, as_stream(stream));
    }
  }
  if (const int nw = tap_windows(p); nw > 1 && fwd->kind == DN_CONV_FWD) {
    for (int w = 0; w < nw; ++w) {
      IgemmParams q;
      tap_window_plan(p, w, nw, &q);
      rc = generic_wgrad(fwd, q, dy, dw, workspace, workspace_bytes, as_stream(stream));
      if (rc != DN_OK) return rc;
    }
    return DN_OK;
  }
  return generic_wgrad(fwd, p, dy, dw, workspace, workspace_bytes, as_stream(stream));
}

}  // extern "C"
