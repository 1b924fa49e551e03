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

#include "dn_internal.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
                                          bool rowvalid, int IH, int IW, int j_vec, int c_vec) {
  AGroup r;
  r.v = f32x4{0.f, 0.f, 0.f, 0.f};
  r.ok = false;
  if (S.vec) {
    if (rowvalid && j_vec < ntaps) {
      int t = taps[j_vec];
      int iy = by + (int)(short)(t & 0xffff), ix = bx + (t >> 16);
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

// ------------------------------------------------------------------------------------------------ forward family
template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) igemm_conv_kernel(const IgemmParams p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                                        // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;                         // [2][BN][LDK]
  int* taps = reinterpret_cast<int*>(Bs + 2 * BN * LDK);   // [kMaxTaps]  (dy | dx<<16)
  int* rowpix = taps + kMaxTaps;                           // [BM] output pixel index or -1

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
      int gx = m % p.GW, t = m / p.GW;
      int gy = t % p.GH, n = t / p.GH;
      int oy = gy * p.osy + ph.ooy, ox = gx * p.osx + ph.oox;
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
      int gx = m % p.GW, t = m / p.GW;
      int gy = t % p.GH;
      rn[i] = t / p.GH;
      rby[i] = gy * p.sy;
      rbx[i] = gx * p.sx;
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

  auto issue_loads = [&](int kc) {
    int kcl;
    const int s = select_operand(p, ntaps, kc, &kcl);
    const KOperand& S = p.in[s];
    const int kl = kcl * kChunk + g * 4;
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
    for (int i = 0; i < AR; ++i) av[i] = gather4(S, kl, ntaps, taps, rn[i], rby[i], rbx[i], rn[i] >= 0, p.IH, p.IW, j, c);
    const float* wrow = p.w + ph.w_off + (long long)(n0 + r0) * Kp + kc * kChunk + g * 4;
#pragma unroll
    for (int i = 0; i < BR; ++i) bv[i] = *reinterpret_cast<const f32x4*>(wrow + (long long)(32 * i) * Kp);
  };

  auto store_stage = [&](int buf) {
    float* a = As + buf * BM * LDK + r0 * LDK + g * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = av[i].v;
      if (aff) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * sc4[e] + sh4[e]);
      }
      if (!av[i].ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (more) issue_loads(kc + 1);
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
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, channel-split store; optional batch-statistic partials
  // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
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
    // per-column sums of the PRE-BIAS accumulators (rows past M are exact zeros and do not disturb them)
    float* red = As;  // reuse: [WAVES_M][BN][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float v = acc[i][j][reg];
          s1 += v;
          s2 += v * v;
        }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) {
        int col = wn * WN + j * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = s1;
        red[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < BM / WM; ++w) {
        s1 += red[(w * BN + tid) * 2 + 0];
        s2 += red[(w * BN + tid) * 2 + 1];
      }
      int n = n0 + tid;
      if (n < p.Ntot) {
        float* dst = p.bn_partial + ((long long)blockIdx.x * p.Ntot + n) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------- weight gradient
// ws[split][n][k] = sum over the split's pixels of G[pixel][n] * A[pixel][k].  Tile: BNW (n) x 128 (k), 32 pixels per step.
template <int BNW, int WNn, int WKk>
__global__ void __launch_bounds__(256) igemm_wgrad_kernel(const IgemmParams p) {
  constexpr int BKW = 128;
  constexpr int WAVES_K = BKW / WKk;
  constexpr int NI = WNn / 32, KI = WKk / 32;
  constexpr int GR = BNW / 32;  // float4 groups per thread for the G tile
  static_assert((BNW / WNn) * WAVES_K == 4, "4 waves per block");
  extern __shared__ __align__(16) float smem[];
  float* Gs = smem;                                        // [2][32][BNW]
  float* Xs = smem + 2 * 32 * BNW;                         // [2][32][BKW]
  int* taps = reinterpret_cast<int*>(Xs + 2 * 32 * BKW);   // [kMaxTaps]

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
  int q_s[4], q_j[4], q_c[4], q_kl[4];
  bool q_live[4];
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
  f32x4 xsc[4], xsh[4];
  bool xaff[4];

  auto issue_loads = [&](int mbase) {
    const int m = mbase + r;
    const bool rowvalid = m < m_end;
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
        xv[q] = gather4(S, q_kl[q], ntaps, taps, n, by, bx, rowvalid, p.IH, p.IW, q_j[q], q_c[q]);
      } else {
        xv[q].v = f32x4{0.f, 0.f, 0.f, 0.f};
        xv[q].ok = false;
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
      if (xaff[q]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * xsc[q][e] + xsh[q][e]);
      }
      if (!xv[q].ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (more) issue_loads(m_begin + (st + 1) * 32);
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
    if (more) store_stage(buf ^ 1);
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

__global__ void pack_weights_kernel(const IgemmParams p, const float* __restrict__ w, float* __restrict__ wp, long long total) {
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

__global__ void wgrad_reduce_kernel(const IgemmParams p, float* __restrict__ dw) {
  const KPhase& ph = p.ph[0];
  const int Kp = ph.nchunks * kChunk;
  const long long total = (long long)p.Ntot * Kp;
  const long long slab = (long long)p.Npad * Kp;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / Kp), k = (int)(idx - (long long)n * Kp);
    const long long dst = packed_to_framework(p, ph, n, k);
    if (dst < 0) continue;
    float s = 0.f;
    for (int z = 0; z < p.splits; ++z) s += p.ws[z * slab + idx];
    dw[dst] = s;
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

template <int BM, int BN, int WM, int WN>
static int launch_conv(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * BM * LDK + 2 * BN * LDK) * sizeof(float) + (kMaxTaps + BM) * sizeof(int);
  auto kernel = igemm_conv_kernel<BM, BN, WM, WN>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  dim3 grid((p.M + BM - 1) / BM, p.Npad / BN, p.nphases);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, p);
  return check_launch("igemm_conv_kernel");
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
  switch (p.BN) {
    case 128: return launch_conv<128, 128, 64, 64>(p, s);
    case 64: return launch_conv<128, 64, 64, 32>(p, s);
    default: return launch_conv<128, 32, 32, 32>(p, s);
  }
}

template <int BNW, int WNn, int WKk>
static int launch_wgrad(const IgemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(2 * 32 * BNW + 2 * 32 * 128) * sizeof(float) + kMaxTaps * sizeof(int);
  auto kernel = igemm_wgrad_kernel<BNW, WNn, WKk>;
  int rc = enable_big_lds(kernel, lds);
  if (rc != DN_OK) return rc;
  dim3 grid((p.ph[0].nchunks + 3) / 4, p.Npad / BNW, p.splits);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, p);
  return check_launch("igemm_wgrad_kernel");
}

static void choose_splits(IgemmParams* p) {
  const int tiles = ((p->ph[0].nchunks + 3) / 4) * (p->Npad / p->BN);
  int want = (1024 + tiles - 1) / tiles;
  int max_by_work = (p->M + 255) / 256;  // at least 8 steps of 32 pixels per split
  int splits = want < 1 ? 1 : want;
  if (splits > max_by_work) splits = max_by_work;
  if (splits < 1) splits = 1;
  int per = (p->M + splits - 1) / splits;
  per = (per + 31) / 32 * 32;
  p->m_per_split = per;
  p->splits = (p->M + per - 1) / per;
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_conv_pack_weights(const dn_conv_desc* d, const float* w, float* w_packed, dn_stream_t stream) {
  IgemmParams p;
  int rc = build_plan(d, false, &p);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(w != nullptr && w_packed != nullptr, DN_ERR_BAD_ARG, "null weight pointer");
  const KPhase& last = p.ph[p.nphases - 1];
  const long long total = last.w_off + (long long)p.Npad * last.nchunks * kChunk;
  if (total == 0) return DN_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, w, w_packed, total);
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
  return (size_t)p.splits * p.Npad * p.ph[0].nchunks * kChunk * sizeof(float);
}

int dn_conv2d_wgrad(const dn_conv_desc* fwd, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                    dn_stream_t stream) {
  IgemmParams p;
  int rc = build_plan(fwd, true, &p);
  if (rc != DN_OK) return rc;
  DN_REQUIRE(dy != nullptr && dw != nullptr && workspace != nullptr, DN_ERR_BAD_ARG, "null pointer");
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
  }
  hipStream_t s = as_stream(stream);
  switch (p.BN) {
    case 128: rc = launch_wgrad<128, 64, 64>(p, s); break;
    case 64: rc = launch_wgrad<64, 64, 32>(p, s); break;
    default: rc = launch_wgrad<32, 32, 32>(p, s); break;
  }
  if (rc != DN_OK) return rc;
  const long long total = (long long)p.Ntot * p.ph[0].nchunks * kChunk;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, dw);
  return check_launch("wgrad_reduce_kernel");
}

}  // extern "C"
