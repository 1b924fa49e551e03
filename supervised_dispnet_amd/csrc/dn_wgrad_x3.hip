// Tiled weight gradient with fp32 products on the bf16 matrix cores (round 4; the three-piece form of igemm_wgrad_u32_kernel, dn_conv.hip).
//   ws[split][n][k] = sum over the split's pixels of G[pixel][n] * X[pixel][k]      (G = dy, X = the gathered input; conv-transpose: swapped)
// The contraction runs over PIXELS, so on v_mfma_f32_32x32x16_bf16 a lane must hold eight consecutive pixels of one channel, while NHWC
// tensors hold the channels of one pixel together.  The transposition is done by the staging threads (as in dn_lds3_wgrad.hip): a thread
// loads 8 pixels x 4 channels (eight float4: pixel rows of G, or the (tap, channel) group of its K chunk with bounds / pending-BatchNorm
// handling per pixel), splits every value ONCE into its three exact bf16 pieces (DN_COMPUTE_F32X3) and writes per
// channel and piece the eight pixels as one 16-byte LDS word: LDS holds [piece][channel][32 pixels] rows of 80 bytes (64 + 16: the 16-lane
// fragment reads and the staging writes both touch 16 distinct 16-byte slots).  A 32-pixel step = two 16-deep matrix steps; six partial
// products per step (x0y0, x0y1, x1y0, x0y2, x1y1, x2y0).  Tile BNW (n) x 128 (k), four waves as 2 x 2, the next step's global loads in
// registers under the current step's matrix instructions, one LDS buffer (two barriers per step; the CU's second block fills the gaps).
// Everything around it -- pixel splits filling whole rounds, XCD-aware tile order, the fixed-order split sum -- is the fp32 kernel's.
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wx_split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
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

__device__ __forceinline__ int wx_select_operand(const IgemmParams& p, int ntaps, int kc, int* kc_local) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < DN_MAX_OPERANDS - 1; ++i) {
    if (s == i && i < p.n_in - 1) {
      const int nch = (ntaps * p.in[i].C + kChunk - 1) / kChunk;
      if (kc >= nch) {
        kc -= nch;
        s = i + 1;
      }
    }
  }
  *kc_local = kc;
  return s;
}

constexpr int WX_ROWB = 80;          // bytes of one channel row: 32 pixels x bf16 + 16 (bank stagger)

template <int BNW, bool AFF>
__global__ void __launch_bounds__(256, 2) igemm_wgrad_x3_kernel(const IgemmParams p) {
  constexpr int BKW = 128;
  constexpr int NCH = BNW + BKW, PIECE = NCH * WX_ROWB;
  constexpr int WAVES_K = BNW == 32 ? 4 : 2, WAVES_N = 4 / WAVES_K;       // four waves as 2 x 2 (1 x 4 for the 32-wide n tile)
  constexpr int WNn = BNW / WAVES_N, WKk = BKW / WAVES_K;
  constexpr int NI = WNn / 32, KI = WKk / 32;      // 32 x 32 tiles per wave
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;
  const KPhase ph = p.ph[0];
  const int ntaps = ph.ntaps, nchunks = ph.nchunks, Kp = nchunks * kChunk;
  // XCD-aware order (igemm_wgrad_u32_kernel): the k tiles and n tiles of ONE pixel split are consecutive logical tiles on one XCD
  const int KT = (nchunks + 3) / 4, NTn = p.Npad / BNW;
  const int total = KT * NTn * p.splits;
  const int per = (total + 7) >> 3;
  const int lq = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || lq >= total) return;
  const int kt = lq % KT, n0 = ((lq / KT) % NTn) * BNW, split = lq / (KT * NTn);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // ---- this thread's staging job (wave-uniform kind): G jobs = threads [0, BNW), X jobs = threads [BNW, BNW + 128)
  const bool gjob = tid < BNW, xjob = tid >= BNW && tid < BNW + BKW;
  const int jt = gjob ? tid : tid - BNW;
  const int pg = jt & 3, quad = jt >> 2;                   // 8-pixel group of the step, channel quad
  const char* base = reinterpret_cast<const char*>(p.g);
  int qsn = 0, qsh = 0, qsw = 0, qup = 0, qdy = 0, qdx = 0;
  unsigned choffB = 0;
  bool live = false;
  f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsh = {0.f, 0.f, 0.f, 0.f};
  float xfloor = -__builtin_huge_valf();
  if (gjob) {
    live = (n0 + 4 * quad) < p.Ntot;
    choffB = (unsigned)(n0 + 4 * quad) * 4u;
  } else if (xjob) {
    const int q = quad >> 3, g8 = quad & 7;
    const int kc = kt * 4 + q;
    int kcl = 0;
    const bool chunk_live = kc < nchunks;
    const int s = chunk_live ? wx_select_operand(p, ntaps, kc, &kcl) : 0;
    const KOperand& S = p.in[s];
    unsigned c;
    const int j = (int)fastdiv_dev((unsigned)(kcl * kChunk + g8 * 4), (unsigned)S.C, S.mC, &c);
    live = chunk_live && j < ntaps;
    const int jj = live ? j : 0;
    base = reinterpret_cast<const char*>(S.p);
    qsn = (int)S.sn; qsh = (int)S.sh; qsw = (int)S.sw; qup = S.up;
    qdy = p.tdy[jj]; qdx = p.tdx[jj];
    choffB = live ? c * 4u : 0u;
    if constexpr (AFF) {
      if (live && S.scale != nullptr) {
        xsc = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(S.scale) + choffB);
        xsh = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(S.shift) + choffB);
        xfloor = 0.f;
      }
    }
  }

  f32x16 acc[NI][KI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < KI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 v[8];
  unsigned okmask = 0;
  auto issue_loads = [&](int mbase) __attribute__((always_inline)) {
    okmask = 0;
    const int m0 = mbase + 8 * pg;
    unsigned gx, gy;
    const unsigned t = fastdiv_dev((unsigned)(m0 < p.M ? m0 : 0), (unsigned)p.GW, p.mGW, &gx);
    int nn = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
    int x = (int)gx, y = (int)gy;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + i;
      bool ok = live && m < m_end;
      unsigned off;
      if (gjob) {
        off = (unsigned)m * (unsigned)p.Ntot * 4u + choffB;
      } else {
        const int iy = y * p.sy + qdy, ix = x * p.sx + qdx;
        ok = ok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        off = (unsigned)((nn * qsn + (iy >> qup) * qsh + (ix >> qup) * qsw) * 4) + choffB;
      }
      asm volatile("" : "+v"(off));                // keep the address arithmetic unconditional (no exec-masked region)
      off = ok ? off : 0u;
      v[i] = *reinterpret_cast<const f32x4*>(base + off);
      okmask |= ok ? (1u << i) : 0u;
      // next pixel of the group (GW >= 8: at most one row wrap per group)
      x += 1;
      if (x >= p.GW) {
        x = 0;
        y += 1;
        if (y >= p.GH) {
          y = 0;
          nn += 1;
        }
      }
    }
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
    if (gjob || xjob) {
      const int chrow = gjob ? 4 * quad : BNW + 4 * quad;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = v[i][e];
          if constexpr (AFF) t = fmaxf(xfloor, fmaf(t, xsc[e], xsh[e]));
          f[i] = (okmask >> i) & 1u ? t : 0.f;
        }
        bf16x8 h, m, l;
        wx_split3(f, h, m, l);
        char* dst = lds + (chrow + e) * WX_ROWB + pg * 16;
        *reinterpret_cast<bf16x8*>(dst) = h;
        *reinterpret_cast<bf16x8*>(dst + PIECE) = m;
        *reinterpret_cast<bf16x8*>(dst + 2 * PIECE) = l;
      }
    }
  };

  const int nsteps = (m_end > m_begin) ? (m_end - m_begin + 31) / 32 : 0;
  const int frA = (wn * WNn + (lane & 31)) * WX_ROWB + (lane >> 5) * 16;
  const int frB = (BNW + wk * WKk + (lane & 31)) * WX_ROWB + (lane >> 5) * 16;
  constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
  if (nsteps > 0) issue_loads(m_begin);
  for (int st = 0; st < nsteps; ++st) {
    store_lds();
    __syncthreads();
    if (st + 1 < nsteps) issue_loads(m_begin + (st + 1) * 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[NI][3], b[KI][3];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int P = 0; P < 3; ++P) a[i][P] = *reinterpret_cast<const bf16x8*>(lds + P * PIECE + frA + i * 32 * WX_ROWB + ks * 32);
#pragma unroll
      for (int j = 0; j < KI; ++j)
#pragma unroll
        for (int P = 0; P < 3; ++P) b[j][P] = *reinterpret_cast<const bf16x8*>(lds + P * PIECE + frB + j * 32 * WX_ROWB + ks * 32);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < KI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][AS[q]], b[j][BS[q]], acc[i][j], 0, 0, 0);
    }
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

// plans the kernel takes: the fast weight-gradient plan (<= 32 taps, zero padding, 32-bit offsets) whose operands are all float4-addressable,
// and a grid row of at least 8 pixels (an 8-pixel staging group wraps at most once)
bool wgrad_x3_eligible(const IgemmParams& p) {
  if (knobs().no_x3_wgrad || p.compute != DN_COMPUTE_F32X3 || !p.wg_uniform || p.GW < 8) return false;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!(o.vec && o.small)) return false;
    if (o.scale != nullptr && ((reinterpret_cast<uintptr_t>(o.scale) | reinterpret_cast<uintptr_t>(o.shift)) & 15)) return false;
  }
  return true;
}

template <int BNW, bool AFF>
static int launch_wx(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)3 * (BNW + 128) * WX_ROWB;
  auto kernel = igemm_wgrad_x3_kernel<BNW, AFF>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS %zu): %s", lds, hipGetErrorString(e));
      return DN_ERR_LAUNCH;
    }
  }
  const int total = ((p.ph[0].nchunks + 3) / 4) * (p.Npad / BNW) * p.splits;
  dim3 grid((total + 7) / 8 * 8);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::igemm_wgrad_x3_kernel<%d, %s>", BNW, AFF ? "true" : "false");
  return check_launch("igemm_wgrad_x3_kernel");
}

int launch_wgrad_x3(const IgemmParams& p, hipStream_t stream) {
  if (p.BN == 128) return p.any_affine ? launch_wx<128, true>(p, stream) : launch_wx<128, false>(p, stream);
  if (p.BN == 64) return p.any_affine ? launch_wx<64, true>(p, stream) : launch_wx<64, false>(p, stream);
  return p.any_affine ? launch_wx<32, true>(p, stream) : launch_wx<32, false>(p, stream);
}

}  // namespace dn
