// Helpers shared by the Winograd F(2x2,3x3) forward / input-gradient kernels (dn_winograd.hip, dn_winograd8.hip).  Not part of the ABI.
#pragma once
#include <type_traits>
#include <utility>

#include "dn_internal.h"
#include "dn_fold.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void wino_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  wino_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int WBT = 64;            // tiles per block
constexpr int WBN = 64;            // output channels per block
constexpr int WKC = 16;            // channels per staged chunk (two 8-k MFMA groups)
constexpr int WZLD = 72;           // padded row (floats) of the cross-wave exchange tile: rows 4 apart land 32 banks apart

__device__ __forceinline__ float wino_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

template <int MTW>
struct WinoCfg {
  static constexpr int BT = 32 * MTW;                 // tiles per block
  static constexpr int HALFB = BT * 16 + 32;          // bytes of one [tile][4 k] plane (+8 banks)
  static constexpr int POSB = 2 * HALFB;              // one position: k 0-3 plane, k 4-7 plane
  static constexpr int SUBB = 16 * POSB + 64;         // one 8-k group: 16 positions (+16 banks)
  static constexpr int BUFB = 2 * SUBB;               // one 16-channel chunk
  static constexpr size_t LDS = (size_t)2 * BUFB;     // double-buffered ring
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int W16_ROWB = 48;                          // bytes of one tile's 16 bf16 (+16 padding)
template <int VW> struct VecOf;
template <> struct VecOf<4> { typedef f32x4 type; };
template <> struct VecOf<2> { typedef f32x2 type; };

template <int VW>
__device__ __forceinline__ typename VecOf<VW>::type buffer_load_vec(__amdgpu_buffer_rsrc_t r, int voffset) {
  if constexpr (VW == 4) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, 0, 0);
    return __builtin_bit_cast(f32x4, v);
  } else {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voffset, 0, 0);
    return __builtin_bit_cast(f32x2, v);
  }
}



// BatchNorm-backward column sums of the layer below, from the input-gradient tile a Winograd block has just produced (dn_conv_desc.bnb_*):
// thread (tg, c4) holds Y[k][a][b] = dx of pixels (a, b) of tiles tg + TSTEP * k, channels n_first .. n_first + 3.  Per 32-tile group g of
// the block:  bnb_partial[groups_per_block * mb + g][n][0 .. 1] = (sum dz, sum dz * xhat),  dz = dx * [y * scale + shift > 0].  `red`: LDS
// scratch of GROUPS * NWAVES * 64 * 2 floats (free after the output exchange); fixed summation order.
template <int NK, int TSTEP, int NWAVES, int GROUPS>
__device__ __forceinline__ void wino_bn_bwd_sums(const IgemmParams& p, const f32x4 (&Y)[NK][2][2], int mb, int tg, int c4, int wave, int lane, int tid,
                                                 int n_first, float* red) {
  constexpr int BT = 32 * GROUPS;
  f32x4 s1[GROUPS], s2[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n_first < p.Ntot) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.bnb_scale + n_first), sh = *reinterpret_cast<const f32x4*>(p.bnb_shift + n_first);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bnb_mean + n_first), is = *reinterpret_cast<const f32x4*>(p.bnb_invstd + n_first);
    const long long C = p.Ntot, rowB = (long long)p.OW * C;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int t = mb * BT + tg + TSTEP * k;
      if (t < p.T) {
        unsigned tx, ty;
        const unsigned r = fastdiv_dev((unsigned)t, (unsigned)p.TW, p.mTW, &tx);
        const int n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
        const float* y00 = p.bnb_y + (((long long)n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx) * C + n_first;
        const f32x4 yv[2][2] = {{*reinterpret_cast<const f32x4*>(y00), *reinterpret_cast<const f32x4*>(y00 + C)},
                                {*reinterpret_cast<const f32x4*>(y00 + rowB), *reinterpret_cast<const f32x4*>(y00 + rowB + C)}};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float yy = yv[a][b][e];
              const float dz = (yy * sc[e] + sh[e] > 0.f) ? Y[k][a][b][e] : 0.f;
              s1[(k * TSTEP) / 32][e] += dz;
              s2[(k * TSTEP) / 32][e] += dz * (yy - mu[e]) * is[e];
            }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < GROUPS; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[g][e] += __shfl_xor(s1[g][e], 16);
      s1[g][e] += __shfl_xor(s1[g][e], 32);
      s2[g][e] += __shfl_xor(s2[g][e], 16);
      s2[g][e] += __shfl_xor(s2[g][e], 32);
    }
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      *reinterpret_cast<f32x4*>(red + ((g * NWAVES + wave) * 2 + 0) * 64 + 4 * c4) = s1[g];
      *reinterpret_cast<f32x4*>(red + ((g * NWAVES + wave) * 2 + 1) * 64 + 4 * c4) = s2[g];
    }
  }
  __syncthreads();
  if (tid < 64 * GROUPS) {
    const int g = tid >> 6, col = tid & 63;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      a1 += red[((g * NWAVES + w) * 2 + 0) * 64 + col];
      a2 += red[((g * NWAVES + w) * 2 + 1) * 64 + col];
    }
    const int n = (n_first - 4 * c4) + col;           // nb * WBN + col
    if (n < p.Ntot && mb * BT + 32 * g < p.T) {
      float* dst = p.bnb_partial + ((long long)(GROUPS * mb + g) * p.Ntot + n) * 2;
      fold_store(dst, a1);             // (agent scope: the block that arrives last may read them, dn_fold.h)
      fold_store(dst + 1, a2);
    }
  }
}

// Last-arrival epilogues of the Winograd forward / input-gradient kernels (dn_fold.h), called by every thread at the very END of a block
// that ran the epilogue (its statistics / sums rows are stored): the block that arrives last for its 64-channel slice `nb` finishes the
// BatchNorm statistics of the slice (p.fold_bn: what dn_bn_finalize would do next) or the two BatchNorm-backward sums (p.fold_bnb: what
// the BatchNorm backward's own sums launch would do).  `mblocks` = blocks per slice that get here, `lds` >= 6 KB of free LDS.
__device__ __forceinline__ void wino_fold_tail(const IgemmParams& p, int nb, int mblocks, float* lds, int tid) {
  if (!(p.fold_bn | p.fold_bnb)) return;
  __shared__ int fold_flag;
  if (!fold_last_arrival(p.fold_cnt + nb, mblocks, &fold_flag)) return;
  const int rows = (p.T + 31) / 32, c0 = nb * WBN;
  const int nlive = p.Ntot - c0 < WBN ? p.Ntot - c0 : WBN;
  double* red = reinterpret_cast<double*>(lds);
  if (p.fold_bn) bn_finalize_sliced<true>(p.bn_partial, rows, p.Ntot, c0, nlive, (double)p.M, p.bnf, red, tid);
  if (p.fold_bnb) colsum2_sliced<true>(p.bnb_partial, rows, p.Ntot, 2, 0, c0, nlive, p.bnb_dbeta, p.bnb_dgamma, red, tid);
}

// dn_winograd8.hip: the 8-wave / one-block-per-CU form of the three-piece kernel (64 tiles x 64 output channels, two positions per wave)
bool wino8_wanted(const IgemmParams& p);
int launch_wino_conv8(const IgemmParams& p, hipStream_t stream);

}  // namespace dn
