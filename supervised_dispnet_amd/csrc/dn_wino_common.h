// Helpers shared by the Winograd F(2x2,3x3) forward / input-gradient kernels (dn_winograd.hip, dn_winograd8.hip).  Not part of the ABI.
#pragma once
#include <type_traits>
#include <utility>

#include "dn_internal.h"

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


// dn_winograd8.hip: the 8-wave / one-block-per-CU form of the three-piece kernel (64 tiles x 64 output channels, two positions per wave)
bool wino8_wanted(const IgemmParams& p);
int launch_wino_conv8(const IgemmParams& p, hipStream_t stream);

}  // namespace dn
