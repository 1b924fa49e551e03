// Last-arrival epilogues (round 5): small reductions over the blocks of ONE launch, finished by the block that arrives last instead of by
// a launch of their own (dn::bn_finalize_kernel after every BatchNorm layer's convolution, dn::colsum2_finalize_kernel in front of every
// BatchNorm backward: 5-8 us each on the critical path of a 4-image step whose whole forward pass is 1.4 ms).
//
// Visibility across XCDs WITHOUT an agent-scope fence (a fence writes back an L2 full of other blocks' results; round 3 measured that
// it doubles a kernel's time): the per-block partials are written with agent-scope relaxed atomic stores (global_store ... sc1: through
// the XCD's L2), every wave waits for its own stores (s_waitcnt vmcnt(0)) and the block meets at a barrier before ONE thread bumps an
// agent-scope counter; the block that reads `expected - 1` from it reads every block's partials with agent-scope relaxed atomic loads
// (global_load ... sc1: past its own XCD's L2).  Probed on the device (dn_last_arrival_probe, tools/exp/last_arrival_probe.py: no stale
// value in 400 launches x 5 grid shapes, buffers reused every launch).  The counters are zero before the first launch and left zero.
//
// Summation ORDER: the same device functions serve the stand-alone kernels (dn_bn_finalize / the BatchNorm-backward entry points, for
// up to kFoldMaxRows partial rows) and the folded epilogues, so a folded reduction is bit-identical to the separate kernel
// (tests/test_gpu_kernels.py::test_folded_reductions_are_bitwise_the_separate_kernels).
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dn_fold.h relies on gfx950 behaviour (write-through sc1 stores + s_waitcnt vmcnt(0), relaxed agent-scope atomics served by the L2): probe before porting (dn_last_arrival_probe)"
#endif
#include "dn_internal.h"

namespace dn {

constexpr int kFoldMaxRows = 128;      // partial rows per channel one block finishes (64 channels x 4 row slices of <= 32 loads per thread)
constexpr int kFoldSlices = 4;         // row slices per channel: thread (channel = tid & 63, slice = tid >> 6), tid < 256
constexpr int kFoldCounters = 64;      // int counters (one per 64-channel slice of the output) at the END of dn_conv_desc.splitk_ws

__device__ __forceinline__ void fold_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool AGENT>
__device__ __forceinline__ float2 fold_load2(const float* p) {
  float2 r;
  if constexpr (AGENT) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.x = __builtin_bit_cast(float, (unsigned)(u & 0xffffffffull));
    r.y = __builtin_bit_cast(float, (unsigned)(u >> 32));
  } else {
    r = *reinterpret_cast<const float2*>(p);
  }
  return r;
}

// Called by every thread of the block after its agent-scope stores.  True in exactly one block of the `expected` that call it with this
// counter: the one that arrives last.  `flag`: an int in LDS.
__device__ __forceinline__ bool fold_last_arrival(int* counter, int expected, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (prev == expected - 1) ? 1 : 0;
    if (prev == expected - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-resetting
  }
  __syncthreads();
  return *flag != 0;
}

// BatchNorm batch statistics of channels [c0, c0 + nlive) from partial[rows][C][2] = (sum, M2 about the row's own mean) of 128-pixel rows
// (the last one ragged), merged about a pivot in fp64 exactly as dn::bn_finalize_kernel does (Chan et al.; see there), in the sliced
// order: thread (channel, slice) walks rows slice, slice + 4, ...; the four slices of a channel are added in slice order.
// All threads of the block call it (tid = threadIdx.x; threads >= 256 only take part in the barriers); red: 3 * 4 * 64 doubles of LDS.
template <bool AGENT>
__device__ __forceinline__ void bn_finalize_sliced(const float* partial, int rows, int C, int c0, int nlive, double count, const BnFinalizeArgs& a,
                                                   double* red, int tid) {
  constexpr double kTile = 128.0;
  const int cl = tid & 63, sl = tid >> 6;
  const int c = c0 + cl;
  const bool live = tid < 64 * kFoldSlices && cl < nlive;
  double pivot = 0.0;
  if (live) {
    double s1 = 0.0, q = 0.0, pp = 0.0;
    const double n0 = count < kTile ? count : kTile;
    pivot = (double)fold_load2<AGENT>(partial + (long long)c * 2).x / n0;
#pragma unroll 8
    for (int r = sl; r < rows; r += kFoldSlices) {
      const float2 v = fold_load2<AGENT>(partial + ((long long)r * C + c) * 2);
      const double left = count - (double)r * kTile;
      const double nt = left < kTile ? left : kTile;
      const double dm = (double)v.x / nt - pivot;
      s1 += (double)v.x;
      q += (double)v.y;
      pp += nt * dm * dm;
    }
    red[(0 * kFoldSlices + sl) * 64 + cl] = s1;
    red[(1 * kFoldSlices + sl) * 64 + cl] = q;
    red[(2 * kFoldSlices + sl) * 64 + cl] = pp;
  }
  __syncthreads();
  if (live && sl == 0) {
    double t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      t[k] = ((red[(k * kFoldSlices + 0) * 64 + cl] + red[(k * kFoldSlices + 1) * 64 + cl]) + red[(k * kFoldSlices + 2) * 64 + cl]) +
             red[(k * kFoldSlices + 3) * 64 + cl];
    const double macc = t[0] / count;
    const double m2tot = t[1] + (t[2] - count * (macc - pivot) * (macc - pivot));
    double var = m2tot / count;   // biased; the conv bias shifts the mean only
    if (var < 0.0) var = 0.0;
    const float mean = (float)(macc + (a.conv_bias ? (double)a.conv_bias[c] : 0.0));
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float sc = a.gamma[c] * invstd;
    a.mean[c] = mean;
    a.invstd[c] = invstd;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - mean * sc;
    if (a.running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
  }
  if (tid == 0 && c0 == 0 && a.num_batches_tracked != nullptr) a.num_batches_tracked[0] += 1;   // nn.BatchNorm2d's step counter
}

// out0[c] = sum_r partial[r][c][stride .. offset], out1[c] = the next float: the two BatchNorm-backward sums of channels [c0, c0 + nlive),
// fp64, in the sliced order above.  red: 2 * 4 * 64 doubles of LDS.
template <bool AGENT>
__device__ __forceinline__ void colsum2_sliced(const float* partial, int rows, int C, int stride, int offset, int c0, int nlive, float* out0,
                                               float* out1, double* red, int tid) {
  const int cl = tid & 63, sl = tid >> 6;
  const int c = c0 + cl;
  const bool live = tid < 64 * kFoldSlices && cl < nlive;
  if (live) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int r = sl; r < rows; r += kFoldSlices) {
      const float2 v = fold_load2<AGENT>(partial + ((long long)r * C + c) * stride + offset);
      s0 += (double)v.x;
      s1 += (double)v.y;
    }
    red[(0 * kFoldSlices + sl) * 64 + cl] = s0;
    red[(1 * kFoldSlices + sl) * 64 + cl] = s1;
  }
  __syncthreads();
  if (live && sl == 0) {
    out0[c] = (float)(((red[(0 * kFoldSlices + 0) * 64 + cl] + red[(0 * kFoldSlices + 1) * 64 + cl]) + red[(0 * kFoldSlices + 2) * 64 + cl]) +
                      red[(0 * kFoldSlices + 3) * 64 + cl]);
    out1[c] = (float)(((red[(1 * kFoldSlices + 0) * 64 + cl] + red[(1 * kFoldSlices + 1) * 64 + cl]) + red[(1 * kFoldSlices + 2) * 64 + cl]) +
                      red[(1 * kFoldSlices + 3) * 64 + cl]);
  }
}

}  // namespace dn
