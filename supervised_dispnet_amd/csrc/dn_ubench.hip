// Attainable-peak probes behind the C ABI (SURVEY.md section 8d "Peaks"): bench.py divides its HBM-bound and MFMA-bound
// kernels by what THIS box sustains, next to the vendor figures.
//   dn_ubench_copy      float4 streaming copy (read n floats, write n floats): 16 KiB tiles, 4 loads in flight per lane
//   dn_ubench_mfma_f32  register-resident v_mfma_f32_32x32x2_f32 loop: 4 waves per CU-slot, 4 independent accumulators per
//                       wave (64-cycle dependent latency = 64-cycle issue, so one accumulator would already pace the pipe)
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one 16 KiB tile per block iteration: 4 float4 per thread, 256-thread-contiguous rows (4 KiB per load instruction per block)
__global__ void __launch_bounds__(256) ubench_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n4) {
  const long long ntiles = n4 / 1024;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long i = t * 1024 + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[i + u * 256] = v[u];
  }
  for (long long i = ntiles * 1024 + (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) ubench_mfma_kernel(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

// store-only patterns: mode 0 = every instruction writes 1 KiB contiguous (lane l: 16 bytes at 16 l); mode 1 = the first layer's pattern
// before round 4's store shuffle: four instructions per 16-pixel tile, instruction m gives pixel j (256-byte rows) its bytes 64 m .. 64 m + 63
// (lane (j, g): 16 bytes at 256 j + 64 m + 16 g); mode 2 = 16 pixels x 64 bytes contiguous rows (a 16-channel layer).  n4 = float4 count.
__global__ void __launch_bounds__(256) ubench_store_kernel(f32x4* __restrict__ dst, long long n4, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const f32x4 v = f32x4{1.f, 2.f, 3.f, (float)lane};
  const long long ntiles = n4 / 256;                      // a "tile" = 16 pixels x 256 bytes = 4 KiB = 256 float4
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += (long long)gridDim.x * 4) {
    f32x4* base = dst + t * 256;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (mode == 1) base[j * 16 + m * 4 + g] = v;
      else base[m * 64 + lane] = v;
    }
  }
}

// one int per block: the XCD (HW_REG_XCC_ID, 0..7) the block was dispatched to, at its linear block index
__global__ void __launch_bounds__(64) xcd_probe_kernel(int* __restrict__ out) {
  if (threadIdx.x == 0) {
    const unsigned id = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 15u;
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)id;
  }
}

// Cross-XCD last-arrival WITHOUT an agent-scope fence (round 5 probe for reductions finished by the last block of a grid): every block
// writes a record with agent-scope (sc1) relaxed atomic stores, waits for them (vmcnt(0)), bumps an agent-scope counter; the block that
// arrives last reads EVERY block's record with agent-scope relaxed atomic loads and counts mismatches.  The same buffers are reused round
// after round, so another XCD's L2 holds the previous round's lines -- exactly what a plain load would return.  mode 1 = plain stores /
// plain loads (the control: expected to see stale data across XCDs).  result[0] += mismatches, result[1] = last block, result[2] += 1.
__global__ void __launch_bounds__(256) last_arrival_probe_kernel(float* __restrict__ data, int* __restrict__ counter, int* __restrict__ result,
                                                                 int rec, int round, int mode) {
  __shared__ int last;
  const int tid = threadIdx.x;
  float* mine = data + (long long)blockIdx.x * rec;
  // (blocks arrive in a scrambled order: a block-dependent number of dependent FMAs first)
  float spin = 1.f + tid;
  for (int i = 0; i < (int)((blockIdx.x * 37u) % 64u) * 16; ++i) spin = __builtin_fmaf(spin, 1.0000001f, 1e-7f);
  const float tag = (float)(round % 1000) * 1000.f + (float)blockIdx.x + (spin < 0.f ? 1.f : 0.f);
  for (int i = tid; i < rec; i += 256) {
    const float v = tag + (float)i * 0.0009765625f;
    if (mode == 0) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else mine[i] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) last = (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  int bad = 0;
  for (int b = 0; b < (int)gridDim.x; ++b) {
    const float want0 = (float)(round % 1000) * 1000.f + (float)b;
    const float* rp = data + (long long)b * rec;
    for (int i = tid; i < rec; i += 256) {
      const float v = mode == 0 ? __hip_atomic_load(rp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *reinterpret_cast<const volatile float*>(rp + i);
      bad += (v != want0 + (float)i * 0.0009765625f) ? 1 : 0;
    }
  }
  if (bad) atomicAdd(result, bad);
  if (tid == 0) {
    result[1] = (int)blockIdx.x;
    atomicAdd(result + 2, 1);
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // self-resetting, like the K-split counters
  }
}

}  // namespace dn

using namespace dn;

extern "C" {

int dn_ubench_copy(const float* src, float* dst, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(src && dst && n > 0 && n % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, DN_ERR_BAD_ARG,
             "dn_ubench_copy: need 16-byte aligned buffers and n %% 4 == 0");
  const int blocks = 256 * 16;
  DN_LAUNCH(ubench_copy_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const f32x4*)src, (f32x4*)dst, (long long)(n / 4));
  return check_launch("ubench_copy_kernel");
}

int64_t dn_ubench_mfma_f32_flops(int32_t blocks, int32_t iters) {
  // per wave and iteration: 32 MFMAs of 32x32x2 = 2*32*32*2 flops each; 4 waves per block
  return (int64_t)blocks * 4 * (int64_t)iters * 32 * 4096;
}

int dn_ubench_mfma_f32(float* out, int32_t blocks, int32_t iters, dn_stream_t stream) {
  DN_REQUIRE(out && blocks > 0 && iters > 0, DN_ERR_BAD_ARG, "dn_ubench_mfma_f32: bad argument");   // out: blocks * 256 floats
  DN_LAUNCH(ubench_mfma_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), out, iters);
  return check_launch("ubench_mfma_kernel");
}

int dn_ubench_store(float* dst, int64_t n, int32_t mode, dn_stream_t stream) {
  DN_REQUIRE(dst && n > 0 && n % 1024 == 0 && ((uintptr_t)dst & 15) == 0, DN_ERR_BAD_ARG, "dn_ubench_store: need a 16-byte aligned buffer of a multiple of 1024 floats");
  DN_LAUNCH(ubench_store_kernel, dim3(256 * 8), dim3(256), 0, as_stream(stream), (f32x4*)dst, (long long)(n / 4), mode);
  return check_launch("ubench_store_kernel");
}

int dn_last_arrival_probe(float* data, int32_t* counter, int32_t* result, int32_t blocks, int32_t rec, int32_t round, int32_t mode, dn_stream_t stream) {
  DN_REQUIRE(data && counter && result && blocks > 0 && rec > 0, DN_ERR_BAD_ARG, "dn_last_arrival_probe: bad argument");   // data: blocks*rec floats; counter: 1 zeroed int; result: 3 ints
  DN_LAUNCH(last_arrival_probe_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), data, counter, result, rec, round, mode);
  return check_launch("last_arrival_probe_kernel");
}

int dn_xcd_probe(int32_t* out, int32_t gx, int32_t gy, int32_t gz, dn_stream_t stream) {
  DN_REQUIRE(out && gx > 0 && gy > 0 && gz > 0 && gy <= 65535 && gz <= 65535, DN_ERR_BAD_ARG, "dn_xcd_probe: bad argument");   // out: gx*gy*gz ints
  DN_LAUNCH(xcd_probe_kernel, dim3(gx, gy, gz), dim3(64), 0, as_stream(stream), out);
  return check_launch("xcd_probe_kernel");
}

}  // extern "C"
