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

}  // namespace dn

using namespace dn;

extern "C" {

int dn_ubench_copy(const float* src, float* dst, int64_t n, dn_stream_t stream) {
  DN_REQUIRE(src && dst && n > 0 && n % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, DN_ERR_BAD_ARG,
             "dn_ubench_copy: need 16-byte aligned buffers and n %% 4 == 0");
  static const int blocks = getenv("DN_UBENCH_COPY_BLOCKS") ? atoi(getenv("DN_UBENCH_COPY_BLOCKS")) : 256 * 16;
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

int dn_xcd_probe(int32_t* out, int32_t gx, int32_t gy, int32_t gz, dn_stream_t stream) {
  DN_REQUIRE(out && gx > 0 && gy > 0 && gz > 0 && gy <= 65535 && gz <= 65535, DN_ERR_BAD_ARG, "dn_xcd_probe: bad argument");   // out: gx*gy*gz ints
  DN_LAUNCH(xcd_probe_kernel, dim3(gx, gy, gz), dim3(64), 0, as_stream(stream), out);
  return check_launch("xcd_probe_kernel");
}

}  // extern "C"
