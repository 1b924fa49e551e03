// Microbenchmark (round 6): a MODEL of the three-piece Winograd forward kernel's 16-channel chunk WITH its memory traffic, to ask what the
// block shape is worth before a kernel is written for it.  Per CU and chunk every variant does the same work as dn::wino_conv8_kernel: 384
// v_mfma_f32_32x32x16_bf16 (64 tiles x 64 couts x 16 positions, six partial products), 96 KB of weight fragments straight into registers
// (global_load_dwordx4, an L2-resident 24 MB stream), 64 KB of patches (buffer-style 8- or 4-byte loads), the transform's adds, 64 KB of LDS
// stores, 64 KB of LDS fragment reads (ds_read_b128), the complete split of the A operand, one barrier.  What varies is how it is dealt out:
//   W2: 8 waves x 256 registers (two per SIMD), a wave = 2 positions: 48 matrix instructions, 12 weight loads, 16 patch loads (8 B), 16
//       ds_write_b64, 8 ds_read_b128 per chunk -- the shipped shape (expect ~5.3 k cycles per chunk if the model is faithful);
//   W4: 16 waves x 128 registers (four per SIMD), a wave = 1 position: 24 matrix instructions on 4 accumulators, 6 weight loads, 16 patch
//       loads of 4 B (a thread stages ONE channel of a tile), 16 ds_write_b32, 4 ds_read_b128.
// Results are meaningless numbers; only the time counts.  build: hipcc --offload-arch=gfx950 -O3 -o chunk_model chunk_model.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int kChunkLds = 64 * 16 * 16 * 4;      // one chunk's transformed tile: 64 tiles x 16 positions x 16 channels fp32 = 64 KB

// NW = waves per block (8: W2, 16: W4).  NPOS = positions per wave (2 / 1).
template <int NW>
__global__ void __launch_bounds__(NW * 64, 1) chunk_model(float* out, long long* t, const float* __restrict__ patches, const bf16x8* __restrict__ weights,
                                                       int nchunks, size_t wstride16) {
  constexpr int NPOS = 16 / NW, UNITS = 2 * NPOS, NMF = 12 * UNITS;        // units = (position, tile half); 12 matrix instructions each
  constexpr int VW = NW == 8 ? 2 : 1;                                        // channels a staging thread handles
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[UNITS][2];
#pragma unroll
  for (int u = 0; u < UNITS; ++u)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[u][n][e] = 0.f;
  // weight stream of this wave: NPOS positions x 2 cout halves x 3 pieces per chunk, 1 KB each; ring of six
  // (every block streams the SAME 96 KB per chunk, as the 32 CUs of an XCD do with one cout slice: served by the L2)
  const bf16x8* wp = weights + (size_t)wave * NPOS * 6 * 64 + lane;
  bf16x8 bq[6];
#pragma unroll
  for (int g = 0; g < 6; ++g) bq[g] = wp[g * 64];
  // staging role: a 4x4 patch of VW channels per thread (64 tiles x 16/VW channel groups = 64 NW threads)
  const int st_tile = tid / (16 / VW), cg = tid % (16 / VW);
  // 16 KB of distinct input per block and chunk (the 64 KB of patches overlap four-fold): 4 distinct lines per thread, each fetched 4 times
  const float* pbase = patches + (size_t)blockIdx.x * nchunks * 4096 + st_tile * 64 + cg * VW;
  // LDS position plane (4 KB) = 4 channel quads x [64 tiles][4 channels]: conflict-free b128 fragment reads, as in the kernel
  const int stA = ((cg * VW) >> 2) * 1024 + st_tile * 16 + ((cg * VW) & 3) * 4;
  float v[16][VW];
  bf16x8 fa3[2][3];
  f32x4 raw[2];
  auto split_pair = [&](int slot, int q) __attribute__((always_inline)) {
    const f32x2 x = f32x2{raw[q >> 1][2 * (q & 1)], raw[q >> 1][2 * (q & 1) + 1]};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    fa3[slot][0][2 * q] = h[0]; fa3[slot][0][2 * q + 1] = h[1];
    fa3[slot][1][2 * q] = m[0]; fa3[slot][1][2 * q + 1] = m[1];
    fa3[slot][2][2 * q] = l[0]; fa3[slot][2][2 * q + 1] = l[1];
  };
  // LDS: [buffer][position][tile 64][16 channels] fp32 (4 KB per position plane, 64 KB per buffer); fragment read: lane (tile & 31, k group)
  const int frA = wave * NPOS * 4096 + (lane >> 5) * 2048 + (lane & 31) * 16;
  auto read_raw = [&](const char* Ab, int u) __attribute__((always_inline)) {
    raw[0] = *reinterpret_cast<const f32x4*>(Ab + (u >> 1) * 4096 + (u & 1) * 512);
    raw[1] = *reinterpret_cast<const f32x4*>(Ab + (u >> 1) * 4096 + (u & 1) * 512 + 1024);
  };
  // fill chunk 0
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < VW; ++e) v[i][e] = pbase[(i & 3) * 16 + e];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if constexpr (VW == 2) *reinterpret_cast<f32x2*>(smemB + i * 4096 + stA) = f32x2{v[i][0], v[i][1]};
    else *reinterpret_cast<float*>(smemB + i * 4096 + stA) = v[i][0];
  }
  __syncthreads();
  int buf = 0;
  const long long t0 = clock64();
  for (int c = 0; c < nchunks; ++c) {
    const char* Ab = smemB + buf * kChunkLds + frA;
    const float* pnext = pbase + (size_t)(c + 1 < nchunks ? c + 1 : c) * 4096;
    read_raw(Ab, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(0, q);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NMF>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int u = m / 12, q12 = m % 12, nn = q12 / 6, tt = q12 % 6, posl = u >> 1, mm = u & 1;
      constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
      acc[u][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[u & 1][AS[tt]], bq[3 * nn + 2 - BS[tt]], acc[u][nn], 0, 0, 0);
      if constexpr (mm == 1) {                              // the position's fragments are released: the next position's (or chunk's) come in
        // next position of this chunk, or the first one of the next chunk (chunk stride = 96 fragments of 1 KB = wstride16 x 16 B)
        constexpr int nextp = posl + 1 < NPOS ? posl + 1 : 0;
        const size_t cn = (size_t)(posl + 1 < NPOS ? c : c + 1) * wstride16;
        if constexpr (tt == 0) bq[3 * nn + 0] = wp[cn + (nextp * 6 + 3 * nn + 0) * 64];
        if constexpr (tt == 2) bq[3 * nn + 1] = wp[cn + (nextp * 6 + 3 * nn + 1) * 64];
        if constexpr (tt == 5) bq[3 * nn + 2] = wp[cn + (nextp * 6 + 3 * nn + 2) * 64];
      }
      if constexpr (u + 1 < UNITS && q12 == 1) read_raw(Ab, u + 1);
      if constexpr (u + 1 < UNITS && q12 >= 6 && q12 < 10) split_pair((u + 1) & 1, q12 - 6);
      // staging of the next chunk spread over the chunk: loads in the first two thirds, transform + stores in the last third
      constexpr int L0 = NW == 8 ? 1 : 0, LSTEP = NW == 8 ? 2 : 1, T0 = NMF - 8, TSTEP = 1;      // W2: loads on slots 1, 3 .. 31, stores 40 .. 47; W4: 0 .. 15, 16 .. 23
      if constexpr (m >= L0 && (m - L0) % LSTEP == 0 && (m - L0) / LSTEP < 16) {
        constexpr int i = (m - L0) / LSTEP;
#pragma unroll
        for (int e = 0; e < VW; ++e) v[i][e] = pnext[(i & 3) * 16 + e];
      }
      if constexpr (m >= T0 && (m - T0) % TSTEP == 0 && (m - T0) / TSTEP < 8) {
        constexpr int j = (m - T0) / TSTEP;               // two transform rows' worth of adds + two stores per step (16 stores per chunk)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          constexpr int dummy = 0;
          const int i = 2 * j + h2;
          float o[VW];
#pragma unroll
          for (int e = 0; e < VW; ++e) o[e] = (v[i][e] - v[(i + 8) & 15][e]) + (v[(i + 4) & 15][e] - v[(i + 12) & 15][e]);      // 3 adds per value: row + column transform share
          if constexpr (VW == 2) *reinterpret_cast<f32x2*>(smemB + (buf ^ 1) * kChunkLds + i * 4096 + stA) = f32x2{o[0], o[1]};
          else *reinterpret_cast<float*>(smemB + (buf ^ 1) * kChunkLds + i * 4096 + stA) = o[0];
          (void)dummy;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
    buf ^= 1;
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < UNITS; ++u)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[u][n][e];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) t[blockIdx.x] = t1 - t0;
}

template <int NW>
void run(const char* name, float* d_out, long long* d_t, const float* d_p, const bf16x8* d_w) {
  const int blocks = 256, nchunks = 32;          // (32 chunks x 96 KB = 3 MB of weights: inside one XCD's L2, like a K = 512 layer's slice)
  auto k = chunk_model<NW>;
  const size_t lds = 2 * kChunkLds;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), lds, 0, d_out, d_t, d_p, d_w, nchunks, (size_t)96 * 64);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= blocks;
  printf("%-44s %2d waves/CU  ticks per chunk %7.0f   wall %8.1f us = %7.1f ns per chunk (matrix-only floor 3072 cycles = ~1750 ns at 1.75 GHz)  [%s]\n", name, NW,
         avg / nchunks, ms * 1e3, ms * 1e6 / nchunks, hipGetErrorString(hipGetLastError()));
}

int main() {
  float *d_out, *d_p;
  long long* d_t;
  bf16x8* d_w;
  hipMalloc(&d_out, (size_t)256 * 1024 * sizeof(float));
  hipMalloc(&d_t, 256 * sizeof(long long));
  const size_t pbytes = (size_t)256 * 33 * 4096 * sizeof(float) + (1 << 20), wbytes = (size_t)34 * 96 * 1024 + (1 << 20);
  hipMalloc(&d_p, pbytes);      // 16 KB of distinct "input" per block and chunk
  hipMalloc(&d_w, wbytes);
  hipMemset(d_p, 0, pbytes);
  hipMemset(d_w, 0, wbytes);
  run<8>("W2: 8 waves x 2 positions (the shipped shape)", d_out, d_t, d_p, d_w);
  run<16>("W4: 16 waves x 1 position", d_out, d_t, d_p, d_w);
  run<8>("W2 again", d_out, d_t, d_p, d_w);
  run<16>("W4 again", d_out, d_t, d_p, d_w);
  return 0;
}
