// Microbenchmark (round 6): the main loop of the three-piece Winograd kernels without its memory traffic -- 48 v_mfma_f32_32x32x16_bf16 per
// "chunk" and wave on 8 accumulators, the A operand of every 12-instruction unit split on the fly into three bf16 pieces (4 pairs per lane
// and unit), S extra independent vector instructions per slot standing in for the staging arithmetic.  What it varies:
//   PK    the subtraction of the split as v_pk_add_f32 (what hipcc emits for the float2 form) or as two scalar v_sub_f32
//   ILP   split chains worked on side by side in one slot (1 = one pair per slot in four slots, as shipped; 2; 4)
//   XI    consecutive matrix instructions on different accumulators (the two cout halves of a unit interleaved) or six in a row on one
//   W     waves per SIMD (1: 256-thread block, 2: 512-thread block); one block per CU, 256 blocks
// Reports clock64 ticks per chunk for wave 0 and the wall time.  build: hipcc --offload-arch=gfx950 -O3 -o split_issue split_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <bool PK, int ILP, bool XI, int S, int TPB, bool NOSPLIT>
__global__ void __launch_bounds__(TPB) chunk_loop(float* out, long long* t, const float* in, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 bq[6];
#pragma unroll
  for (int g = 0; g < 6; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) bq[g][e] = (__bf16)(in[(threadIdx.x + 64 * g + e) & 1023]);
  float raw[4][8];                   // the four units' operands (the real kernel reads them from LDS)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[a][e] = in[(threadIdx.x * 8 + 37 * a + e) & 1023];
  float g[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) g[u] = in[(threadIdx.x + u) & 1023];
  bf16x8 fa3[2][3];
  auto split_pair = [&](int slot, int a, int q) __attribute__((always_inline)) {
    const f32x2 x = f32x2{raw[a][2 * q], raw[a][2 * q + 1]};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    if constexpr (NOSPLIT) {
      fa3[slot][0][2 * q] = h[0]; fa3[slot][0][2 * q + 1] = h[1];
      fa3[slot][1][2 * q] = h[1]; fa3[slot][1][2 * q + 1] = h[0];
      fa3[slot][2][2 * q] = h[0]; fa3[slot][2][2 * q + 1] = h[0];
      return;
    }
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    f32x2 r1, r2;
    if constexpr (PK) r1 = x - hf; else r1 = f32x2{s_sub(x[0], hf[0]), s_sub(x[1], hf[1])};
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 mf = __builtin_convertvector(m, f32x2);
    if constexpr (PK) r2 = r1 - mf; else r2 = f32x2{s_sub(r1[0], mf[0]), s_sub(r1[1], mf[1])};
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    fa3[slot][0][2 * q] = h[0]; fa3[slot][0][2 * q + 1] = h[1];
    fa3[slot][1][2 * q] = m[0]; fa3[slot][1][2 * q + 1] = m[1];
    fa3[slot][2][2 * q] = l[0]; fa3[slot][2][2 * q + 1] = l[1];
  };
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(raw[a][e]));     // (fresh operands every chunk: nothing is hoisted)
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(0, 0, q);
    __builtin_amdgcn_sched_barrier(0);
    static_for<48>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int a = m / 12, q12 = m % 12, nn = XI ? q12 % 2 : q12 / 6, tt = XI ? q12 / 2 : q12 % 6;
      constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
      acc[2 * a + nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[a & 1][AS[tt]], bq[3 * nn + 2 - BS[tt]], acc[2 * a + nn], 0, 0, 0);
      if constexpr (a < 3) {
        constexpr int first = 6, per = ILP, nslots = 4 / ILP;
        if constexpr (q12 >= first && q12 < first + nslots) {
#pragma unroll
          for (int u = 0; u < per; ++u) split_pair((a + 1) & 1, a + 1, (q12 - first) * per + u);
        }
      }
#pragma unroll
      for (int u = 0; u < S; ++u) g[(m * S + u) & 7] = s_add(g[(m * S + u) & 7], g[(m * S + u + 3) & 7]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += g[u];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

static float* d_out;
static long long* d_t;
static float* d_in;

template <bool PK, int ILP, bool XI, int S, int TPB, bool NOSPLIT = false>
void run(const char* name) {
  const int blocks = 256, iters = 400;
  auto k = chunk_loop<PK, ILP, XI, S, TPB, NOSPLIT>;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(TPB), 0, 0, d_out, d_t, d_in, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(TPB), 0, 0, d_out, d_t, d_in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= blocks;
  const int waves_per_simd = TPB / 256;
  printf("%-60s W=%d  ticks/chunk %7.0f  (per matrix instruction and SIMD %5.1f)  wall %8.1f us  = %6.1f ns per chunk and SIMD\n", name, waves_per_simd,
         avg / iters, avg / iters / 48 / waves_per_simd, ms * 1e3, ms * 1e6 / iters);
}

#define RUN(PK, ILP, XI, S) \
  run<PK, ILP, XI, S, 256>("split " #PK "=pk ilp" #ILP " xi" #XI " extra" #S); \
  run<PK, ILP, XI, S, 512>("split " #PK "=pk ilp" #ILP " xi" #XI " extra" #S);

int main() {
  hipMalloc(&d_out, 256 * 512 * sizeof(float));
  hipMalloc(&d_t, 256 * sizeof(long long));
  hipMalloc(&d_in, 1024 * sizeof(float));
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.37f + 0.001f * (float)((i * 7919) % 1013);
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  printf("# matrix instructions only (no split, no extras): the floor = 48 x 32 = 1536 ticks per chunk and wave\n");
  run<true, 1, false, 0, 256, true>("no split, six in a row per accumulator");
  run<true, 1, false, 0, 512, true>("no split, six in a row per accumulator");
  run<true, 1, true, 0, 256, true>("no split, accumulators interleaved");
  run<true, 1, true, 0, 512, true>("no split, accumulators interleaved");
  printf("# the split beside them\n");
  RUN(true, 1, false, 0) RUN(false, 1, false, 0) RUN(true, 1, true, 0) RUN(false, 1, true, 0)
  RUN(true, 2, false, 0) RUN(false, 2, false, 0) RUN(true, 2, true, 0) RUN(false, 2, true, 0)
  RUN(true, 4, false, 0) RUN(false, 4, false, 0) RUN(false, 4, true, 0)
  printf("# + 2 independent vector instructions per slot (the staging arithmetic's share)\n");
  RUN(true, 1, false, 2) RUN(false, 1, false, 2) RUN(false, 1, true, 2) RUN(false, 2, true, 2) RUN(false, 4, true, 2)
  printf("# + 4 per slot\n");
  RUN(true, 1, false, 4) RUN(false, 1, false, 4) RUN(false, 2, true, 4)
  return 0;
}
