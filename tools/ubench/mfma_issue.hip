// Microbenchmark: issue cadence of v_mfma_f32_32x32x2_f32 from ONE wave per SIMD vs TWO (cycles per MFMA, clock64).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip ; run: ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int FILL, int TPB = 256>
__global__ void __launch_bounds__(TPB) k(float* out, long long* t, int iters) {
  extern __shared__ float smem[];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float f0 = a, f1 = b, f2 = a + b, f3 = a - b;
  float g[16];
  for (int u = 0; u < 16; ++u) g[u] = a * u;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 128 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (FILL == 10) {   // 8 INDEPENDENT VALU ops
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = fmaf(g[u], f1, f2);
        }
        if (FILL == 11) {   // 16 independent
#pragma unroll
          for (int u = 0; u < 16; ++u) g[u] = fmaf(g[u], f1, f2);
        }
        if (FILL == 12) {   // 4 independent
#pragma unroll
          for (int u = 0; u < 4; ++u) g[u] = fmaf(g[u], f1, f2);
        }
        if (FILL == 13) {   // 12 independent
#pragma unroll
          for (int u = 0; u < 12; ++u) g[u] = fmaf(g[u], f1, f2);
        }
        if (FILL == 20) {   // 2 LDS reads (b128) + 4 independent VALU
          f32x4 l0 = *(const f32x4*)(smem + threadIdx.x * 4 + (r * NACC + i) * 8);
          f32x4 l1 = *(const f32x4*)(smem + 2048 + threadIdx.x * 4 + (r * NACC + i) * 8);
          g[0] += l0[0]; g[1] += l1[1]; g[2] = fmaf(g[2], f1, f2); g[3] = fmaf(g[3], f1, f2);
        }
        if (FILL >= 1 && FILL < 10) { f0 = fmaf(f0, f1, f2); f1 = fmaxf(f1, f3); f2 = fmaf(f2, f3, f0); f3 = fmaxf(f3, f0); }
        if (FILL >= 2 && FILL < 10) { f0 = fmaf(f0, f1, f2); f1 = fmaxf(f1, f3); f2 = fmaf(f2, f3, f0); f3 = fmaxf(f3, f0); }
        if (FILL >= 3 && FILL < 10) { f0 = fmaf(f0, f1, f2); f1 = fmaxf(f1, f3); f2 = fmaf(f2, f3, f0); f3 = fmaxf(f3, f0); f0 = fmaf(f0, f1, f2); f1 = fmaxf(f1, f3); f2 = fmaf(f2, f3, f0); f3 = fmaxf(f3, f0);}
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  long long t1 = clock64();
  float s = f0 + f1 + f2 + f3;
  for (int u = 0; u < 16; ++u) s += g[u];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int NACC, int FILL, int TPB = 256>
void run(const char* name, int threads, size_t lds, int blocks) {
  float* out; long long* t;
  hipMalloc(&out, blocks * threads * 4); hipMalloc(&t, blocks * 8);
  hipFuncSetAttribute((const void*)k<NACC, FILL, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, FILL, TPB>), dim3(blocks), dim3(threads), lds, 0, out, t, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, FILL, TPB>), dim3(blocks), dim3(threads), lds, 0, out, t, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long* h = new long long[blocks]; hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double mn = 1e30, mx = 0, av = 0;
  for (int i = 0; i < blocks; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; av += h[i]; }
  av /= blocks;
  printf("%-58s ticks/MFMA min %.1f avg %.1f max %.1f | wall %.1f us = %.1f ns/MFMA/wave\n", name, mn / (iters * 128.0), av / (iters * 128.0),
         mx / (iters * 128.0), ms * 1e3, ms * 1e6 / (iters * 128.0));
  hipFree(out); hipFree(t); delete[] h;
}
int main() {
  run<8, 0>("1 wave/SIMD, 8 acc, no filler", 256, 128 * 1024, 256);
  run<8, 1>("1 wave/SIMD, 8 acc, 4 VALU/slot", 256, 128 * 1024, 256);
  run<8, 2>("1 wave/SIMD, 8 acc, 8 VALU/slot", 256, 128 * 1024, 256);
  run<8, 3>("1 wave/SIMD, 8 acc, 16 VALU/slot", 256, 128 * 1024, 256);
  run<16, 0>("1 wave/SIMD, 16 acc, no filler", 256, 128 * 1024, 256);
  run<16, 2>("1 wave/SIMD, 16 acc, 8 VALU/slot", 256, 128 * 1024, 256);
  run<8, 12>("1 wave/SIMD, 8 acc, 4 independent VALU/slot", 256, 128 * 1024, 256);
  run<8, 10>("1 wave/SIMD, 8 acc, 8 independent VALU/slot", 256, 128 * 1024, 256);
  run<8, 13>("1 wave/SIMD, 8 acc, 12 independent VALU/slot", 256, 128 * 1024, 256);
  run<8, 11>("1 wave/SIMD, 8 acc, 16 independent VALU/slot", 256, 128 * 1024, 256);
  run<8, 20>("1 wave/SIMD, 8 acc, 2 ds_read_b128 + 4 VALU/slot", 256, 128 * 1024, 256);
  run<4, 0, 512>("2 waves/SIMD (512-thread block), 4 acc, no filler", 512, 128 * 1024, 256);
  run<4, 12, 512>("2 waves/SIMD (512-thread block), 4 acc, 4 indep VALU/slot", 512, 128 * 1024, 256);
  run<4, 10, 512>("2 waves/SIMD (512-thread block), 4 acc, 8 indep VALU/slot", 512, 128 * 1024, 256);
  run<4, 11, 512>("2 waves/SIMD (512-thread block), 4 acc, 16 indep VALU/slot", 512, 128 * 1024, 256);
  run<4, 20, 512>("2 waves/SIMD (512-thread block), 4 acc, 2 ds_read_b128 + 4 VALU", 512, 128 * 1024, 256);
  return 0;
}
