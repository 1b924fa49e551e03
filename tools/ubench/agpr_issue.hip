// Does the placement of the MFMA accumulators (arch VGPRs vs AccVGPRs) change what a vector instruction between two MFMAs costs?
// build: hipcc --offload-arch=gfx950 -O3 -w -o bin/agpr_issue agpr_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int AG, int NACC, int FILL, int TPB>   // KIND 0: f32 32x32x2, 1: bf16 32x32x16; AG 1: accumulators in AccVGPRs
__global__ void __launch_bounds__(TPB) cadence(float* out, long long* t, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(threadIdx.x * 0.001f + e);
    b[e] = (__bf16)(1.0f + threadIdx.x * 0.002f - e);
  }
  float fa = threadIdx.x * 0.001f, fb = 1.f + threadIdx.x * 0.002f;
  float f1 = 1.0001f, f2 = 0.5f;
  float g[16];
  for (int u = 0; u < 16; ++u) g[u] = threadIdx.x * 0.01f * u;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 128 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (AG) {
          if (KIND == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(fa), "v"(fb));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        } else {
          if (KIND == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int u = 0; u < FILL; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[u]) : "v"(f1), "v"(f2));
      }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int u = 0; u < 16; ++u) s += g[u];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int KIND, int AG, int NACC, int FILL, int TPB>
void run(const char* name) {
  const int blocks = 256;
  float* out;
  long long* t;
  hipMalloc(&out, blocks * TPB * 4);
  hipMalloc(&t, blocks * 8);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((cadence<KIND, AG, NACC, FILL, TPB>), dim3(blocks), dim3(TPB), 0, 0, out, t, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((cadence<KIND, AG, NACC, FILL, TPB>), dim3(blocks), dim3(TPB), 0, 0, out, t, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long* h = new long long[blocks];
  hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double av = 0;
  for (int i = 0; i < blocks; ++i) av += h[i];
  av /= blocks;
  printf("%-8s acc in %-8s %d accumulators, %d waves/SIMD, %2d VALU/slot: ticks/MFMA %.1f | wall %.1f us\n", KIND ? "bf16x16" : "f32x2", AG ? "AccVGPR" : "VGPR", NACC, TPB / 256, FILL,
         av / (iters * 128.0), ms * 1e3);
  hipFree(out);
  hipFree(t);
  delete[] h;
}

template <int KIND, int AG>
void sweep() {
  run<KIND, AG, 8, 0, 256>("");
  run<KIND, AG, 8, 4, 256>("");
  run<KIND, AG, 8, 8, 256>("");
  run<KIND, AG, 8, 16, 256>("");
  run<KIND, AG, 4, 0, 512>("");
  run<KIND, AG, 4, 4, 512>("");
  run<KIND, AG, 4, 8, 512>("");
  run<KIND, AG, 4, 16, 512>("");
}

int main() {
  // dependent chains: one / two accumulators per wave
  run<1, 0, 1, 0, 256>("");
  run<1, 0, 2, 0, 256>("");
  run<1, 0, 1, 4, 256>("");
  run<1, 0, 1, 0, 512>("");
  run<1, 0, 2, 0, 512>("");
  run<0, 0, 1, 0, 256>("");
  sweep<0, 0>();
  sweep<0, 1>();
  sweep<1, 0>();
  sweep<1, 1>();
  return 0;
}
