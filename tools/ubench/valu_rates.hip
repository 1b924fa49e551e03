// Issue cost (cycles per wave64 instruction, one wave per SIMD and two) of the vector instructions the f32x3 kernels lean on.
// build: hipcc --offload-arch=gfx950 -O3 -w -o bin/valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND, int TPB>
__global__ void __launch_bounds__(TPB) k(float* out, long long* t, int iters) {
  float a[8], b[8];
  unsigned u[8], w[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.01f + i; b[i] = 1.5f + i; u[i] = threadIdx.x * 7 + i; w[i] = threadIdx.x * 3 + i; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
        if (KIND == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(w[i]));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&b[i & 6]));
        if (KIND == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(w[i]));
        if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(w[i]));
        if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(w[i]));
        if (KIND == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(w[i]), "v"(u[(i + 1) & 7]), "v"(w[(i + 2) & 7]));
        if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
      }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + u[i] + w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int KIND, int TPB>
void run(const char* name) {
  const int blocks = 256, iters = 500;
  float* out; long long* t;
  hipMalloc(&out, blocks * TPB * 4); hipMalloc(&t, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, TPB>), dim3(blocks), dim3(TPB), 0, 0, out, t, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KIND, TPB>), dim3(blocks), dim3(TPB), 0, 0, out, t, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double av = 0; for (int i = 0; i < blocks; ++i) av += h[i]; av /= blocks;
  const double n = iters * 64.0;
  printf("%-24s %d wave(s)/SIMD: %.2f ticks/instr (wave 0), wall %.1f us = %.2f ns per instr per SIMD\n", name, TPB / 256, av / n, ms * 1e3, ms * 1e6 / (n * (TPB / 256)));
  hipFree(out); hipFree(t);
}
#define BOTH(K, NAME) run<K, 256>(NAME); run<K, 512>(NAME);
int main() {
  BOTH(0, "v_fma_f32");
  BOTH(1, "v_cvt_pk_bf16_f32");
  BOTH(2, "v_permlane32_swap_b32");
  BOTH(3, "v_pk_add_f32");
  BOTH(4, "v_and_b32");
  BOTH(5, "v_lshlrev_b32");
  BOTH(6, "v_mov_b32");
  BOTH(7, "v_perm_b32");
  BOTH(8, "v_med3_f32");
  return 0;
}
