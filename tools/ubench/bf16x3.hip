// Microbenchmark for the "f32 on the bf16 matrix cores" idea (3-way bf16 split of both operands, 6 of the 9 partial products,
// fp32 accumulation):
//   1. issue cadence of v_mfma_f32_32x32x16_bf16, alone and with N independent vector instructions between two MFMAs, from one and
//      from two waves per SIMD (does vector work hide under the bf16 matrix pipe? -- it does NOT under v_mfma_f32_32x32x2_f32,
//      profiles/r01_mfma_issue_ubench.txt);
//   2. accuracy: C = A B (M = N = 32, K = 16 * KB) by the split scheme, by an fp32 FMA chain (v_mfma_f32_32x32x2_f32) and by plain
//      bf16, each against an fp64 reference on the host.
// build: hipcc --offload-arch=gfx950 -O3 -o bf16x3 bf16x3.hip ; run: ./bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int FILL, int TPB>
__global__ void __launch_bounds__(TPB) cadence(float* out, long long* t, int iters) {
  extern __shared__ float smem[];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(threadIdx.x * 0.001f + e);
    b[e] = (__bf16)(1.0f + threadIdx.x * 0.002f - e);
  }
  float f1 = 1.0001f, f2 = 0.5f;
  float g[16];
  for (int u = 0; u < 16; ++u) g[u] = threadIdx.x * 0.01f * u;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 128 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        if (FILL > 0 && FILL <= 16) {
#pragma unroll
          for (int u = 0; u < FILL; ++u) g[u] = fmaf(g[u], f1, f2);
        }
        if (FILL == 20) {   // 1 ds_read_b128 + 2 VALU
          f32x4 l0 = *(const f32x4*)(smem + threadIdx.x * 4 + (r * NACC + i) * 8);
          g[0] += l0[0];
          g[1] = fmaf(g[1], f1, f2);
        }
        if (FILL == 21) {   // 1 ds_write_b32 + 2 VALU
          smem[threadIdx.x + (r * NACC + i) * 8] = g[0];
          g[0] = fmaf(g[0], f1, f2);
          g[1] = fmaf(g[1], f1, f2);
        }
        __builtin_amdgcn_sched_barrier(0);
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

template <int NACC, int FILL, int TPB>
void run(const char* name, int blocks) {
  float* out;
  long long* t;
  const size_t lds = 64 * 1024;
  hipMalloc(&out, blocks * TPB * 4);
  hipMalloc(&t, blocks * 8);
  hipFuncSetAttribute((const void*)cadence<NACC, FILL, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((cadence<NACC, FILL, TPB>), dim3(blocks), dim3(TPB), lds, 0, out, t, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((cadence<NACC, FILL, TPB>), dim3(blocks), dim3(TPB), lds, 0, out, t, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long* h = new long long[blocks];
  hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
  double av = 0;
  for (int i = 0; i < blocks; ++i) av += h[i];
  av /= blocks;
  const double waves = (double)blocks * TPB / 64, flop = waves * iters * 128.0 * 32 * 32 * 16 * 2;
  printf("%-64s ticks/MFMA %.1f | wall %.1f us = %.1f TFLOP/s bf16\n", name, av / (iters * 128.0), ms * 1e3, flop / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(t);
  delete[] h;
}

// ---------------------------------------------------------------------------------------------- accuracy
// one wave: C[32][32] = A[32][K] B[K][32];  lane l: row / column (l & 31), k group (l >> 5)
__device__ __forceinline__ void split3(float x, __bf16* h, __bf16* m, __bf16* l) {
  const __bf16 a = (__bf16)x;                 // round to nearest even
  const float r1 = x - (float)a;              // exact
  const __bf16 b = (__bf16)r1;
  const float r2 = r1 - (float)b;             // exact, fits 8 significant bits
  *h = a;
  *m = b;
  *l = (__bf16)r2;
}

template <int MODE>   // 0: fp32 FMA chain (32x32x2 f32), 1: bf16 only, 6: six split products, 9: all nine, 3: three (hh, hm, mh)
__global__ void __launch_bounds__(64) gemm_modes(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
  const int lane = threadIdx.x, rc = lane & 31, kg = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rc * K + k + kg], B[(k + kg) * 32 + rc], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a[3], b[3];
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + 8 * kg + e;
        __bf16 h, m, l;
        split3(A[rc * K + k], &h, &m, &l);
        a[0][e] = h; a[1][e] = m; a[2][e] = l;
        split3(B[k * 32 + rc], &h, &m, &l);
        b[0][e] = h; b[1][e] = m; b[2][e] = l;
      }
      if (MODE >= 9) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
      if (MODE >= 9) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
      if (MODE >= 9) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
      if (MODE >= 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      if (MODE >= 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      if (MODE >= 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      if (MODE >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      if (MODE >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
    C[row * 32 + rc] = acc[r];
  }
}

template <int MODE>
void accuracy(const char* name, const float* dA, const float* dB, float* dC, const double* ref, int K, double scale) {
  hipLaunchKernelGGL((gemm_modes<MODE>), dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  float h[1024];
  hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0, l2 = 0, rn = 0;
  for (int i = 0; i < 1024; ++i) {
    const double e = fabs((double)h[i] - ref[i]);
    mx = e > mx ? e : mx;
    l2 += e * e;
    rn += ref[i] * ref[i];
  }
  printf("  %-44s max |err| %.3e (%.2e of the result's magnitude), relative L2 %.3e\n", name, mx, mx / scale, sqrt(l2 / rn));
}

int main() {
  printf("# issue cadence of v_mfma_f32_32x32x16_bf16 (256 blocks; FILL = independent v_fma_f32 between two MFMAs)\n");
  run<8, 0, 256>("1 wave/SIMD, no filler", 256);
  run<8, 2, 256>("1 wave/SIMD, 2 VALU/slot", 256);
  run<8, 4, 256>("1 wave/SIMD, 4 VALU/slot", 256);
  run<8, 8, 256>("1 wave/SIMD, 8 VALU/slot", 256);
  run<8, 16, 256>("1 wave/SIMD, 16 VALU/slot", 256);
  run<8, 20, 256>("1 wave/SIMD, 1 ds_read_b128 + 2 VALU/slot", 256);
  run<8, 21, 256>("1 wave/SIMD, 1 ds_write_b32 + 2 VALU/slot", 256);
  run<4, 0, 512>("2 waves/SIMD, no filler", 256);
  run<4, 2, 512>("2 waves/SIMD, 2 VALU/slot", 256);
  run<4, 4, 512>("2 waves/SIMD, 4 VALU/slot", 256);
  run<4, 8, 512>("2 waves/SIMD, 8 VALU/slot", 256);
  run<4, 16, 512>("2 waves/SIMD, 16 VALU/slot", 256);
  run<4, 20, 512>("2 waves/SIMD, 1 ds_read_b128 + 2 VALU/slot", 256);
  run<4, 21, 512>("2 waves/SIMD, 1 ds_write_b32 + 2 VALU/slot", 256);
  run<2, 0, 1024>("4 waves/SIMD, no filler", 256);
  run<2, 8, 1024>("4 waves/SIMD, 8 VALU/slot", 256);

  for (int variant = 0; variant < 2; ++variant) {
    for (int K : {64, 512, 4608}) {
      float *A = new float[32 * K], *B = new float[K * 32];
      double* ref = new double[1024];
      srand(1234 + K);
      for (int i = 0; i < 32 * K; ++i) {
        const float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX;
        A[i] = variant == 0 ? u : 2.f * u - 1.f;                  // variant 0: non-negative (post-ReLU-like, no cancellation)
        B[i] = (2.f * v - 1.f) * 0.05f;
      }
      double scale = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0;
          for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
          ref[i * 32 + j] = s;
          scale = fabs(s) > scale ? fabs(s) : scale;
        }
      float *dA, *dB, *dC;
      hipMalloc(&dA, 32 * K * 4);
      hipMalloc(&dB, 32 * K * 4);
      hipMalloc(&dC, 4096);
      hipMemcpy(dA, A, 32 * K * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B, 32 * K * 4, hipMemcpyHostToDevice);
      printf("# accuracy vs fp64, K = %d, A %s, |C| max %.3g\n", K, variant == 0 ? "in [0, 1)" : "in (-1, 1)", scale);
      accuracy<0>("fp32 FMA chain (v_mfma_f32_32x32x2_f32)", dA, dB, dC, ref, K, scale);
      accuracy<9>("bf16 x 3 split, 9 products", dA, dB, dC, ref, K, scale);
      accuracy<6>("bf16 x 3 split, 6 products", dA, dB, dC, ref, K, scale);
      accuracy<3>("bf16 x 2 split, 3 products", dA, dB, dC, ref, K, scale);
      accuracy<1>("plain bf16", dA, dB, dC, ref, K, scale);
      hipFree(dA); hipFree(dB); hipFree(dC);
      delete[] A; delete[] B; delete[] ref;
    }
  }
  return 0;
}
