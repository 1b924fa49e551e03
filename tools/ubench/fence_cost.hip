// What a cross-stream fence (hipEventRecord on stream A + hipStreamWaitEvent on stream B) costs stream A's queue: N short kernels back to back
// on A, with and without a fence after each, for plain events (hipEventDisableTiming) and for events without the system-scope fence
// (hipEventDisableSystemFence).  Also: a fence after every 4th kernel, and the kernels 50 us long instead of 3 us.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/bin/fence_cost tools/ubench/fence_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* p, int iters) {
  float x = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

#include <hip/hip_ext.h>
static bool g_ext = false;     // the event as the STOP event of the kernel's own launch (hipExtLaunchKernel): no marker packet behind the kernel

static double run(hipStream_t a, hipStream_t b, float* buf, float* buf2, int n, int iters, int every, unsigned flags, bool side_work, double* host_ms) {
  std::vector<hipEvent_t> evs(n);
  for (auto& e : evs) hipEventCreateWithFlags(&e, flags);
  hipEvent_t t0, t1;
  hipEventCreate(&t0);
  hipEventCreate(&t1);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipDeviceSynchronize();
    auto h0 = std::chrono::steady_clock::now();
    hipEventRecord(t0, a);
    for (int i = 0; i < n; ++i) {
      const bool fence = every > 0 && i % every == every - 1;
      if (fence && g_ext) {
        void* argv[] = {&buf, &iters};
        hipExtLaunchKernel(reinterpret_cast<const void*>(spin_kernel), dim3(64), dim3(256), argv, 0, a, nullptr, evs[i], 0);
      } else {
        spin_kernel<<<64, 256, 0, a>>>(buf, iters);
      }
      if (fence) {
        if (!g_ext) hipEventRecord(evs[i], a);
        hipStreamWaitEvent(b, evs[i], 0);
        if (side_work) spin_kernel<<<64, 256, 0, b>>>(buf2, iters);
      }
    }
    hipStreamWaitEvent(a, t0, 0);   // (no-op ordering; keeps the pattern symmetric)
    hipEventRecord(t1, a);
    auto h1 = std::chrono::steady_clock::now();
    hipEventSynchronize(t1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    if (ms < best) {
      best = ms;
      *host_ms = std::chrono::duration<double, std::milli>(h1 - h0).count();
    }
  }
  for (auto& e : evs) hipEventDestroy(e);
  return best;
}

int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  float *buf, *buf2;
  hipMalloc(&buf, 64 * 256 * 4);
  hipMalloc(&buf2, 64 * 256 * 4);
  hipMemset(buf, 0, 64 * 256 * 4);
  const int n = 200;
  const unsigned plain = hipEventDisableTiming, nosys = hipEventDisableTiming | hipEventDisableSystemFence;
  for (int iters : {200, 6000}) {
    double h;
    const double base = run(a, b, buf, buf2, n, iters, 0, plain, false, &h);
    printf("kernel ~%.1f us: %d kernels back to back %.3f ms (host %.3f ms)\n", base * 1e3 / n, n, base, h);
    for (int every : {1, 4}) {
      for (int sw = 0; sw < 2; ++sw) {
        const double t1 = run(a, b, buf, buf2, n, iters, every, plain, sw, &h);
        const double h1 = h;
        const double t2 = run(a, b, buf, buf2, n, iters, every, nosys, sw, &h);
        g_ext = true;
        const double t3 = run(a, b, buf, buf2, n, iters, every, nosys, sw, &h);
        const double h3 = h;
        g_ext = false;
        printf("  fence every %d kernel(s)%s: event = the launch's stop event (hipExtLaunchKernel) %.3f ms (+%.2f us per fence, host %.3f ms)\n", every,
               sw ? " + a kernel on the waiting stream" : "", t3, (t3 - base) * 1e3 / (n / every), h3);
        printf("  fence every %d kernel(s)%s: plain events %.3f ms (+%.2f us per fence, host %.3f ms), no system fence %.3f ms (+%.2f us per fence, host %.3f ms)\n",
               every, sw ? " + a kernel on the waiting stream" : "", t1, (t1 - base) * 1e3 / (n / every), h1, t2, (t2 - base) * 1e3 / (n / every), h);
      }
    }
  }
  return 0;
}
