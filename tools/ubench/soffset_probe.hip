// Probe: raw buffer loads with a scalar offset -- address and range check (gfx950).  hipcc --offload-arch=gfx950 tools/ubench/soffset_probe.hip -o tools/ubench/bin/soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* src, float* out, int sB) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x80000000u, 0x00020000);
  const int lane = threadIdx.x;
  int voff = lane * 8;
  if (lane & 1) voff |= (int)0x80000000;
  if ((lane & 3) == 2) voff = -1;
  const i32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, sB, 0);
  const i32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
  out[4 * lane + 0] = __builtin_bit_cast(float, a[0]);
  out[4 * lane + 1] = __builtin_bit_cast(float, a[1]);
  out[4 * lane + 2] = __builtin_bit_cast(float, b[0]);
  out[4 * lane + 3] = __builtin_bit_cast(float, b[1]);
}
int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 64 * 16);
  float h[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) h[i] = (float)i;
  hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(src, out, 4096);
  float o[256];
  hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) printf("lane %d: soffset 4096 -> %.0f %.0f   soffset 0 -> %.0f %.0f\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
  return 0;
}
