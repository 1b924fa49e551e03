// Winograd F(2x2, 3x3) convolution on the CDNA4 fp32 matrix core (gfx950 only): forward and input gradient of the
// 3x3 / stride 1 / pad 1 layers (the whole VGG16-BN encoder but its first layer, and the channel-aligned decoder iconvs;
// models/Disp_vgg_BN.py:84,93-105,137-186).  2.25x fewer multiply-accumulates than the direct contraction, all of them
// exact fp32 FMAs on v_mfma_f32_32x32x2_f32; the transforms are fp32 adds (and multiplications by 1/2 on the weights).
//
//   Y_tile(2x2) = A^T [ sum_c  U_c (.) V_c ] A      V = B^T d B   (4x4 input patch d, per channel, on the fly)
//                                                    U = G g G^T   (3x3 filter g, once per optimizer step: wino_pack_kernel)
//
// = 16 independent GEMMs  M_p[tile][cout] = sum_c V_p[tile][c] * U_p[c][cout],  p = 4i + j the position in the 4x4 transform
// domain.  One block: 64 tiles (= 256 output pixels) x 64 output channels x all 16 positions, 4 waves, ONE wave per SIMD with
// the whole 512-register file: wave w owns the four positions of transform row i = w, i.e. 4 x (64x64) accumulators = 256
// registers.
//   A side : each thread gathers one 4x4 patch of 4 channels straight from the NHWC operands (virtual concat, pending
//            BatchNorm-apply + ReLU of the producer, zero halo), transforms it in registers and writes the 16 transformed
//            float4 into LDS ([position][tile][8 k] rows of 32 B: the per-lane ds_read_b128 fragment reads are contiguous);
//            16-channel chunks, double-buffered, one barrier per chunk.
//   B side : the transformed weights never touch LDS.  They are packed in MFMA fragment order ([k/8][position][cout/32]
//            [lane][4]), so a wave's B fragment is one fully coalesced 1 KiB global load per (position, 32 couts, 8 k), issued
//            one 8-k step ahead of its use; each element is read by exactly one wave of the block.
//   Output : the transform along j is done in registers, the transform along i crosses the four waves through LDS; then the
//            usual epilogue (batch-statistic partials of the pre-bias result per 32 tiles = 128 pixels, bias, activation,
//            channel-split / accumulating stores, float4 along the channels).
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "dn_internal.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WBT = 64;            // tiles per block
constexpr int WBN = 64;            // output channels per block
constexpr int WKC = 16;            // channels per staged chunk (two 8-k MFMA groups)
constexpr int WZLD = 72;           // padded row (floats) of the cross-wave exchange tile: rows 4 apart land 32 banks apart
constexpr size_t kWinoLds = (size_t)2 * 2 * 16 * WBT * 8 * sizeof(float);   // A ring: [buf][sub][pos][tile][8] = 128 KiB

__device__ __forceinline__ float wino_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------------ eligibility
bool wino_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (getenv("DN_NO_WINOGRAD")) return false;
  if (!(d->kind == DN_CONV_FWD || d->kind == DN_CONV_DGRAD)) return false;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0) return false;
  if (d->IH != d->OH || d->IW != d->OW || (d->OH & 1) || (d->OW & 1)) return false;
  if (p.nphases != 1 || p.ph[0].ntaps != 9) return false;
  if (p.Ntot < 64 || (p.Ntot & 3)) return false;
  if ((long long)p.M * 4 >= (1ll << 31)) return false;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!o.vec || !o.small || o.up != 0 || (o.C % WKC) != 0) return false;
    if (o.scale != nullptr && ((reinterpret_cast<uintptr_t>(o.scale) | reinterpret_cast<uintptr_t>(o.shift)) & 15)) return false;
  }
  for (int i = 0; i < p.n_out; ++i) {
    const KResult& r = p.out[i];
    if ((r.C & 3) || (r.sw & 3) || (reinterpret_cast<uintptr_t>(r.p) & 15) || !r.linear) return false;
  }
  if (p.bias != nullptr && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
  return true;
}

static int wino_ktot(const IgemmParams& p) {
  int k = 0;
  for (int i = 0; i < p.n_in; ++i) k += p.in[i].C;
  return k;
}

static int wino_npad(const IgemmParams& p) { return (p.Ntot + WBN - 1) / WBN * WBN; }

long long wino_packed_elems(const IgemmParams& p) { return (long long)wino_ktot(p) * wino_npad(p) * 16; }

// ------------------------------------------------------------------------------------------------ weight transform
// wp[k/8][pos][n/32][lane][e] = U_pos[n][k],  k = 8*(k/8) + 4*(lane >> 5) + e,  n = 32*(n/32) + (lane & 31):  exactly the
// float4 a lane feeds to four consecutive v_mfma_f32_32x32x2_f32 as the B operand (the two half-waves hold k 0-3 / 4-7).
__global__ void wino_pack_kernel(const IgemmParams p, const float* __restrict__ w, float* __restrict__ wp, int NS, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long long rest = idx >> 8;
    const int nsub = (int)(rest % NS);
    rest /= NS;
    const int pos = (int)(rest & 15), kc8 = (int)(rest >> 4);
    const int k = kc8 * 8 + (lane >> 5) * 4 + e, n = nsub * 32 + (lane & 31);
    float u = 0.f;
    if (n < p.Ntot) {
      float g[3][3];
      const long long base = (p.n_is_dim0 ? ((long long)n * p.D1 + k) : ((long long)k * p.D1 + n)) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int a = p.tdy[t] + 1, b = p.tdx[t] + 1;       // filter tap applied to input offset (a-1, b-1)
        const float v = w[base + p.tr[t] * 3 + p.ts[t]];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb)
            if (aa == a && bb == b) g[aa][bb] = v;
      }
      const int i = pos >> 2, j = pos & 3;
      // rows of G: (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1)
      float t3[3];
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        const float g0 = g[0][bb], g1 = g[1][bb], g2 = g[2][bb];
        t3[bb] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
      }
      u = j == 0 ? t3[0] : (j == 1 ? 0.5f * (t3[0] + t3[1] + t3[2]) : (j == 2 ? 0.5f * (t3[0] - t3[1] + t3[2]) : t3[2]));
    }
    wp[idx] = u;
  }
}

int launch_wino_pack(const IgemmParams& p, const float* w, float* wp, hipStream_t stream) {
  const long long total = wino_packed_elems(p);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, stream, p, w, wp, wino_npad(p) / 32, total);
  return check_launch("wino_pack_kernel");
}

// ------------------------------------------------------------------------------------------------ the convolution
__global__ void __launch_bounds__(256, 1) wino_conv_kernel(const IgemmParams p) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order (see igemm_conv_u32_kernel): contiguous logical tile ranges per XCD, N tile fastest
  const int MT = (p.T + WBT - 1) / WBT, NT = p.Npad / WBN;
  const int per = (MT * NT + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= MT * NT) return;
  const int mb = q / NT, nb = q % NT;

  // ---- staging role: one 4x4 patch of 4 channels per thread and chunk
  const int st_tile = tid >> 2, cg = tid & 3;
  unsigned pmask = 0;            // bit 4a+b: patch pixel (a, b) lies inside the image (and the tile exists)
  int pn, py, px;
  {
    const int t = mb * WBT + st_tile;
    unsigned tx, ty;
    const unsigned r = fastdiv_dev(t < p.T ? (unsigned)t : 0u, (unsigned)p.TW, p.mTW, &tx);
    pn = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
    py = 2 * (int)ty - 1;
    px = 2 * (int)tx - 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        pmask |= (t < p.T && (unsigned)(py + a) < (unsigned)p.IH && (unsigned)(px + b) < (unsigned)p.IW) ? (1u << (4 * a + b)) : 0u;
  }
  int rowoffB[DN_MAX_OPERANDS];   // byte offset of patch pixel (0,0), channel 4*cg, per operand
#pragma unroll
  for (int s = 0; s < DN_MAX_OPERANDS; ++s) {
    const KOperand& S = p.in[s < p.n_in ? s : 0];
    rowoffB[s] = (pn * (int)S.sn + py * (int)S.sh + px * (int)S.sw + cg * 4) * 4;
  }

  // total chunks over the concatenated K axis
  int nchunks = 0;
  for (int s = 0; s < p.n_in; ++s) nchunks += p.in[s].C / WKC;

  // load cursor: (operand ls, chunk lc within it)
  int ls = 0, lc = 0;
  f32x4 v[16];
  f32x4 sc4, sh4;
  float relu_floor = 0.f;
  auto load_patch = [&]() {
    const KOperand& S = p.in[ls];
    const char* base = reinterpret_cast<const char*>(S.p);
    const int shB = (int)S.sh * 4, swB = (int)S.sw * 4;
    const int off0 = (ls == 0 ? rowoffB[0] : (ls == 1 ? rowoffB[1] : rowoffB[2])) + lc * (WKC * 4);
    const bool has_aff = S.scale != nullptr;
    const int coff = (lc * WKC + cg * 4) * 4;
    const f32x4 l1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(has_aff ? S.scale : S.p) + (has_aff ? coff : 0));
    const f32x4 l2 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(has_aff ? S.shift : S.p) + (has_aff ? coff : 0));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc4[e] = has_aff ? l1[e] : 1.f;
      sh4[e] = has_aff ? l2[e] : 0.f;
    }
    relu_floor = has_aff ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool ok = (pmask >> (4 * a + b)) & 1u;
        const int off = ok ? off0 + a * shB + b * swB : 0;
        v[4 * a + b] = *reinterpret_cast<const f32x4*>(base + off);
      }
  };
  auto advance_cursor = [&]() {     // clamps on the last chunk (the final iteration re-fetches it into the idle buffer)
    const int nch = p.in[ls].C / WKC;
    const bool last_of_op = lc + 1 == nch;
    const bool last = last_of_op && (ls + 1 == p.n_in);
    lc = last ? lc : (last_of_op ? 0 : lc + 1);
    ls = (!last && last_of_op) ? ls + 1 : ls;
  };
  // BatchNorm-apply + ReLU (or identity), zero halo, B^T d B, 16 float4 into LDS
  auto transform_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = (pmask >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = fmaxf(relu_floor, fmaf(v[i][e], sc4[e], sh4[e]));
        v[i][e] = ok ? t : 0.f;
      }
    }
    f32x4 r[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {       // rows: B^T d
      r[0 + b] = v[0 + b] - v[8 + b];
      r[4 + b] = v[4 + b] + v[8 + b];
      r[8 + b] = v[8 + b] - v[4 + b];
      r[12 + b] = v[4 + b] - v[12 + b];
    }
    float* dst = smem + (size_t)((buf * 2 + (cg >> 1)) * 16) * (WBT * 8) + st_tile * 8 + (cg & 1) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // columns: (B^T d) B
      const f32x4 c0 = r[4 * i + 0] - r[4 * i + 2];
      const f32x4 c1 = r[4 * i + 1] + r[4 * i + 2];
      const f32x4 c2 = r[4 * i + 2] - r[4 * i + 1];
      const f32x4 c3 = r[4 * i + 1] - r[4 * i + 3];
      *reinterpret_cast<f32x4*>(dst + (4 * i + 0) * (WBT * 8)) = c0;
      *reinterpret_cast<f32x4*>(dst + (4 * i + 1) * (WBT * 8)) = c1;
      *reinterpret_cast<f32x4*>(dst + (4 * i + 2) * (WBT * 8)) = c2;
      *reinterpret_cast<f32x4*>(dst + (4 * i + 3) * (WBT * 8)) = c3;
    }
  };

  // ---- B fragments straight from the packed weights
  const int NS = p.Npad / 32;
  const float* wbase = p.w + ((size_t)(4 * wave) * NS + 2 * nb) * 256 + lane * 4;   // + kc8 * 16*NS*256 + j * NS*256 + nn * 256
  const size_t wstep8 = (size_t)16 * NS * 256;
  f32x4 breg[2][4][2];
  auto load_b = [&](int which, int kc8) {
    const float* wsrc = wbase + (size_t)kc8 * wstep8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) breg[which][j][nn] = *reinterpret_cast<const f32x4*>(wsrc + ((size_t)j * NS + nn) * 256);
  };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][m][nn][e] = 0.f;

  const int frA = ((4 * wave) * WBT + (lane & 31)) * 8 + (lane >> 5) * 4;   // + (buf*2+sub)*16*WBT*8 + j*WBT*8 + m*32*8
  auto compute = [&](int buf, int sub, int which) {
    const float* Ab = smem + (size_t)((buf * 2 + sub) * 16) * (WBT * 8) + frA;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + j * (WBT * 8));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + j * (WBT * 8) + 32 * 8);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[j][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], breg[which][j][0][kk], acc[j][0][0], 0, 0, 0);
        acc[j][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], breg[which][j][1][kk], acc[j][0][1], 0, 0, 0);
        acc[j][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], breg[which][j][0][kk], acc[j][1][0], 0, 0, 0);
        acc[j][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], breg[which][j][1][kk], acc[j][1][1], 0, 0, 0);
      }
    }
  };

  // ---- pipeline
  load_patch();
  load_b(0, 0);
  transform_store(0);
  advance_cursor();
  __syncthreads();
  const int last8 = 2 * nchunks - 1;
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    load_patch();                                  // chunk min(c+1, nchunks-1)
    load_b(1, 2 * c + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(buf, 0, 0);
    load_b(0, 2 * c + 2 < last8 ? 2 * c + 2 : last8);
    compute(buf, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    transform_store(buf ^ 1);
    advance_cursor();
    __syncthreads();
  }

  // ---- output transform.  Along j in registers:  Z[b] = sum_j A^T[b][j] M[i][j],  A^T = (1 1 1 0 / 0 1 -1 -1)
  f32x16 Z[2][2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      Z[0][m][nn] = acc[0][m][nn] + acc[1][m][nn] + acc[2][m][nn];
      Z[1][m][nn] = acc[1][m][nn] - acc[2][m][nn] - acc[3][m][nn];
    }
  // Along i across the four waves through LDS (one half b at a time: 4 x 64 x 72 floats = 72 KiB)
  const int c4 = tid & 15, tg = tid >> 4;          // final role: couts 4*c4..+3 of tiles tg + 16*k, k = 0..3
  f32x4 Y[4][2][2];                                // [k][a][b]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          smem[(wave * WBT + row) * WZLD + 32 * nn + (lane & 31)] = Z[b][m][nn][r];
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int tile = tg + 16 * k;
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(smem + (0 * WBT + tile) * WZLD + 4 * c4);
      const f32x4 z1 = *reinterpret_cast<const f32x4*>(smem + (1 * WBT + tile) * WZLD + 4 * c4);
      const f32x4 z2 = *reinterpret_cast<const f32x4*>(smem + (2 * WBT + tile) * WZLD + 4 * c4);
      const f32x4 z3 = *reinterpret_cast<const f32x4*>(smem + (3 * WBT + tile) * WZLD + 4 * c4);
      Y[k][0][b] = z0 + z1 + z2;
      Y[k][1][b] = z1 - z2 - z3;
    }
  }

  // ---- batch statistics of the pre-bias result: partial row 2*mb + m covers tiles [32(2mb+m), +32) = 128 pixels;
  //      (sum, M2 about the group's own mean), merged by dn_bn_finalize
  const int n_first = nb * WBN + 4 * c4;
  if (p.bn_partial != nullptr) {
    float* red = smem + 4 * WBT * WZLD;            // [2 m][4 waves][64 couts], then gmean [2][64]
    float* gmean = red + 2 * 4 * 64;
    int gcount[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      int left = p.T - (mb * WBT + 32 * m);
      gcount[m] = left < 0 ? 0 : (left > 32 ? 32 : left);
    }
    f32x4 s[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      s[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) s[m] += Y[2 * m + kk][a][b];      // tiles past T hold exact zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[m][e] += __shfl_xor(s[m][e], 16);
        s[m][e] += __shfl_xor(s[m][e], 32);
      }
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(red + (m * 4 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < 128) {
      const int m = tid >> 6, col = tid & 63;
      tot = red[(m * 4 + 0) * 64 + col] + red[(m * 4 + 1) * 64 + col] + red[(m * 4 + 2) * 64 + col] + red[(m * 4 + 3) * 64 + col];
      gmean[tid] = gcount[m] > 0 ? tot / (float)(4 * gcount[m]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(gmean + m * 64 + 4 * c4);
      f32x4 s2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bool live = (tg + 16 * kk) < gcount[m];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const f32x4 dv = Y[2 * m + kk][a][b] - mu;
#pragma unroll
            for (int e = 0; e < 4; ++e) s2[e] += live ? dv[e] * dv[e] : 0.f;
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s2[e] += __shfl_xor(s2[e], 16);
        s2[e] += __shfl_xor(s2[e], 32);
      }
      s[m] = s2;
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(red + (m * 4 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    if (tid < 128) {
      const int m = tid >> 6, col = tid & 63;
      const float m2 = red[(m * 4 + 0) * 64 + col] + red[(m * 4 + 1) * 64 + col] + red[(m * 4 + 2) * 64 + col] + red[(m * 4 + 3) * 64 + col];
      const int n = nb * WBN + col;
      if (n < p.Ntot && gcount[m] > 0) {
        float* dst = p.bn_partial + ((long long)(2 * mb + m) * p.Ntot + n) * 2;
        dst[0] = tot;
        dst[1] = m2;
      }
    }
  }

  // ---- bias, activation, channel-split / accumulating float4 stores
  if (n_first < p.Ntot) {
    int seg = 0;
    if (p.n_out > 1 && n_first >= p.out[1].n_begin) seg = 1;
    if (p.n_out > 2 && n_first >= p.out[2].n_begin) seg = 2;
    const KResult& R = p.out[seg];
    float* obase = R.p + (n_first - R.n_begin);
    const long long sw = R.sw;
    const bool accumulate = R.accumulate != 0;
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bias = *reinterpret_cast<const f32x4*>(p.bias + n_first);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = mb * WBT + tg + 16 * k;
      if (t < p.T) {
        unsigned tx, ty;
        const unsigned r = fastdiv_dev((unsigned)t, (unsigned)p.TW, p.mTW, &tx);
        const int n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
        const long long pix0 = ((long long)n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float* o = obase + (pix0 + (long long)a * p.OW + b) * sw;
            f32x4 val;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = wino_act(Y[k][a][b][e] + bias[e], p.act, p.act_p0, p.act_p1);
            if (accumulate) val += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = val;
          }
      }
    }
  }
}

int launch_wino_conv(IgemmParams& p, hipStream_t stream) {
  p.Npad = wino_npad(p);
  p.T = p.M / 4;
  p.TH = p.OH / 2;
  p.TW = p.OW / 2;
  p.mTW = fastdiv_magic((unsigned)p.TW);
  p.mTH = fastdiv_magic((unsigned)p.TH);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino_conv_kernel, %zu): %s", kWinoLds, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  const int tiles = ((p.T + WBT - 1) / WBT) * (p.Npad / WBN);
  dim3 grid((tiles + 7) / 8 * 8);
  hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), kWinoLds, stream, p);
  set_last_kernel("dn::wino_conv_kernel");
  return check_launch("wino_conv_kernel");
}

}  // namespace dn
