// Fused DORN head: 1x1 convolution 16 -> 2K logits + Dropout2d channel mask + logit clamp + pair softmax, with
// the 2K-channel logits NEVER written to HBM -- forward writes only what the caller's API exposes (ord_c1 [N,K,H,W] and decode_c),
// backward goes from d(ord_c1) straight to d(iconv0), d(conv_ord.weight) and d(conv_ord.bias) by recomputing the logits
// (16 x 2K multiply-accumulates per pixel: cheaper than reading them back).
//
// Replaces reference models/Disp_vgg_BN_DORN.py:112-114,191-227: self.dropout -> self.conv_ord -> OrdinalRegressionLayer
// (x[:, ::2] / x[:, 1::2] pairs, clamp(1e-8, 1e8), softmax over the pair, ord_c1 = P(second), decode_c = sum(ord_c1 > 0.5)).
//
// Unfused (r01): conv writes 2K floats/px, ordinal_fwd re-reads them and writes K; backward writes and re-reads the 2K-channel
// dpre three times (ordinal_bwd, bias/act reduce, wgrad, dgrad) = ~1500 floats/px of traffic at K = 80.  Fused: forward 16 + K + 2,
// backward K + 2 x 16 + 16 = ~210 floats/px.  HBM-bound: 0.65 GB forward at b32 128x416 K = 80.
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32, exact fp32): a wave owns 64 consecutive pixels = 4 pixel tiles of 16.  Logit
// channels are visited in tiles of 16 such that tile 2u holds the FIRST logits of pairs 16u..16u+15 and tile 2u+1 their SECOND
// logits: both members of a pair land in the same lane.  The D-fragment of an MFMA (lane (g, j): rows 4g..4g+3, column j) is
// directly usable as the A or B operand of a following MFMA whose contraction index is D's ROW index (operand step r <-> row
// 4g + r: the contraction visits its 16 indices in a permuted order, which a sum does not mind).  So
//   channel-major fragments  L^T[ch][px] = W[ch][c] x^T[c][px]   (rows = channels)  feed  dx^T[c][px] = W^T[c][ch] dL^T[ch][px]
//   pixel-major fragments    L[px][ch]   = x[px][c] W^T[c][ch]   (rows = pixels)    feed  dW[ch][c]   = dL^T[ch][px] x[px][c]
// and the backward kernel builds the gradient of the logits in both orientations from one pass over d(ord_c1) (the second
// fetch of a 64-pixel chunk hits L2).  The forward kernel uses the channel-major form: a lane then stores 16 consecutive pixels
// of one ord plane per 64-byte segment.
#include "dn_internal.h"

namespace dn {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kOhThreads = 256;          // 4 waves
constexpr int kOhMaxPT = 5;              // pair tiles of 16: K <= 80
constexpr int kOhCin = 16;

__device__ __forceinline__ float oh_clamp(float v) { return fminf(fmaxf(v, 1e-8f), 1e8f); }
__device__ __forceinline__ bool oh_inrange(float v) { return v >= 1e-8f && v <= 1e8f; }

// P(second) of softmax over (a, b): with m = max(a, b) one exponential is exp(0) = 1 and the other is exp(-|b - a|) in (0, 1]:
// v_exp_f32 (1 ulp on 2^x, argument already <= 0, no range handling needed) and v_rcp_f32 (1 ulp) -- every vector instruction
// here costs matrix-pipe time (the fp32 MFMA issues on the same lanes), and the libm forms are ~25 instructions per probability
__device__ __forceinline__ float oh_prob(float a, float b) {
  const float d = b - a;
  const float e = __builtin_amdgcn_exp2f(-fabsf(d) * 1.44269504088896340736f);
  const float inv = __builtin_amdgcn_rcpf(1.f + e);
  return d >= 0.f ? inv : e * inv;
}

// LDS image of the layer: W[T][row][c] = weight[chmap(T,row)][c] (zero rows past K), B[T][row] = bias[chmap(T,row)],
//   chmap(T, row) = 2 * (16 * (T >> 1) + row) + (T & 1);  rows padded to 20 floats: the b128 fragment reads (16 lanes = 16 rows)
//   and the dword reads (16 consecutive c of 4 rows) both touch 64 distinct banks
constexpr int kOhLd = 20;
struct OhLds {
  float W[2 * kOhMaxPT][16][kOhLd];
  float B[2 * kOhMaxPT][16];
};

__device__ __forceinline__ void oh_fill_lds(OhLds& s, const float* __restrict__ w, const float* __restrict__ bias, int K, int NPT) {
  for (int i = threadIdx.x; i < 2 * NPT * 16 * kOhCin; i += kOhThreads) {
    const int c = i & 15, row = (i >> 4) & 15, T = i >> 8;
    const int pair = 16 * (T >> 1) + row;
    s.W[T][row][c] = pair < K ? w[(2 * pair + (T & 1)) * kOhCin + c] : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * NPT * 16; i += kOhThreads) {
    const int row = i & 15, T = i >> 4;
    const int pair = 16 * (T >> 1) + row;
    s.B[T][row] = pair < K ? bias[2 * pair + (T & 1)] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------ forward
// pixel-major fragments L[px][ch] = x[px][c] W^T[c][ch]: lane (g, j) ends up with pixels 4g..4g+3 of pair 16u + j, i.e. one
// 16-byte store per lane into the ord plane of that pair (a store instruction writes 16 planes x 64 contiguous bytes)
template <int NPT>
__global__ void __launch_bounds__(kOhThreads) ord_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                                  const float* __restrict__ w, const float* __restrict__ bias, long long HW,
                                                                  long long chunks, int K, float* __restrict__ ord,
                                                                  long long* __restrict__ decode) {
  __shared__ OhLds s;
  oh_fill_lds(s, w, bias, K, NPT);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  f4 wb[2 * NPT];                                   // B operand: W^T[c = 4g + step][ch = chmap(T, j)]
  float bj[2 * NPT];
#pragma unroll
  for (int T = 0; T < 2 * NPT; ++T) {
    wb[T] = *reinterpret_cast<const f4*>(&s.W[T][j][4 * g]);
    bj[T] = s.B[T][j];
  }
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long ch = wave0; ch < chunks; ch += nwaves) {
    const long long pix0 = ch * 64;                     // 64 pixels of one image (HW % 64 == 0)
    const long long n = pix0 / HW, p0 = pix0 - n * HW;
    f4 mk = {1.f, 1.f, 1.f, 1.f};
    if (mask != nullptr) mk = *reinterpret_cast<const f4*>(mask + n * kOhCin + 4 * g);
#pragma unroll 1
    for (int pt = 0; pt < 4; ++pt) {
      f4 xa = *reinterpret_cast<const f4*>(x + (pix0 + pt * 16 + j) * kOhCin + 4 * g);    // A operand: x[px = j][c = 4g + step]
      xa *= mk;
      int cnt[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < NPT; ++u) {
        f4 la = {bj[2 * u], bj[2 * u], bj[2 * u], bj[2 * u]};
        f4 lb = {bj[2 * u + 1], bj[2 * u + 1], bj[2 * u + 1], bj[2 * u + 1]};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          la = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], wb[2 * u][st], la, 0, 0, 0);
          lb = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], wb[2 * u + 1][st], lb, 0, 0, 0);
        }
        f4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = oh_prob(oh_clamp(la[r]), oh_clamp(lb[r]));
        const int k = 16 * u + j;
        if (k < K) {
          *reinterpret_cast<f4*>(ord + (n * K + k) * HW + p0 + pt * 16 + 4 * g) = pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) cnt[r] += pv[r] > 0.5f ? 1 : 0;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) cnt[r] += __shfl_xor(cnt[r], o);
      }
      if (j == 0) {
        long long* d = decode + pix0 + pt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = cnt[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------ backward
// partial[block] = [2K*16 dW (framework layout [ch][c])][2K db]
template <int NPT>
__global__ void __launch_bounds__(kOhThreads) ord_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  const float* __restrict__ dord, long long HW, long long chunks, int K,
                                                                  float* __restrict__ dx, int accumulate, float* __restrict__ partial) {
  __shared__ OhLds s;
  __shared__ float tr[4][2][16][kOhLd];              // per wave: the two gradient tiles of a pair tile, for the transposed re-read
  oh_fill_lds(s, w, bias, K, NPT);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  f4 dw[2 * NPT];                                    // dW fragment: rows = channel index within tile T, column = c
  float db[2 * NPT];
#pragma unroll
  for (int T = 0; T < 2 * NPT; ++T) {
    dw[T] = f4{0.f, 0.f, 0.f, 0.f};
    db[T] = 0.f;
  }
  const long long wave0 = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
  for (long long ch = wave0; ch < chunks; ch += nwaves) {
    const long long pix0 = ch * 64;
    const long long n = pix0 / HW, p0 = pix0 - n * HW;
    f4 mk = {1.f, 1.f, 1.f, 1.f};
    float mkj = 1.f;
    if (mask != nullptr) {
      mk = *reinterpret_cast<const f4*>(mask + n * kOhCin + 4 * g);
      mkj = mask[n * kOhCin + j];
    }
#pragma unroll 1
    for (int pt = 0; pt < 4; ++pt) {
      asm volatile("" ::: "memory");                  // keep the per-tile LDS operand reads inside the loop (register budget)
      const long long pb = pix0 + pt * 16;
      f4 xa = *reinterpret_cast<const f4*>(x + (pb + j) * kOhCin + 4 * g);     // A of the logits: x[px = j][c = 4g + step] (masked)
      xa *= mk;
      float xr[4];                                                            // B of dW: x[px = 4g + r][c = j] (masked)
#pragma unroll
      for (int r = 0; r < 4; ++r) xr[r] = x[(pb + 4 * g + r) * kOhCin + j] * mkj;
      f4 dxa = {0.f, 0.f, 0.f, 0.f}, dxb = {0.f, 0.f, 0.f, 0.f};              // dx^T fragments (two chains): rows c = 4g + r, column px = j
#pragma unroll
      for (int u = 0; u < NPT; ++u) {
        // ---- pixel-major logits L[px][ch] (rows px = 4g + r, column pair j): both logits of pair 16u + j in this lane
        const f4 w0 = *reinterpret_cast<const f4*>(&s.W[2 * u][j][4 * g]), w1 = *reinterpret_cast<const f4*>(&s.W[2 * u + 1][j][4 * g]);
        const float ba = s.B[2 * u][j], bb = s.B[2 * u + 1][j];
        f4 la = {ba, ba, ba, ba}, lb = {bb, bb, bb, bb};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          la = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], w0[st], la, 0, 0, 0);
          lb = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], w1[st], lb, 0, 0, 0);
        }
        const int k = 16 * u + j;
        f4 gv = {0.f, 0.f, 0.f, 0.f};
        if (k < K) gv = *reinterpret_cast<const f4*>(dord + (n * K + k) * HW + p0 + pt * 16 + 4 * g);   // 4 consecutive pixels
        f4 da, dbv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float P = oh_prob(oh_clamp(la[r]), oh_clamp(lb[r]));
          const float t = gv[r] * P * (1.f - P);
          da[r] = oh_inrange(la[r]) ? -t : 0.f;
          dbv[r] = oh_inrange(lb[r]) ? t : 0.f;
        }
        db[2 * u] += (da[0] + da[1]) + (da[2] + da[3]);
        db[2 * u + 1] += (dbv[0] + dbv[1]) + (dbv[2] + dbv[3]);
        // dW[ch][c] += sum_px dL[px][ch] x[px][c]: A = fragment register r (i = ch = column j, k = px = row 4g + r), B = x[px][c = j]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dw[2 * u] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[r], xr[r], dw[2 * u], 0, 0, 0);
          dw[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dbv[r], xr[r], dw[2 * u + 1], 0, 0, 0);
        }
        // ---- the same tiles channel-major (rows ch = 4g + r, column px = j) through LDS: dx^T[c][px] += W^T[c][ch] dL^T[ch][px]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tr[wave][0][4 * g + r][j] = da[r];
          tr[wave][1][4 * g + r][j] = dbv[r];
        }
        const f4 dat = *reinterpret_cast<const f4*>(&tr[wave][0][j][4 * g]);   // dL[px = j][ch = 4g + r]
        const f4 dbt = *reinterpret_cast<const f4*>(&tr[wave][1][j][4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dxa = __builtin_amdgcn_mfma_f32_16x16x4f32(s.W[2 * u][4 * g + r][j], dat[r], dxa, 0, 0, 0);
          dxb = __builtin_amdgcn_mfma_f32_16x16x4f32(s.W[2 * u + 1][4 * g + r][j], dbt[r], dxb, 0, 0, 0);
        }
      }
      dxa = (dxa + dxb) * mk;                                                 // d(masked x)/dx = mask[n][c], c = 4g + r
      float* dst = dx + (pb + j) * kOhCin + 4 * g;
      if (accumulate) dxa += *reinterpret_cast<const f4*>(dst);
      *reinterpret_cast<f4*>(dst) = dxa;
    }
  }
  // ---- block reduction (fixed order: waves 0..3) through the operand image's LDS, then one partial per block
  __syncthreads();
  float* red = &s.W[0][0][0];                         // >= 4 * 64 floats; one (T, r) slice at a time
  float* out = partial + (long long)blockIdx.x * (2 * K * kOhCin + 2 * K);
#pragma unroll
  for (int T = 0; T < 2 * NPT; ++T) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave * 64 + lane] = dw[T][r];
      __syncthreads();
      if (wave == 0) {
        const float v = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        const int pair = 16 * (T >> 1) + 4 * g + r;    // fragment element: channel row 4g + r of tile T, input channel c = j
        if (pair < K) out[(2 * pair + (T & 1)) * kOhCin + j] = v;
      }
      __syncthreads();
    }
    red[wave * 64 + lane] = db[T];
    __syncthreads();
    if (wave == 0 && g == 0) {
      float v = 0.f;
      for (int q = 0; q < 16; ++q) v += red[q * 16 + j];          // 4 waves x 4 row groups of column j, fixed order
      const int pair = 16 * (T >> 1) + j;
      if (pair < K) out[2 * K * kOhCin + 2 * pair + (T & 1)] = v;
    }
    __syncthreads();
  }
}

// dW / dbias = sum over the blocks' partials.  A block takes 16 consecutive elements; its 256 threads are 16 slices of the block
// dimension (thread (sl, e) sums partials sl, sl + 16, ... of element e: 64-byte coalesced rows), the slices meet in LDS in slice
// order -- fixed order, deterministic.  (One thread per element walking all 768 partials serially took 259 us: longer than half the
// backward kernel it follows.)
__global__ void __launch_bounds__(256) ord_head_reduce_kernel(const float* __restrict__ partial, int blocks, int n, int nw, float* __restrict__ dw,
                                                              float* __restrict__ dbias) {
  __shared__ float sl_sum[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    int b = sl;
    for (; b + 16 < blocks; b += 32) {
      a0 += partial[(long long)b * n + i];
      a1 += partial[(long long)(b + 16) * n + i];
    }
    if (b < blocks) a0 += partial[(long long)b * n + i];
  }
  sl_sum[sl][e] = a0 + a1;
  __syncthreads();
  if (sl == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sl_sum[q][e];
    if (i < nw) dw[i] = s;
    else dbias[i - nw] = s;
  }
}

static int oh_blocks(long long chunks) {
  long long b = (chunks + 3) / 4;
  const long long cap = 256 * 3;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace dn

using namespace dn;

extern "C" {

int32_t dn_ord_head_supported(int32_t C_in, int64_t HW, int32_t K) { return (C_in == kOhCin && HW % 64 == 0 && K >= 1 && K <= 16 * kOhMaxPT) ? 1 : 0; }

int32_t dn_ord_head_bwd_blocks(int32_t N, int64_t HW) { return oh_blocks((long long)N * HW / 64); }

int dn_ord_head_fwd(const float* x, const float* mask, const float* w, const float* bias, int32_t N, int64_t HW, int32_t K, float* ord,
                    int64_t* decode, dn_stream_t stream) {
  DN_REQUIRE(x && w && bias && ord && decode && N > 0, DN_ERR_BAD_ARG, "dn_ord_head_fwd: bad argument");
  DN_REQUIRE(dn_ord_head_supported(kOhCin, HW, K), DN_ERR_UNSUPPORTED, "dn_ord_head_fwd: needs 16 input channels, H*W %% 64 == 0, K <= 80");
  const long long chunks = (long long)N * HW / 64;
  const int npt = (K + 15) / 16;
  hipStream_t s = as_stream(stream);
  dim3 grid(oh_blocks(chunks)), block(kOhThreads);
  long long* dec = reinterpret_cast<long long*>(decode);
  switch (npt) {
    case 1: DN_LAUNCH(ord_head_fwd_kernel<1>, grid, block, 0, s, x, mask, w, bias, (long long)HW, chunks, K, ord, dec); break;
    case 2: DN_LAUNCH(ord_head_fwd_kernel<2>, grid, block, 0, s, x, mask, w, bias, (long long)HW, chunks, K, ord, dec); break;
    case 3: DN_LAUNCH(ord_head_fwd_kernel<3>, grid, block, 0, s, x, mask, w, bias, (long long)HW, chunks, K, ord, dec); break;
    case 4: DN_LAUNCH(ord_head_fwd_kernel<4>, grid, block, 0, s, x, mask, w, bias, (long long)HW, chunks, K, ord, dec); break;
    default: DN_LAUNCH(ord_head_fwd_kernel<5>, grid, block, 0, s, x, mask, w, bias, (long long)HW, chunks, K, ord, dec); break;
  }
  return check_launch("ord_head_fwd_kernel");
}

int dn_ord_head_bwd(const float* x, const float* mask, const float* w, const float* bias, const float* dord, int32_t N, int64_t HW, int32_t K,
                    float* dx, int32_t accumulate, float* workspace, float* dw, float* dbias, dn_stream_t stream) {
  DN_REQUIRE(x && w && bias && dord && dx && workspace && dw && dbias && N > 0, DN_ERR_BAD_ARG, "dn_ord_head_bwd: bad argument");
  DN_REQUIRE(dn_ord_head_supported(kOhCin, HW, K), DN_ERR_UNSUPPORTED, "dn_ord_head_bwd: needs 16 input channels, H*W %% 64 == 0, K <= 80");
  const long long chunks = (long long)N * HW / 64;
  const int npt = (K + 15) / 16;
  hipStream_t s = as_stream(stream);
  const int nb = oh_blocks(chunks);
  dim3 grid(nb), block(kOhThreads);
  switch (npt) {
    case 1: DN_LAUNCH(ord_head_bwd_kernel<1>, grid, block, 0, s, x, mask, w, bias, dord, (long long)HW, chunks, K, dx, accumulate, workspace); break;
    case 2: DN_LAUNCH(ord_head_bwd_kernel<2>, grid, block, 0, s, x, mask, w, bias, dord, (long long)HW, chunks, K, dx, accumulate, workspace); break;
    case 3: DN_LAUNCH(ord_head_bwd_kernel<3>, grid, block, 0, s, x, mask, w, bias, dord, (long long)HW, chunks, K, dx, accumulate, workspace); break;
    case 4: DN_LAUNCH(ord_head_bwd_kernel<4>, grid, block, 0, s, x, mask, w, bias, dord, (long long)HW, chunks, K, dx, accumulate, workspace); break;
    default: DN_LAUNCH(ord_head_bwd_kernel<5>, grid, block, 0, s, x, mask, w, bias, dord, (long long)HW, chunks, K, dx, accumulate, workspace); break;
  }
  const int n = 2 * K * kOhCin + 2 * K;
  DN_LAUNCH(ord_head_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, s, workspace, nb, n, 2 * K * kOhCin, dw, dbias);
  return check_launch("ord_head_bwd_kernel");
}

}  // extern "C"
