// Weight gradient of the THIN full-resolution 3x3 layers (gfx950 only): the first encoder layer (3 -> 64 on the NCHW image,
// torchvision vgg16_bn features[0]) and iconv0 (16 + 1 -> 16, models/Disp_vgg_BN.py:105,185).  On the 32-wide implicit-GEMM
// tiles these pad 16 output channels to 32 and every operand's taps*channels to a multiple of 32 (16+1 channels: 153 -> 192),
// and their 1.7 M pixels x 16-64 gradient channels are read through LDS for 6-8 GFLOP of useful work: 0.5 ms per launch.
//
// Here:  dW[co][k] = sum over pixels of dy[pixel][co] * x[pixel + tap(k)][c(k)]  as 16x16x4 MFMAs straight from global memory:
//   A = dy^T   (lane: co = lane & 15, pixel = lane >> 4)   one coalesced dword load per 16 output channels and 4 pixels,
//   B = x      (lane: column k = lane & 15, pixel = lane >> 4)   one gathered dword load per 16 columns and 4 pixels; the
//              columns are the framework's own [channel][3][3] order, operand by operand (a 16-column tile never straddles two
//              operands, so the operand -- strides, nearest-x2 upsample, buffer descriptor -- is wave-uniform); halo and dead
//              columns read zeros through the buffer bounds check.
// A wave walks whole image rows, four pixels of one row per step, the next step's loads in flight under the current MFMAs; the
// four waves of a block are summed through LDS and the blocks by a second, fixed-order pass (deterministic).
#include <stdlib.h>

#include "dn_internal.h"

namespace dn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThinMaxTiles = 10;

struct ThinTab {
  int op[kThinMaxTiles];      // operand of this 16-column tile (-1: dead tile)
  int e0[kThinMaxTiles];      // scalar-mode tile (tap < 0): first local column (c*9 + tap) inside its operand; else first channel
  int tap[kThinMaxTiles];     // channel-mode tile (operands with C % 16 == 0): the tile is 16 consecutive channels of this tap, so
                              //   a wave's gather is 64 contiguous bytes per pixel (the TA, one per CU, is what bounds this kernel)
  int ntiles;
};

static bool thin_build_tab(const IgemmParams& p, ThinTab* tab) {
  int nt = 0;
  for (int s = 0; s < p.n_in; ++s) {
    const int C = p.in[s].C;
    if (C % 16 == 0 && p.in[s].sc == 1) {
      for (int t = 0; t < 9; ++t)
        for (int c0 = 0; c0 < C; c0 += 16) {
          if (nt >= kThinMaxTiles) return false;
          tab->op[nt] = s;
          tab->e0[nt] = c0;
          tab->tap[nt] = t;
          ++nt;
        }
    } else {
      for (int e = 0; e < C * 9; e += 16) {
        if (nt >= kThinMaxTiles) return false;
        tab->op[nt] = s;
        tab->e0[nt] = e;
        tab->tap[nt] = -1;
        ++nt;
      }
    }
  }
  tab->ntiles = nt;
  for (int j = nt; j < kThinMaxTiles; ++j) {
    tab->op[j] = -1;
    tab->e0[j] = 0;
    tab->tap[j] = -1;
  }
  return true;
}

static bool thin_shape(const IgemmParams& p, int* ntc, int* nkt) {
  ThinTab tab;
  if (!thin_build_tab(p, &tab)) return false;
  *ntc = p.Ntot / 16;
  *nkt = tab.ntiles;
  if (*ntc == 1 && *nkt <= 10) { *nkt = 10; return true; }
  if (*ntc == 4 && *nkt <= 2) { *nkt = 2; return true; }
  return false;
}

bool thin_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_thin) return false;
  if (d->kind != DN_CONV_FWD) return false;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return false;
  if (d->IH != d->OH || d->IW != d->OW || (d->OW & 3)) return false;
  if (p.Ntot % 16 != 0 || p.Ntot > 64) return false;
  if ((long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31)) return false;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!o.small || o.scale != nullptr || o.C > 64) return false;
  }
  int ntc, nkt;
  return thin_shape(p, &ntc, &nkt);
}

static int thin_blocks(const IgemmParams& p) {
  const int rows = p.N * p.GH;
  int blocks = (rows + 3) / 4;          // one image row per wave at least
  if (blocks > 512) blocks = 512;       // two blocks per CU
  return blocks;
}

size_t thin_wgrad_workspace_bytes(const IgemmParams& p) {
  int ntc, nkt;
  if (!thin_shape(p, &ntc, &nkt)) return 0;
  return (size_t)thin_blocks(p) * ntc * 16 * nkt * 16 * sizeof(float);
}

template <int NTC, int NKT>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(const IgemmParams p, const ThinTab tab, int rows_per_wave) {
  __shared__ float red[NTC * NKT * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, pp = lane >> 4;
  const int gwave = blockIdx.x * 4 + wave;
  const int OW = p.GW, OH = p.GH, Cout = p.Ntot;

  // per-lane column constants, per tile: tap offsets, channel byte offset inside the operand, liveness
  int cdy[NKT], cdx[NKT], ccb[NKT];
  bool cval[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int s = tab.op[j];
    const KOperand& S = p.in[s < 0 ? 0 : s];
    const bool chan_mode = tab.tap[j] >= 0;
    const int e = tab.e0[j] + col;
    const int c = chan_mode ? e : e / 9, tap = chan_mode ? tab.tap[j] : e - 9 * (e / 9);
    cval[j] = s >= 0 && (chan_mode ? c < S.C : e < S.C * 9);
    cdy[j] = tap / 3 - 1;
    cdx[j] = tap - 3 * (tap / 3) - 1;
    ccb[j] = c * (int)S.sc;
  }
  const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);

  f32x4 acc[NTC][NKT];
#pragma unroll
  for (int i = 0; i < NTC; ++i)
#pragma unroll
    for (int j = 0; j < NKT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rows = p.N * OH;
  const int r_begin = gwave * rows_per_wave;
  const int r_end = min(rows, r_begin + rows_per_wave);
  const int steps_per_row = OW / 4;
  const int nsteps = r_end > r_begin ? (r_end - r_begin) * steps_per_row : 0;

  constexpr int D = 4;       // steps in flight: the gathers are latency-bound (a step is ~300 cycles of MFMA, a miss is thousands)
  float a[D][NTC], b[D][NKT];
  auto issue = [&](int which, int st) {
    const int rr = st / steps_per_row;
    const int x0 = (st - rr * steps_per_row) * 4;
    const int r = r_begin + rr;
    const int n = r / OH, y = r - n * OH;
    const int m = r * OW + x0 + pp;
    if constexpr (NTC == 4) {
      // 64 output channels: row tile i holds channels 4*col + i, so the four A values of a lane are ONE float4 of dy
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const f32x4 a4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcG, (m * Cout + 4 * col) * 4, 0, 0));
#pragma unroll
      for (int i = 0; i < NTC; ++i) a[which][i] = a4[i];
    } else {
#pragma unroll
      for (int i = 0; i < NTC; ++i) a[which][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcG, (m * Cout + 16 * i + col) * 4, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int s = tab.op[j];
      const KOperand& S = p.in[s < 0 ? 0 : s];
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
      const int iy = y + cdy[j], ix = x0 + pp + cdx[j];
      const bool ok = cval[j] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      int off = (n * (int)S.sn + (iy >> S.up) * (int)S.sh + (ix >> S.up) * (int)S.sw + ccb[j]) * 4;
      off = ok ? off : -1;
      b[which][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
    }
  };
  auto compute = [&](int which) {
#pragma unroll
    for (int i = 0; i < NTC; ++i)
#pragma unroll
      for (int j = 0; j < NKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[which][i], b[which][j], acc[i][j], 0, 0, 0);
  };
  // software pipeline, D-1 steps ahead; steps past the end re-issue the last one (harmless) so the loop body has no branches
  const int last = nsteps - 1;
  if (nsteps > 0) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(d, d < last ? d : last);
    for (int st = 0; st < nsteps; st += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int nx = st + d + D - 1;
        issue((d + D - 1) % D, nx < last ? nx : last);
        if (st + d < nsteps) compute(d);
      }
    }
  }

  // ---- block sum through LDS, then this block's partial [NTC*16 co][NKT*16 columns]
  // C/D layout of the 16x16 tile: column = lane & 15, rows 4*(lane >> 4) + r
  for (int w = 0; w < 4; ++w) {          // fixed order: deterministic
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NTC; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = &red[(i * NKT + j) * 256 + (4 * pp + r) * 16 + col];
            *dst = (w == 0 ? 0.f : *dst) + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  float* out = p.ws + (size_t)blockIdx.x * (NTC * NKT * 256);
  for (int e = tid; e < NTC * NKT * 256; e += 256) out[e] = red[e];
}

// dw[co][(ch_off + c)][tap] = sum over blocks of partial[(i*NKT + j)*256 + (co & 15)*16 + col]: 32 consecutive elements per block,
// the partials dealt over 8 thread groups and folded through LDS in a fixed order (deterministic)
__global__ void __launch_bounds__(256) thin_wgrad_reduce_kernel(const IgemmParams p, const ThinTab tab, const float* __restrict__ ws, float* __restrict__ dw,
                                                                int nblocks, int NTC, int NKT) {
  __shared__ float part[8][32];
  const int total = NTC * NKT * 256;
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), zs = threadIdx.x >> 5;
  float sum = 0.f;
  if (e < total)
    for (int z = zs; z < nblocks; z += 8) sum += ws[(size_t)z * total + e];
  part[zs][threadIdx.x & 31] = sum;
  __syncthreads();
  if (zs != 0 || e >= total) return;
  sum = ((part[0][e & 31] + part[1][e & 31]) + (part[2][e & 31] + part[3][e & 31])) + ((part[4][e & 31] + part[5][e & 31]) + (part[6][e & 31] + part[7][e & 31]));
  const int tile = e >> 8, within = e & 255;
  const int i = tile / NKT, j = tile - i * NKT;
  const int co = NTC == 4 ? 4 * (within >> 4) + i : 16 * i + (within >> 4), col = within & 15;    // (64-channel layers interleave the row tiles)
  const int s = tab.op[j];
  if (s < 0) return;
  const KOperand& S = p.in[s];
  const bool chan_mode = tab.tap[j] >= 0;
  const int le = tab.e0[j] + col;
  const int c = chan_mode ? le : le / 9, tap = chan_mode ? tab.tap[j] : le - 9 * (le / 9);
  if (c >= S.C || co >= p.Ntot) return;
  dw[((long long)co * p.D1 + S.ch_off + c) * 9 + tap] = sum;
}

int launch_thin_wgrad(IgemmParams& p, float* dw, hipStream_t stream) {
  ThinTab tab;
  int ntc, nkt;
  if (!thin_build_tab(p, &tab) || !thin_shape(p, &ntc, &nkt)) {
    set_error("thin wgrad: unsupported shape");
    return DN_ERR_UNSUPPORTED;
  }
  const int blocks = thin_blocks(p);
  const int rows = p.N * p.GH;
  const int rpw = (rows + blocks * 4 - 1) / (blocks * 4);
  if (ntc == 1) DN_LAUNCH((thin_wgrad_kernel<1, 10>), dim3(blocks), dim3(256), 0, stream, p, tab, rpw);
  else DN_LAUNCH((thin_wgrad_kernel<4, 2>), dim3(blocks), dim3(256), 0, stream, p, tab, rpw);
  set_last_kernel("dn::thin_wgrad_kernel<%d, %d>", ntc, nkt);
  int rc = check_launch("thin_wgrad_kernel");
  if (rc != DN_OK) return rc;
  const int total = ntc * nkt * 256;
  DN_LAUNCH(thin_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, stream, p, tab, p.ws, dw, blocks, ntc, nkt);
  return check_launch("thin_wgrad_reduce_kernel");
}


// =====================================================================================================================
// Thin convolutions: at most 32 output channels and a short contraction (iconv0 17 -> 16 and its input gradient, the last two
// conv-transposes 32 -> 16 / 64 -> 32 and the 32 -> 16 one's input gradient; models/Disp_vgg_BN.py:101-105).  The tiled kernel
// pads 16 output channels to a 32-wide MFMA tile and every operand's taps*channels to a multiple of 32, and with K of 128-256
// a block's main loop is 4-8 barrier-separated iterations between a prologue and an epilogue.  Here: v_mfma_f32_16x16x4_f32,
// A fragments gathered straight from global memory (one float4 = 4 channels x 4 MFMA steps per pixel and tap; the 1-channel
// upsampled-disparity piece as dword gathers), the whole weight matrix of the phase resident in LDS in B-fragment order
// ([k-step][4][16 or 32]: a fragment read is 4 x 16 consecutive floats), waves persistent over groups of 64 pixels, results
// through a per-wave LDS tile so that they leave as whole pixels.  Driven by the same plan (taps, phases, strides) as the tiled
// kernel and reading the same packed weights, so all four conv kinds are served.
// =====================================================================================================================
__device__ __forceinline__ float thin_act(float v, int act, float p0, float p1) {
  switch (act) {
    case DN_ACT_RELU: return v > 0.f ? v : 0.f;
    case DN_ACT_LEAKY: return v > 0.f ? v : v * p0;
    case DN_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
    case DN_ACT_SIGMOID_AFFINE: return p0 / (1.f + expf(-v)) + p1;
    default: return v;
  }
}

static int thin_ksteps(const IgemmParams& p, int ntaps) {       // 4-wide MFMA k-steps of one phase
  int k4 = 0;
  for (int s = 0; s < p.n_in; ++s) k4 += (p.in[s].C % 16 == 0) ? ntaps * (p.in[s].C / 16) * 4 : (ntaps + 3) / 4;
  return k4;
}

static size_t thin_conv_lds(const IgemmParams& p, int NT) {
  int k4 = 0;
  for (int z = 0; z < p.nphases; ++z) {
    const int k = thin_ksteps(p, p.ph[z].ntaps);
    if (k > k4) k4 = k;
  }
  const int NP = 16 * NT;
  return (size_t)k4 * 4 * NP * sizeof(float) + 4 * 16 * (NP + 4) * sizeof(float) + 4 * 64 * sizeof(int) + 16 * sizeof(int);
}

bool thin_conv_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_thin || knobs().no_thin_conv) return false;
  // measured: with 32 output channels / 256-long contractions the tiled kernel wins (0.096 vs 0.132 ms, 0.094 vs 0.164); the
  // 16-channel layers are where padding to a 32-wide tile hurts (0.31 -> 0.15 ms, 0.23 -> 0.16)
  // (measured in round 3: the input gradient of iconv0 -- 16 -> 16 + 1 channels, i.e. 17 columns of a 32-wide tile and two results --
  //  takes 0.321 ms here against 0.268 ms on the tiled kernel: at 16 channels x 9 taps these layers are bound by the fp32 matrix
  //  instruction (8.3 GFLOP = 53 us at its 157 TFLOP/s peak) rather than by their 220 MB of HBM traffic (27 us), and a second column
  //  tile doubles the matrix work; it stays on the tiled kernel)
  if (p.reflect || p.bn_partial != nullptr || p.Ntot > 16) return false;
  for (int z = 0; z < p.nphases; ++z)
    if (p.ph[z].ntaps > 16 || p.ph[z].ntaps < 1) return false;
  int kmax = 0;
  for (int s = 0; s < p.n_in; ++s) {
    const KOperand& o = p.in[s];
    const bool vec16 = o.vec && o.small && o.up == 0 && o.C % 16 == 0;
    const bool scalar1 = o.C == 1 && o.small;
    if (o.scale != nullptr || !(vec16 || scalar1)) return false;
    kmax += o.C;
  }
  for (int z = 0; z < p.nphases; ++z)
    if (thin_ksteps(p, p.ph[z].ntaps) * 4 > 320) return false;           // long contractions belong to the tiled kernel
  // one dense float4-addressable result (the whole-pixel store path); a channel-split input gradient stays on the tiled kernel
  const KResult& r = p.out[0];
  if (p.n_out != 1 || (p.Ntot & 3) || (r.sw & 3) || (r.sh & 3) || (r.sn & 3) || r.accumulate || (reinterpret_cast<uintptr_t>(r.p) & 15)) return false;
  return thin_conv_lds(p, p.Ntot <= 16 ? 1 : 2) <= 64 * 1024;
}

template <int NT>
__global__ void __launch_bounds__(256) thin_conv_kernel(const IgemmParams p, int k4max) {
  constexpr int NP = 16 * NT, TLD = NP + 4;
  extern __shared__ __align__(16) float smem[];
  float* wl = smem;                                         // [k4][4][NP]
  float* tiles = smem + (size_t)k4max * 4 * NP;             // [4 waves][16][TLD]
  int* outpix = reinterpret_cast<int*>(tiles + 4 * 16 * TLD);   // [4 waves][64]
  int* tapl = outpix + 4 * 64;                              // [16]  dy | dx << 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kk = lane >> 4;
  const KPhase ph = p.ph[blockIdx.y];
  const int ntaps = ph.ntaps, Kp = ph.nchunks * kChunk;

  if (tid < 16) tapl[tid] = tid < ntaps ? (((int)p.tdy[ph.tap0 + tid] & 0xffff) | ((int)p.tdx[ph.tap0 + tid] << 16)) : 0;
  // weights of this phase -> B-fragment order.  k-step order: operand by operand; a 16-aligned operand contributes, per tap and
  // 16-channel block, four steps (step s holds channels 4*kk + s of the block); a 1-channel operand one step per four taps.
  {
    int k4 = 0, kbase = 0;
    for (int s = 0; s < p.n_in; ++s) {
      const int C = p.in[s].C;
      const int span = ((ntaps * C + kChunk - 1) / kChunk) * kChunk;
      const int steps = (C % 16 == 0) ? ntaps * (C / 16) * 4 : (ntaps + 3) / 4;
      for (int e = tid; e < steps * 4 * NP; e += 256) {
        const int n = e % NP, kq = (e / NP) & 3, st = e / (4 * NP);
        int kpk = -1;
        if (C % 16 == 0) {
          const int s4 = st & 3, blk = st >> 2;             // blk = tap * (C/16) + c16
          const int tap = blk / (C / 16), c16 = blk - tap * (C / 16);
          kpk = kbase + tap * C + c16 * 16 + 4 * kq + s4;
        } else {
          const int tap = st * 4 + kq;
          if (tap < ntaps) kpk = kbase + tap * C;
        }
        wl[(size_t)(k4 + st) * 4 * NP + kq * NP + n] = (kpk >= 0 && n < p.Ntot) ? p.w[ph.w_off + (long long)n * Kp + kpk] : 0.f;
      }
      k4 += steps;
      kbase += span;
    }
  }
  __syncthreads();

  float* tile = tiles + wave * 16 * TLD;
  int* opix = outpix + wave * 64;
  const int ngroups = (p.M + 63) / 64;
  const int iters = (ngroups + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);
  const KResult& R0 = p.out[0];
  for (int it = 0; it < iters; ++it) {
    const int group = (it * (int)gridDim.x + (int)blockIdx.x) * 4 + wave;
    const int m0 = group * 64;
    // output pixel of row `lane` of this group (element offset of channel 0 in result 0's pixel grid, or -1)
    {
      const int m = m0 + lane;
      int pix = -1;
      if (m < p.M) {
        unsigned gx, gy;
        const unsigned t = fastdiv_dev((unsigned)m, (unsigned)p.GW, p.mGW, &gx);
        const int n = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
        const int oy = (int)gy * p.osy + ph.ooy, ox = (int)gx * p.osx + ph.oox;
        if (oy < p.OH && ox < p.OW) pix = (n * p.OH + oy) * p.OW + ox;
      }
      opix[lane] = pix;
    }
    // the four pixels this lane gathers for (one per 16-row MFMA tile)
    int pn[4], pby[4], pbx[4];
    bool plive[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + 16 * mt + col;
      plive[mt] = m < p.M;
      unsigned gx, gy;
      const unsigned t = fastdiv_dev(plive[mt] ? (unsigned)m : 0u, (unsigned)p.GW, p.mGW, &gx);
      pn[mt] = (int)fastdiv_dev(t, (unsigned)p.GH, p.mGH, &gy);
      pby[mt] = (int)gy * p.sy;
      pbx[mt] = (int)gx * p.sx;
    }
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int k4 = 0;
    for (int s = 0; s < p.n_in; ++s) {
      const KOperand& S = p.in[s];
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
      if (S.C % 16 == 0) {
        const int nblk = S.C / 16;
        int base[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) base[mt] = (pn[mt] * (int)S.sn + pby[mt] * (int)S.sh + pbx[mt] * (int)S.sw + 4 * kk) * 4;
        // (tap, channel block) steps in one flat sequence, the NEXT step's four patch loads in flight under this step's 16 matrix
        // instructions (the plain nest issued them right before their first use: every step paid a full memory latency)
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto load_step = [&](int st, f32x4 (&dst)[4]) __attribute__((always_inline)) {
          const int j = st / nblk, cb = st - j * nblk;
          const int tp = tapl[j];
          const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
          const int toff = (dy * (int)S.sh + dx * (int)S.sw) * 4 + cb * 64;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const bool ok = plive[mt] && (unsigned)(pby[mt] + dy) < (unsigned)p.IH && (unsigned)(pbx[mt] + dx) < (unsigned)p.IW;
            dst[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? base[mt] + toff : -1, 0, 0));
          }
        };
        const int nst = ntaps * nblk;
        f32x4 a4[2][4];
        load_step(0, a4[0]);
        for (int st = 0; st < nst; st += 2) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (st + h >= nst) break;
            if (st + h + 1 < nst) load_step(st + h + 1, a4[h ^ 1]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              float b[NT];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) b[nt] = wl[(size_t)(k4 + s4) * 4 * NP + kk * NP + 16 * nt + col];
#pragma unroll
              for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[h][mt][s4], b[nt], acc[mt][nt], 0, 0, 0);
            }
            k4 += 4;
          }
        }
      } else {            // 1-channel piece: lane kk takes tap 4q + kk
        for (int q4 = 0; q4 < (ntaps + 3) / 4; ++q4) {
          const int tap = q4 * 4 + kk;
          const int tp = tapl[tap < 16 ? tap : 15];
          const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
          float a[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int iy = pby[mt] + dy, ix = pbx[mt] + dx;
            const bool ok = plive[mt] && tap < ntaps && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const int off = ok ? (pn[mt] * (int)S.sn + (iy >> S.up) * (int)S.sh + (ix >> S.up) * (int)S.sw) * 4 : -1;
            a[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
          }
          float b[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b[nt] = wl[(size_t)k4 * 4 * NP + kk * NP + 16 * nt + col];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
          k4 += 1;
        }
      }
    }

    // ---- epilogue: bias + activation into this wave's 16-row tile (C/D layout: col = lane & 15, rows 4*kk + r), one MFMA row
    //      tile at a time, then float4 stores of whole pixels
    __syncthreads();                                   // opix written by all lanes of this wave (and tile free)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + col;
        const float bias = (p.bias != nullptr && n < p.Ntot) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * kk + r) * TLD + n] = thin_act(acc[mt][nt][r] + bias, p.act, p.act_p0, p.act_p1);
      }
      __syncthreads();
      constexpr int C4 = NP / 4;
      for (int e = lane; e < 16 * C4; e += 64) {
        const int row = e / C4, c4 = e - row * C4;
        const int pix = opix[16 * mt + row];
        if (pix >= 0 && 4 * c4 < p.Ntot)
          *reinterpret_cast<f32x4*>(R0.p + (long long)pix * R0.sw + 4 * c4) = *reinterpret_cast<const f32x4*>(tile + row * TLD + 4 * c4);
      }
      __syncthreads();
    }
  }
}

int launch_thin_conv(const IgemmParams& p, hipStream_t stream) {
  const int NT = p.Ntot <= 16 ? 1 : 2;
  int k4 = 0;
  for (int z = 0; z < p.nphases; ++z) {
    const int k = thin_ksteps(p, p.ph[z].ntaps);
    if (k > k4) k4 = k;
  }
  const size_t lds = thin_conv_lds(p, NT);
  const int ngroups = (p.M + 63) / 64;
  int blocks = (ngroups + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  if (NT == 1) DN_LAUNCH((thin_conv_kernel<1>), dim3(blocks, p.nphases), dim3(256), lds, stream, p, k4);
  else DN_LAUNCH((thin_conv_kernel<2>), dim3(blocks, p.nphases), dim3(256), lds, stream, p, k4);
  set_last_kernel("dn::thin_conv_kernel<%d>", NT);
  return check_launch("thin_conv_kernel");
}

}  // namespace dn
