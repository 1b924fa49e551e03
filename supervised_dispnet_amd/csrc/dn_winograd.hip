// Winograd F(2x2, 3x3) convolution on the CDNA4 matrix cores (gfx950 only): forward and input gradient of the
// 3x3 / stride 1 / pad 1 layers (the whole VGG16-BN encoder but its first layer, and the channel-aligned decoder iconvs;
// models/Disp_vgg_BN.py:84,93-105,137-186).  2.25x fewer multiply-accumulates than the direct contraction; the transforms are
// fp32 adds (and multiplications by 1/2 on the weights).  Three arithmetic variants of the 16 GEMMs (template parameter PREC,
// dn_conv_desc.compute):
//   PREC 3 (DN_COMPUTE_F32X3, the Python layer's default): every fp32 product from three exact bf16 pieces per operand, six partial
//           products on v_mfma_f32_32x32x16_bf16, fp32 accumulation -- an fp32 result at 2.7x the fp32 instruction's matrix rate;
//   PREC 0 (DN_COMPUTE_F32): exact fp32 FMAs on v_mfma_f32_32x32x2_f32 (the description below is written for this one; the others
//           differ in the main loop only);
//   PREC 1 (DN_COMPUTE_BF16): operands rounded to bf16 (opt-in mixed precision).
//
//   Y_tile(2x2) = A^T [ sum_c  U_c (.) V_c ] A      V = B^T d B   (4x4 input patch d, per channel, on the fly)
//                                                    U = G g G^T   (3x3 filter g, once per optimizer step: wino_pack_kernel)
//
// = 16 independent GEMMs  M_p[tile][cout] = sum_c V_p[tile][c] * U_p[c][cout],  p = 4i + j the position in the 4x4 transform
// domain.  One block: 32 tiles (= 128 output pixels) x 64 output channels x all 16 positions, 4 waves, TWO blocks per CU:
// wave w owns the four positions of transform row i = w, i.e. 4 x (32x64) accumulators = 128 registers next to at most 128
// others (the 64-tile / 256-accumulator / one-block-per-CU variant is kept for comparison; not dispatched since round 5).
//   A side : each thread gathers one 4x4 patch of 2 (4) channels straight from the NHWC operands (virtual concat, pending
//            BatchNorm-apply + ReLU of the producer, zero halo by the buffer bounds check), transforms it in registers and
//            writes the 16 transformed values into LDS planes [position][k half][tile][4], padded so that both the stores and
//            the per-lane ds_read_b128 fragment reads are bank-conflict free; 16-channel chunks, double-buffered, one barrier
//            per chunk.  A 1-channel piece of a concat (the upsampled disparity) is a 16-wide chunk with one live channel.
//   B side : the transformed weights never touch LDS.  They are packed in MFMA fragment order ([k/8][position][cout/32]
//            [lane][4]), so a wave's B fragment is one fully coalesced 1 KiB global load per (position, 32 couts, 8 k), issued
//            three steps ahead of its use into a 4-deep register ring; each element is read by exactly one wave of the block.
//   Output : the transform along j is done in registers, the transform along i crosses the four waves through LDS; then the
//            usual epilogue (batch-statistic partials of the pre-bias result per 32 tiles = 128 pixels, bias, activation,
//            channel-split / accumulating stores, float4 along the channels; element-wise for non-float4 results).
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "dn_internal.h"
#include "dn_wino_common.h"

namespace dn {

// ------------------------------------------------------------------------------------------------ eligibility
bool wino_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_winograd) return false;
  if (!(d->kind == DN_CONV_FWD || d->kind == DN_CONV_DGRAD)) return false;
  // the one-channel disparity heads have their own kernels, dispatched before this one (dn_conv.hip::run_conv); the packed
  // weight layout must follow the same decision (a head's input gradient has a 1-channel operand and would qualify below)
  if (!knobs().no_direct && (head_fwd_eligible(d, p) || head_dgrad_eligible(d, p))) return false;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return false;
  if (d->IH != d->OH || d->IW != d->OW || (d->OH & 1) || (d->OW & 1)) return false;
  if (p.nphases != 1 || p.ph[0].ntaps != 9) return false;
  if (p.Ntot < 64) return false;
  if ((long long)p.M * 4 >= (1ll << 31)) return false;
  {
    // a block always computes 64 tiles x 64 output channels: not worth it (and not better than the direct kernel's 64-row tiles)
    // when padding eats the 2.25x, e.g. the 12-tile deep layers of a 64x96 test image
    const long long T = p.M / 4, Tpad = (T + 31) / 32 * 32, Npad = (p.Ntot + WBN - 1) / WBN * WBN;
    if (T * p.Ntot * 100 < Tpad * Npad * 60) return false;
    // Few tiles: F(2x2,3x3) rounds 2-3x coarser than the direct FMA chain (tests/test_gpu_kernels.py::test_winograd_error_vs_fp64),
    // which the BatchNorm of a tiny map (batch statistics over a few dozen values) amplifies: such maps keep the direct kernel.
    // The floor is 192 tiles so that the 8x26 levels of a 4-image shard (208 tiles: BASELINE's b32 split over 8 GPUs) stay on
    // this path -- 56 blocks that each do 2.25x less work beat the direct kernel's 28 (0.32 -> 0.1 ms per layer).
    if (T < knobs().wino_min_tiles) return false;
  }
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    // a 1-channel piece (the upsampled disparity inside the decoder's concats, models/Disp_vgg_BN.py:176,182) rides along as a
    // sixteen-wide chunk with one live channel; everything else must be 16-aligned float4-addressable channels
    const bool scalar1 = o.C == 1 && o.small && o.scale == nullptr;
    if (!scalar1 && (!o.vec || !o.small || o.up != 0 || (o.C % WKC) != 0)) return false;
    if (o.scale != nullptr && ((reinterpret_cast<uintptr_t>(o.scale) | reinterpret_cast<uintptr_t>(o.shift)) & 15)) return false;
  }
  for (int i = 0; i < p.n_out; ++i) {
    if (!p.out[i].linear) return false;       // (results that are not float4-addressable take the element-wise store path)
  }
  if (p.bias != nullptr && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
  return true;
}

static int wino_ktot(const IgemmParams& p) {      // K axis: the operands one after the other, each padded to whole 16-channel chunks
  int k = 0;
  for (int i = 0; i < p.n_in; ++i) k += (p.in[i].C + WKC - 1) / WKC * WKC;
  return k;
}

static int wino_npad(const IgemmParams& p) { return (p.Ntot + WBN - 1) / WBN * WBN; }

// ---- input-channel split of the 4-wave three-piece kernel for small grids
constexpr int kSplitKCounterBytes = 4096;        // int counters [tile], zero between launches (self-resetting), in front of the partial tiles
int wino_splitk_choice(const IgemmParams& p) {
  if (knobs().no_wino_splitk) return 1;
  const int T = p.M / 4;
  const int blocks = ((T + 31) / 32) * (wino_npad(p) / WBN), chunks = wino_ktot(p) / WKC;
  if (blocks > knobs().wino_splitk_maxblocks || blocks > (int)(kSplitKCounterBytes / sizeof(int))) return 1;
  int ks = knobs().wino_splitk_target / blocks;  // aim at two blocks per CU ...
  if (ks > chunks / 8) ks = chunks / 8;   // ... of at least eight chunks each
  if (ks > 8) ks = 8;
  return ks < 2 ? 1 : ks;
}
size_t wino_splitk_workspace_bytes(const IgemmParams& p) {
  const int ks = wino_splitk_choice(p);
  if (ks <= 1) return 0;
  const int T = p.M / 4;
  const size_t blocks = (size_t)((T + 31) / 32) * (wino_npad(p) / WBN);
  return kSplitKCounterBytes + blocks * ks * 8 * 256 * sizeof(float) * 4;
}

// + one 16-channel chunk of slack: the kernel's B stream prefetches three steps past the last one
long long wino_packed_elems(const IgemmParams& p) { return (long long)(wino_ktot(p) + WKC) * wino_npad(p) * 16; }

// ------------------------------------------------------------------------------------------------ weight transform
// wp[k/8][pos][n/32][lane][e] = U_pos[n][k],  k = 8*(k/8) + 4*(lane >> 5) + e,  n = 32*(n/32) + (lane & 31):  exactly the
// float4 a lane feeds to four consecutive v_mfma_f32_32x32x2_f32 as the B operand (the two half-waves hold k 0-3 / 4-7).
__device__ __forceinline__ void wino_pack_body(const IgemmParams& p, const float* __restrict__ w, float* __restrict__ wp, int NS, long long total) {
  // one thread per (8-k group, 32-cout group, lane, e) = one (n, k) pair: 9 weight loads, all 16 positions written
  const long long npairs = total >> 4;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < npairs; idx += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int nsub = (int)(rest % NS), kc8 = (int)(rest / NS);
    const int k = kc8 * 8 + (lane >> 5) * 4 + e, n = nsub * 32 + (lane & 31);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = 0.f;
    // k -> (operand, channel): operands are padded to whole 16-channel chunks on the K axis
    int cc = -1, kb = 0;
#pragma unroll
    for (int s = 0; s < DN_MAX_OPERANDS; ++s) {
      if (s < p.n_in) {
        const int C = p.in[s].C;
        if (k >= kb && k < kb + C) cc = p.in[s].ch_off + (k - kb);
        kb += (C + WKC - 1) / WKC * WKC;
      }
    }
    if (n < p.Ntot && cc >= 0) {                            // (padding rows / columns and the slack chunk past the last k are zeros)
      const long long base = (p.n_is_dim0 ? ((long long)n * p.D1 + cc) : ((long long)cc * p.D1 + n)) * 9;
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
    }
    // U = G g G^T, rows of G: (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1)
    float t4[4][3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float g0 = g[0][bb], g1 = g[1][bb], g2 = g[2][bb];
      t4[0][bb] = g0;
      t4[1][bb] = 0.5f * (g0 + g1 + g2);
      t4[2][bb] = 0.5f * (g0 - g1 + g2);
      t4[3][bb] = g2;
    }
    float* dst = wp + ((long long)kc8 * 16 * NS + nsub) * 256 + lane * 4 + e;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u0 = t4[i][0], u1 = 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]), u2 = 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]), u3 = t4[i][2];
      dst[(long long)(4 * i + 0) * NS * 256] = u0;
      dst[(long long)(4 * i + 1) * NS * 256] = u1;
      dst[(long long)(4 * i + 2) * NS * 256] = u2;
      dst[(long long)(4 * i + 3) * NS * 256] = u3;
    }
  }
}

__global__ void wino_pack_kernel(const IgemmParams p, const float* __restrict__ w, float* __restrict__ wp, int NS, long long total) {
  wino_pack_body(p, w, wp, NS, total);
}

// every Winograd layer of a step in ONE launch: blockIdx.y walks the table (dn_pack_many), x is the per-entry grid-stride
__global__ void wino_pack_many_kernel(const PackEntry* __restrict__ tab) {
  const PackEntry& e = tab[blockIdx.y];
  wino_pack_body(e.p, e.w, e.wp, e.NS, e.total);
}

int launch_wino_pack_many(const PackEntry* tab_dev, int first, int n, hipStream_t stream) {
  DN_LAUNCH(wino_pack_many_kernel, dim3(2 * knobs().pack_blocks, n), dim3(256), 0, stream, tab_dev + first);
  return check_launch("wino_pack_many_kernel");
}

// bf16 fragment order: wp16[k/16][pos][n/32][piece][lane][e] = piece_s(U_pos[n][k]),  k = 16*(k/16) + 8*(lane >> 5) + e (e = 0..7),
// n = 32*(n/32) + (lane & 31): the 16 bytes a lane feeds to one v_mfma_f32_32x32x16_bf16 as the B operand.
//   NSPL = 1 (DN_COMPUTE_BF16) : piece_0 = U rounded to bf16 (nearest even)
//   NSPL = 3 (DN_COMPUTE_F32X3): U = piece_0 + piece_1 + piece_2 EXACTLY (each piece the bf16 rounding of what the previous ones left;
//                                3 x 8 significant bits hold any fp32 value)
__device__ __forceinline__ void wino_split3(float x, __bf16* pc) {
  pc[0] = (__bf16)x;
  const float r1 = x - (float)pc[0];        // exact (Sterbenz), at most 16 significant bits
  pc[1] = (__bf16)r1;
  pc[2] = (__bf16)(r1 - (float)pc[1]);      // exact, at most 8 significant bits: the conversion does not round
}

template <int NSPL>
__device__ __forceinline__ void wino_pack16_body(const IgemmParams& p, const float* __restrict__ w, float* __restrict__ wp, int NS, long long total) {
  __bf16* wp16 = reinterpret_cast<__bf16*>(wp);
  // framework tap (r * 3 + s) behind every slot (a, b) of the 3 x 3 correlation kernel: wave-uniform, once per thread (round 5: the
  // round-2 body chose each of the nine values through a 9 x 9 compare chain per pair -- 81 selects of the ~300 vector instructions)
  int src_of[9];
#pragma unroll
  for (int d = 0; d < 9; ++d) src_of[d] = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int slot = (p.tdy[t] + 1) * 3 + (p.tdx[t] + 1), so = p.tr[t] * 3 + p.ts[t];
#pragma unroll
    for (int d = 0; d < 9; ++d) src_of[d] = slot == d ? so : src_of[d];
  }
  // one thread per (n, k PAIR): two adjacent k share every store (one dword = two bf16) and every conversion (v_cvt_pk_bf16_f32 takes
  // two values) -- half the stores, conversions and address arithmetic per weight of the one-thread-per-(n, k) form (round 5)
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const long long nduos = total >> 5;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < nduos; idx += (long long)gridDim.x * blockDim.x) {
    const int e2 = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int nsub = (int)(rest % NS), kc16 = (int)(rest / NS);
    const int k0 = kc16 * 16 + (lane >> 5) * 8 + 2 * e2, n = nsub * 32 + (lane & 31);
    f32x2v g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = f32x2v{0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + h;
      int cc = -1, kb = 0;
#pragma unroll
      for (int s = 0; s < DN_MAX_OPERANDS; ++s) {
        if (s < p.n_in) {
          const int C = p.in[s].C;
          if (k >= kb && k < kb + C) cc = p.in[s].ch_off + (k - kb);
          kb += (C + WKC - 1) / WKC * WKC;
        }
      }
      if (n < p.Ntot && cc >= 0) {
        const long long base = (p.n_is_dim0 ? ((long long)n * p.D1 + cc) : ((long long)cc * p.D1 + n)) * 9;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) g[aa][bb][h] = w[base + src_of[aa * 3 + bb]];
      }
    }
    f32x2v t4[4][3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const f32x2v g0 = g[0][bb], g1 = g[1][bb], g2 = g[2][bb];
      t4[0][bb] = g0;
      t4[1][bb] = 0.5f * (g0 + g1 + g2);
      t4[2][bb] = 0.5f * (g0 - g1 + g2);
      t4[3][bb] = g2;
    }
    __bf16* dst = wp16 + ((((long long)kc16 * 16 * NS + nsub) * NSPL) * 64 + lane) * 8 + 2 * e2;
    const long long posB = (long long)NS * NSPL * 64 * 8;     // elements between two positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2v u[4];
      u[0] = t4[i][0];
      u[1] = 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]);
      u[2] = 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]);
      u[3] = t4[i][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // the same three-piece split as wino_split3, two values at a time: piece = bf16(what the previous pieces left), exactly
        const bf16x2v p0 = __builtin_convertvector(u[j], bf16x2v);
        if constexpr (NSPL == 1) {
          *reinterpret_cast<bf16x2v*>(dst + (4 * i + j) * posB) = p0;
        } else {
          const f32x2v r1 = u[j] - __builtin_convertvector(p0, f32x2v);
          const bf16x2v p1 = __builtin_convertvector(r1, bf16x2v);
          const f32x2v r2 = r1 - __builtin_convertvector(p1, f32x2v);
          const bf16x2v p2 = __builtin_convertvector(r2, bf16x2v);
          *reinterpret_cast<bf16x2v*>(dst + (4 * i + j) * posB) = p0;
          *reinterpret_cast<bf16x2v*>(dst + (4 * i + j) * posB + 512) = p1;
          *reinterpret_cast<bf16x2v*>(dst + (4 * i + j) * posB + 1024) = p2;
        }
      }
    }
  }
}

template <int NSPL>
__global__ void __launch_bounds__(256) wino_pack16_kernel(const IgemmParams p, const float* __restrict__ w, float* __restrict__ wp, int NS, long long total) {
  wino_pack16_body<NSPL>(p, w, wp, NS, total);
}

template <int NSPL>
__global__ void __launch_bounds__(256) wino_pack16_many_kernel(const PackEntry* __restrict__ tab) {
  const PackEntry& e = tab[blockIdx.y];
  wino_pack16_body<NSPL>(e.p, e.w, e.wp, e.NS, e.total);
}

int launch_wino_pack16_many(const PackEntry* tab_dev, int first, int n, int pieces, hipStream_t stream) {
  if (pieces == 3) DN_LAUNCH(wino_pack16_many_kernel<3>, dim3(2 * knobs().pack_blocks, n), dim3(256), 0, stream, tab_dev + first);
  else DN_LAUNCH(wino_pack16_many_kernel<1>, dim3(2 * knobs().pack_blocks, n), dim3(256), 0, stream, tab_dev + first);
  return check_launch("wino_pack16_many_kernel");
}

int launch_wino_pack16(const IgemmParams& p, const float* w, float* wp, int pieces, hipStream_t stream) {
  const long long total = wino_packed_elems(p);              // (n, k) pairs x 16 positions, as for the fp32 layout
  int blocks = (int)(((total >> 5) + 255) / 256);            // one thread per (n, k pair)
  if (blocks > 8192) blocks = 8192;
  if (pieces == 3) DN_LAUNCH(wino_pack16_kernel<3>, dim3(blocks), dim3(256), 0, stream, p, w, wp, wino_npad(p) / 32, total);
  else DN_LAUNCH(wino_pack16_kernel<1>, dim3(blocks), dim3(256), 0, stream, p, w, wp, wino_npad(p) / 32, total);
  return check_launch("wino_pack16_kernel");
}

// 0: not a Winograd layer; 1: fp32 Winograd; 2: bf16-multiply Winograd (descriptor compute = DN_COMPUTE_BF16); 3: fp32 products from
// three bf16 pieces per operand (DN_COMPUTE_F32X3) -- the last two on the default tile variant only
int wino_layout(const dn_conv_desc* d, const IgemmParams& p) {
  if (!wino_eligible(d, p)) return 0;
  if (knobs().wino_dbg != 0 && knobs().wino_dbg != 4 && knobs().wino_dbg < 16) return 1;
  if (p.compute == DN_COMPUTE_F32X3) return 3;          // (either tile height)
  if (1 != 1) return 1;
  return p.compute == DN_COMPUTE_BF16 ? 2 : 1;
}

// floats of the packed buffer for a given layout (layout 3 holds three bf16 per element: 1.5 floats)
long long wino_packed_floats(const IgemmParams& p, int layout) {
  const long long n = wino_packed_elems(p);
  return layout == 3 ? n + n / 2 : n;
}

int launch_wino_pack(const IgemmParams& p, const float* w, float* wp, hipStream_t stream) {
  const long long total = wino_packed_elems(p);
  int blocks = (int)(((total >> 4) + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  DN_LAUNCH(wino_pack_kernel, dim3(blocks), dim3(256), 0, stream, p, w, wp, wino_npad(p) / 32, total);
  return check_launch("wino_pack_kernel");
}

// ------------------------------------------------------------------------------------------------ the convolution
// What the matrix pipe shares (measured, tools/ubench/mfma_issue.hip): v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 FMA
// lanes, so every vector instruction a SIMD issues -- from this wave or from its partner -- takes its cycles away from the
// MFMAs (~2.4 cycles per plain fp32 op, ~8 per VOP3 v_cndmask with an SGPR mask, 13-27 per LDS store / global load).  Only
// LATENCY overlaps (memory, LDS, barriers), and only with a second wave on the SIMD.  Hence: as few vector instructions
// per MFMA as possible (hardware zero fill through buffer descriptors instead of select, no masking after the fact unless a
// BatchNorm is pending), conflict-free LDS rows, and two tile heights:
//   MTW = 2 : 64 tiles per block, 256 accumulator registers, one block per CU (least L2 traffic per MFMA),
//   MTW = 1 : 32 tiles per block, 128 accumulator registers, two blocks per CU (a partner wave covers the stalls and one
//             block's epilogue runs under the other's main loop).
// bf16-multiply / fp32-accumulate variant (descriptor compute = DN_COMPUTE_BF16, BF = true below): the transformed input tiles are
// rounded to bf16 on their way into LDS ([position][tile][16 k], 48-byte tile rows: the 16-byte fragment reads of a 16-lane group
// and the dword staging stores land on distinct banks), the transformed weights are packed as bf16 in fragment order, and one
// v_mfma_f32_32x32x16_bf16 consumes a whole 16-channel chunk of one position: 8 matrix instructions (256 cycles) per chunk
// instead of 64 (4096).  The kernel is then bound by the staging work and by HBM, not by the matrix pipe; activations, weights
// (master copy), accumulators, transforms, statistics and everything outside this kernel stay fp32.
// F32X3 variant (PREC = 3): fp32 products on the bf16 matrix cores.  Staging is the fp32 kernel's (fp32 planes in LDS, same layout,
// double-buffered, two blocks per CU); a wave reads the 8 fp32 values of its A fragment (two ds_read_b128) and splits them EXACTLY
// into three bf16 pieces in registers (x = x0 + x1 + x2: round, subtract, round, subtract -- each transformed value is consumed by
// exactly one wave, so splitting at the consumer costs the same vector work as splitting at the producer and needs neither 1.5x the
// LDS nor 48 registers of pending pieces); the weights are packed as three pieces; SIX matrix instructions per (position, 32 couts):
// x0y2, x2y0, x1y1, x1y0, x0y1, x0y0 -- 48 per chunk = 1536 cycles against the 4096 of the fp32 instruction, and unlike that one they
// leave the vector ALUs to the wave: ~6 vector instructions issue under each of them (tools/ubench/agpr_issue.hip).
template <int MTW, bool HA, int DBG, int PREC = 0>     // PREC 0: fp32 matrix instruction, 1: bf16 operands, 3: three bf16 pieces per operand
__global__ void __launch_bounds__(256, MTW == 2 ? 1 : 2) wino_conv_kernel(const IgemmParams p) {
  constexpr bool BF = PREC != 0;
  static_assert(PREC != 1 || MTW == 1, "the bf16-rounded variant exists for the two-blocks-per-CU tile only");
  using Cfg = WinoCfg<MTW>;
  constexpr int ROW16 = W16_ROWB;
  constexpr int PLANE16 = Cfg::BT * ROW16, BUF16 = 16 * PLANE16;      // PREC 1: one position plane / one 16-channel chunk
  constexpr int BT = Cfg::BT, HALFB = Cfg::HALFB, POSB = Cfg::POSB, SUBB = Cfg::SUBB, BUFB = Cfg::BUFB;
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order (see igemm_conv_u32_kernel): contiguous logical tile ranges per XCD, N tile fastest
  const int MT = (p.T + BT - 1) / BT, NT = p.Npad / WBN;
  const int per = (MT * NT + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= MT * NT) return;
  const int mb = p.nmajor ? q % MT : q / NT, nb = p.nmajor ? q / MT : q % NT;
  // Input-channel split for small grids (p.ksplit > 1, three-piece variant; launch_wino_conv): blockIdx.y = kz takes the global 16-channel
  // chunks [g0, g1) of the K axis (the operands one after the other); the partial output tiles meet in a workspace and the block that
  // arrives last sums them in index order (deterministic) and runs the epilogue.  A 4-image shard of the metric's batch leaves the deep
  // layers 32-104 blocks for 256 CUs, each walking all 32-48 chunks alone.
  int kz = 0, g0 = 0, g1 = 0x7fffffff;
  if constexpr (PREC == 3 && MTW == 1) {
    if (p.ksplit > 1) {
      kz = (int)blockIdx.y;
      const int cps = (p.ks_chunks + p.ksplit - 1) / p.ksplit;
      g0 = kz * cps;
      g1 = g0 + cps < p.ks_chunks ? g0 + cps : p.ks_chunks;
    }
  }
  long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, te1 = 0, te2 = 0;
  if (DBG & 4) t0 = clock64();

  // ---- staging role: one 4x4 patch of VW channels per thread and chunk: 64 tiles x 4 float4 groups, or 32 tiles x 8 float2
  //      groups (the narrow variant keeps the patch in 32 registers: its budget is 128 next to the 128 accumulators)
  constexpr int VW = MTW == 2 ? 4 : 2;
  typedef typename VecOf<VW>::type fV;
  const int st_tile = MTW == 2 ? (tid >> 2) : (tid >> 3), cg = MTW == 2 ? (tid & 3) : (tid & 7);
  const int k0 = cg * VW;        // first channel of this thread inside a 16-channel chunk
  unsigned pmask = 0;            // bit 4a+b: patch pixel (a, b) lies inside the image (and the tile exists)
  int pn, py, px;
  {
    const int t = mb * BT + st_tile;
    unsigned tx, ty;
    const unsigned r = fastdiv_dev(t < p.T ? (unsigned)t : 0u, (unsigned)p.TW, p.mTW, &tx);
    pn = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
    py = 2 * (int)ty - 1;
    px = 2 * (int)tx - 1;
    // outer product of a 4-bit row mask and a 4-bit column mask (the prologue's instructions are issued between the partner block's
    // MFMAs, ~40 cycles apiece: 16 compare-and-or chains were 4 k ticks of it)
    unsigned colm = 0, rowm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) colm |= ((unsigned)(px + b) < (unsigned)p.IW) ? (1u << b) : 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) rowm |= ((unsigned)(py + a) < (unsigned)p.IH) ? (1u << (4 * a)) : 0u;
    pmask = t < p.T ? rowm * colm : 0u;        // bit 4a + b = row bit a AND column bit b
  }

  f32x16 acc[4][MTW][2];          // zeroed below, AFTER the first operand's patch loads have been issued (128 moves under their latency)

  // B fragments straight from the packed weights: wcur -> this wave's slice of the current 16-k chunk
  const int NS = p.Npad / 32;
  const char* wcur = reinterpret_cast<const char*>(p.w + ((size_t)(4 * wave) * NS + 2 * nb) * 256 + lane * 4);
  const size_t wstep8B = (size_t)16 * NS * 1024;        // bytes per 8-k group (all 16 positions, all couts)
  const unsigned wjB = (unsigned)NS * 1024u;            // bytes between two positions j of one 8-k group
  // B ring: the fragments of (8-k group, position j) step g live in breg[g & 3] and are requested three steps (24 / 48 MFMAs)
  // ahead; the stream runs on across chunks and operands (the packed buffer carries one chunk of slack at its end)
  f32x4 breg[4][2];
  auto load_b = [&](int slot) {
    breg[slot][0] = *reinterpret_cast<const f32x4*>(wcur);
    breg[slot][1] = *reinterpret_cast<const f32x4*>(wcur + 1024);
  };
  auto advance_b = [&](int j_loaded) { wcur += j_loaded == 3 ? wstep8B - 3 * (size_t)wjB : (size_t)wjB; };
  const int stA = (k0 >> 3) * SUBB + ((k0 >> 2) & 1) * HALFB + st_tile * 16 + (k0 & 3) * 4;   // staging store offset in a buffer (+ pos * POSB)
  const int frA = (4 * wave) * POSB + (lane >> 5) * HALFB + (lane & 31) * 16;  // fragment read offset in an 8-k group
  // BF: staging store (one dword = channels k0, k0+1 as bf16) and fragment read (8 bf16 = k 8*(lane>>5)..+7 of tile lane&31)
  const int stA16 = st_tile * ROW16 + k0 * 2;                                    // (+ pos * PLANE16, + 32 * piece)
  const int frA16 = (4 * wave) * PLANE16 + (lane & 31) * ROW16 + (lane >> 5) * 16;
  // BF weights: wp16[chunk][pos][n/32][piece][lane][8 bf16]: 1 KiB per (chunk, position, 32 couts, piece)
  constexpr int NPC = PREC == 3 ? 3 : 1;
  const char* wcur16 = reinterpret_cast<const char*>(p.w) + ((size_t)(4 * wave) * NS + 2 * nb) * (1024 * NPC) + lane * 16;
  const size_t wchunk16B = (size_t)16 * NS * 1024 * NPC, wj16B = (size_t)NS * 1024 * NPC;
  wcur16 += (size_t)g0 * wchunk16B;                     // (split-K: the stream starts at this block's first chunk)
  bf16x8 breg16[PREC == 1 ? 4 : 1][2];
  auto load_b16 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) breg16[PREC == 1 ? j : 0][nn] = *reinterpret_cast<const bf16x8*>(wcur16 + (size_t)j * wj16B + nn * 1024);
    wcur16 += wchunk16B;
  };
  // PREC 3: weight unit u = 2 j + half of a chunk = the three pieces of one (position, 32 couts), streamed in the order they are
  // released (piece 2, 1, 0): stream index g = 3 u + (2 - piece), 24 per chunk, held in a ring of WRING piece registers (WRING divides
  // 24, so every slot of the unrolled chunk has a fixed register); the matrix instruction that releases piece g is followed by the
  // request for piece g + WRING (which may lie in the next chunk / operand / the slack chunk): WRING = 8 -> 32 registers, 14-16 matrix
  // instructions of lead
  constexpr int WRING = 8;
  bf16x8 bq[PREC == 3 ? WRING : 1];
  auto load_b3 = [&](int g) {                          // g = stream index relative to the current chunk's base pointer wcur16 (0 .. 24 + WRING)
    const int u = g / 3, piece = 2 - g % 3;
    const size_t off = (size_t)(u >> 3) * wchunk16B + (size_t)((u & 7) >> 1) * wj16B + (size_t)(u & 1) * 3072 + (size_t)piece * 1024;
    bq[PREC == 3 ? g % WRING : 0] = *reinterpret_cast<const bf16x8*>(wcur16 + off);
  };
  if constexpr (PREC == 0) {
    load_b(0);
    advance_b(0);
    load_b(1);
    advance_b(1);
    load_b(2);
    advance_b(2);
  } else if constexpr (PREC == 1) {
    load_b16();
  } else {
#pragma unroll
    for (int g = 0; g < WRING; ++g) load_b3(g);
  }

  // slot schedule (compile-time): NSL slots per 16-channel chunk, one MFMA each; H = first slot of the second 8-k group
  constexpr int NSL = 64 * MTW, H = NSL / 2, GRP = 8 * MTW;     // GRP = MFMAs per (8-k group, position j)
  constexpr int S_V = 1, S_AFF = 17, S_T = H + 8;   // patch loads | scale/shift | transform + LDS stores
  constexpr int ROWS_AT = S_T + 16, COLS_AT = ROWS_AT + 4;
  constexpr int COLS_PER_SLOT = (NSL - COLS_AT) >= 8 ? 1 : 2;

  int buf = 0, gbase = 0;
  bool first_op = true;
  for (int s = 0; s < p.n_in; ++s) {
    const KOperand& S = p.in[s];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
    const int shB = (int)S.sh * 4, swB = (int)S.sw * 4;
    const int nch_all = (S.C + WKC - 1) / WKC;
    const int c_lo = g0 > gbase ? g0 - gbase : 0, nch = (g1 - gbase) < nch_all ? (g1 - gbase) : nch_all;    // this block's chunks [c_lo, nch) of the operand
    gbase += nch_all;
    if (c_lo >= nch) continue;
    const bool scalar1 = S.C == 1;         // 1-channel piece: dword gathers (with the nearest-x2 upsample), live in channel 0 only
    const int off0 = (pn * (int)S.sn + py * (int)S.sh + px * (int)S.sw + k0) * 4;
    const bool op_aff = S.scale != nullptr;
    const char* scp = reinterpret_cast<const char*>(op_aff ? S.scale : S.p) + (op_aff ? k0 * 4 : 0);
    const char* shp = reinterpret_cast<const char*>(op_aff ? S.shift : S.p) + (op_aff ? k0 * 4 : 0);
    fV v[16], sc4, sh4;
    f32x4 fa[2][MTW];
    float relu_floor = 0.f;
    int cnB = c_lo * (WKC * 4);        // byte offset (channels) of the chunk whose loads are issued next

    auto load_v_t = [&](int i, auto sc_tag, unsigned mask) __attribute__((always_inline)) {
      constexpr bool SC1 = decltype(sc_tag)::value;
      const int a = i >> 2, b = i & 3;
      const bool ok = (mask >> i) & 1u;
      if constexpr (SC1) {
        int off = (pn * (int)S.sn + ((py + a) >> S.up) * (int)S.sh + ((px + b) >> S.up) * (int)S.sw) * 4;
        off = (ok && k0 == 0) ? off : -1;
        const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        fV t;
#pragma unroll
        for (int e = 0; e < VW; ++e) t[e] = e == 0 ? x : 0.f;
        v[i] = t;
        return;
      }
      int off = off0 + cnB + a * shB + b * swB;
      asm volatile("" : "+v"(off));
      off = ok ? off : -1;                               // past num_records: the buffer load returns zeros
      v[i] = buffer_load_vec<VW>(rsrc, off);
    };
    auto load_v = [&](int i) __attribute__((always_inline)) {
      if (scalar1) load_v_t(i, std::true_type{}, pmask);
      else load_v_t(i, std::false_type{}, pmask);
    };
    auto load_aff = [&]() {
      if constexpr (HA) {
        const fV l1 = *reinterpret_cast<const fV*>(scp + (op_aff ? cnB : 0));
        const fV l2 = *reinterpret_cast<const fV*>(shp + (op_aff ? cnB : 0));
#pragma unroll
        for (int e = 0; e < VW; ++e) {                    // an operand without a pending BatchNorm: identity, no floor
          sc4[e] = op_aff ? l1[e] : 1.f;
          sh4[e] = op_aff ? l2[e] : 0.f;
        }
        relu_floor = op_aff ? 0.f : -__builtin_huge_valf();
      }
    };
    auto affine_piece = [&](int i) {
      if constexpr (HA) {
        // clamp to [floor, cap]: floor = 0 is the ReLU, cap = 0 re-zeroes a halo pixel the BatchNorm shift lifted (one v_med3 each)
        if constexpr (PREC == 3) {
          // the vector ALUs are this variant's second resource: one packed fma for the pair, the cap from a signed bit-field extract
          // (v_bfe_i32: -1 / 0) and one AND (+inf / 0), recomputed per chunk (the asm keeps the 16 caps from being hoisted into registers)
          unsigned pm = pmask;
          asm volatile("" : "+v"(pm));
          const int msk = __builtin_amdgcn_sbfe((int)pm, i, 1);
          const float cap = __builtin_bit_cast(float, msk & 0x7f800000);
          const fV t = __builtin_elementwise_fma(v[i], sc4, sh4);
#pragma unroll
          for (int e = 0; e < VW; ++e) v[i][e] = __builtin_amdgcn_fmed3f(t[e], relu_floor, cap);
          return;
        }
        const float cap = ((pmask >> i) & 1u) ? __builtin_huge_valf() : 0.f;
#pragma unroll
        for (int e = 0; e < VW; ++e) v[i][e] = __builtin_amdgcn_fmed3f(fmaf(v[i][e], sc4[e], sh4[e]), relu_floor, cap);
      }
    };
    auto row_piece = [&](int b) {                      // B^T d, in place: rows (0,1,2,3) <- (d0-d2, d1+d2, d2-d1, d1-d3)
      const fV d0 = v[0 + b], d1 = v[4 + b], d2 = v[8 + b];
      v[0 + b] = d0 - d2;
      v[4 + b] = d1 + d2;
      v[8 + b] = d2 - d1;
      v[12 + b] = d1 - v[12 + b];
    };
    auto col_piece = [&](int b2, int i, int half) {    // (B^T d) B and the LDS stores of transform row i
      if constexpr (PREC == 1) {
        char* dst16 = smemB + b2 * BUF16 + stA16 + (4 * i) * PLANE16;
        auto put = [&](int pos, const fV& val) {
          const bf16x2 h = __builtin_convertvector(val, bf16x2);              // v_cvt_pk_bf16_f32 (round to nearest even)
          *reinterpret_cast<unsigned*>(dst16 + pos * PLANE16) = __builtin_bit_cast(unsigned, h);
        };
        if (half == 0) {
          put(0, v[4 * i + 0] - v[4 * i + 2]);
          put(1, v[4 * i + 1] + v[4 * i + 2]);
        } else {
          put(2, v[4 * i + 2] - v[4 * i + 1]);
          put(3, v[4 * i + 1] - v[4 * i + 3]);
        }
        return;
      }
      char* dst = smemB + b2 * BUFB + stA + (4 * i) * POSB;
      if (half == 0) {
        *reinterpret_cast<fV*>(dst + 0 * POSB) = v[4 * i + 0] - v[4 * i + 2];
        *reinterpret_cast<fV*>(dst + 1 * POSB) = v[4 * i + 1] + v[4 * i + 2];
      } else {
        *reinterpret_cast<fV*>(dst + 2 * POSB) = v[4 * i + 2] - v[4 * i + 1];
        *reinterpret_cast<fV*>(dst + 3 * POSB) = v[4 * i + 1] - v[4 * i + 3];
      }
    };

    // ---- pipeline fill for this operand (one exposed memory latency + transform per operand)
    {
      if constexpr (DBG & 128) {            // ablation: what a prologue without its exposed memory latency would cost (timing only)
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < VW; ++e) v[i][e] = 1.f;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) load_v(i);
      }
      load_aff();
      if (first_op) {
        first_op = false;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[j][m][nn][e] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) affine_piece(i);
#pragma unroll
      for (int b = 0; b < 4; ++b) row_piece(b);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        col_piece(buf, i, 0);
        col_piece(buf, i, 1);
      }
    }
    __syncthreads();
    if (DBG & 4) t1 = clock64();

    if constexpr (PREC == 3) {
      // One chunk = 48 matrix instructions (slot m = 12 j + 2 t + half); the side work of a slot is what the wave issues while that
      // instruction executes: the fp32 kernel's staging of the NEXT chunk (patch loads, pending BatchNorm, transform, LDS stores) and
      // the split of the next position's A fragment.
      const int frA3 = (4 * wave) * POSB + (lane >> 5) * SUBB + (lane & 31) * 16;   // lane (tile, g): k 8g..8g+7 = 8-k group g, both halves
      f32x4 raw[MTW][2];
      bf16x8 fa3[2][MTW][3];
      auto read_raw = [&](const char* Ab, int j) {
#pragma unroll
        for (int mm = 0; mm < MTW; ++mm) {
          raw[mm][0] = *reinterpret_cast<const f32x4*>(Ab + j * POSB + mm * 512);
          raw[mm][1] = *reinterpret_cast<const f32x4*>(Ab + j * POSB + HALFB + mm * 512);
        }
      };
      auto split_pair = [&](int slot, int mm, int q) {  // channels 2q, 2q+1 of the fragment of tile half mm: x = h + m + l exactly
        const f32x2 x = f32x2{raw[mm][q >> 1][2 * (q & 1)], raw[mm][q >> 1][2 * (q & 1) + 1]};
        if constexpr (DBG & 16) {                       // ablation (timing only, wrong results): no split arithmetic
          const bf16x2 h = __builtin_convertvector(x, bf16x2);
          fa3[slot][mm][0][2 * q] = h[0]; fa3[slot][mm][0][2 * q + 1] = h[1];
          fa3[slot][mm][1][2 * q] = h[1]; fa3[slot][mm][1][2 * q + 1] = h[0];
          fa3[slot][mm][2][2 * q] = h[0]; fa3[slot][mm][2][2 * q + 1] = h[0];
          return;
        }
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        fa3[slot][mm][0][2 * q] = h[0]; fa3[slot][mm][0][2 * q + 1] = h[1];
        fa3[slot][mm][1][2 * q] = m[0]; fa3[slot][mm][1][2 * q + 1] = m[1];
        fa3[slot][mm][2][2 * q] = l[0]; fa3[slot][mm][2][2 * q + 1] = l[1];
      };
      // (a 1-channel piece is a single chunk: what the loop "re-fetches" for it is never used, so its loads are masked off instead of
      //  carrying the gather path's instruction stream and branches through every slot)
      const unsigned lmask = scalar1 ? 0u : pmask;
      for (int c = c_lo; c < nch; ++c) {
        const bool more = c + 1 < nch;
        cnB = (more ? c + 1 : c) * (WKC * 4);            // the last chunk re-fetches itself into the idle buffer: no branch
        const char* Ab = smemB + buf * BUFB + frA3;
        read_raw(Ab, 0);
#pragma unroll
        for (int mm = 0; mm < MTW; ++mm)
#pragma unroll
          for (int q = 0; q < 4; ++q) split_pair(0, mm, q);
        __builtin_amdgcn_sched_barrier(0);
        // slot m = SL j + 6 MTW half + MTW t + tile half: the MTW tile halves of a product share its weight piece
        constexpr int SL = 12 * MTW;
        static_for<4 * SL>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          constexpr int j = m / SL, q12 = m % SL, nn = q12 / (6 * MTW), r6 = q12 % (6 * MTW), t = r6 / MTW, mm = r6 % MTW, u = 2 * j + nn;   // u: weight unit
          // x0y2, x0y1, x1y1, x0y0, x1y0, x2y0: the weight pieces are released in the order 2, 1, 0
          // (a chain of dependent matrix instructions on one accumulator issues at the full rate: tools/ubench/agpr_issue.hip)
          constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
          acc[j][mm][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[j & 1][mm][AS[t]], bq[(3 * u + 2 - BS[t]) % WRING], acc[j][mm][nn], 0, 0, 0);
          // ---- side work of this slot
          if constexpr (!(DBG & 32) && mm == 0) {                  // (DBG 32: ablation without the weight stream)
            if constexpr (t == 1) load_b3(3 * u + WRING);          // piece 2 of this unit was released by the instructions t = 0
            if constexpr (t == 3) load_b3(3 * u + 1 + WRING);      // piece 1 by t = 2
            if constexpr (t == 0 && m > 0) load_b3(3 * (u - 1) + 2 + WRING);     // piece 0 of the previous unit by its last instruction
          }
          if constexpr (j < 3 && q12 == 1) read_raw(Ab, j + 1);
          if constexpr (j < 3 && q12 >= 6 * MTW && q12 < 10 * MTW) split_pair((j + 1) & 1, (q12 - 6 * MTW) / 4, (q12 - 6 * MTW) % 4);
          constexpr bool STG = !(DBG & 64);                       // (DBG 64: ablation without the staging of the next chunk)
          if constexpr (MTW == 1 && !(DBG & 256)) {
            // round 3 (measured on the 8-wave form, dn_winograd8.hip): the 16 patch loads on every OTHER slot instead of back to back --
            // a wave whose load finds the vector-memory queue full stalls in order, with its matrix instructions behind it --, the
            // clamp / row / column transform + stores packed into the last fourteen slots
            if constexpr (STG && m >= 1 && m < 33 && (m & 1)) load_v_t((m - 1) / 2, std::false_type{}, lmask);
            if constexpr (STG && m == 2) load_aff();
            if constexpr (STG && m >= 34 && m < 38) {
#pragma unroll
              for (int u4 = 0; u4 < 4; ++u4) affine_piece(4 * (m - 34) + u4);
            }
            if constexpr (STG && m >= 38 && m < 40) {
              row_piece(2 * (m - 38));
              row_piece(2 * (m - 38) + 1);
            }
            if constexpr (STG && m >= 40 && m < 48) col_piece(buf ^ 1, (m - 40) / 2, (m - 40) % 2);
          } else {
          if constexpr (STG && m >= 2 && m < 18) load_v_t(m - 2, std::false_type{}, lmask);
          if constexpr (STG && m == 18) load_aff();
          // transform + stores of the next chunk: 16 affine pieces, 4 row pieces, 8 column pieces from slot 22 MTW on
          constexpr int A0 = 22 * MTW, R0 = A0 + 8 * MTW, C0 = R0 + 4;
          if constexpr (STG && m >= A0 && m < R0) {
            if constexpr (MTW == 1) {
              affine_piece(2 * (m - A0));
              affine_piece(2 * (m - A0) + 1);
            } else {
              affine_piece(m - A0);
            }
          }
          if constexpr (STG && m >= R0 && m < R0 + 4) row_piece(m - R0);
          if constexpr (STG && m >= C0 && m < C0 + 8) col_piece(buf ^ 1, (m - C0) / 2, (m - C0) % 2);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!(DBG & 32)) load_b3(3 * 7 + 2 + WRING);      // successor of the last unit's piece 0
        wcur16 += wchunk16B;
        __syncthreads();
        buf ^= 1;
      }
      continue;
    }
    if constexpr (PREC == 1) {
      for (int c = c_lo; c < nch; ++c) {
        const bool more = c + 1 < nch;
        cnB = (more ? c + 1 : c) * (WKC * 4);          // the last chunk re-fetches itself into the idle buffer: no branch
        const char* Ab16 = smemB + buf * BUF16 + frA16;
        bf16x8 fa16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fa16[j] = *reinterpret_cast<const bf16x8*>(Ab16 + j * PLANE16);
#pragma unroll
        for (int i = 0; i < 16; ++i) load_v(i);        // next chunk's patch: in flight under the MFMAs and the weight loads
        load_aff();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
            acc[j][0][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa16[j], breg16[j][nn], acc[j][0][nn], 0, 0, 0);
        load_b16();                                     // next chunk's weights (the buffer carries one chunk of slack)
#pragma unroll
        for (int i = 0; i < 16; ++i) affine_piece(i);
#pragma unroll
        for (int b = 0; b < 4; ++b) row_piece(b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          col_piece(buf ^ 1, i, 0);
          col_piece(buf ^ 1, i, 1);
        }
        __syncthreads();
        buf ^= 1;
      }
      continue;
    }
    for (int c = c_lo; c < nch; ++c) {
      const bool more = c + 1 < nch;
      cnB = (more ? c + 1 : c) * (WKC * 4);            // the last chunk re-fetches itself into the idle buffer: no branch
      const char* Ab = smemB + buf * BUFB + frA;
#pragma unroll
      for (int mm = 0; mm < MTW; ++mm) fa[0][mm] = *reinterpret_cast<const f32x4*>(Ab + mm * 512);
      auto body = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSL>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          constexpr int grp = m / GRP, qq = m % GRP;          // grp = (8-k group, position j)
          constexpr int sub = grp / 4, j = grp % 4;
          constexpr int kk = qq / (2 * MTW), mm = (qq / 2) % MTW, nn = qq % 2;
          constexpr int cur = grp & 1, nxt = cur ^ 1;
          acc[j][mm][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mm][kk], breg[grp & 3][nn][kk], acc[j][mm][nn], 0, 0, 0);
          // ---- side work of this slot
          if constexpr (grp < 7 && qq >= 2 && qq < 2 + MTW) {
            constexpr int g1 = grp + 1;
            fa[nxt][qq - 2] = *reinterpret_cast<const f32x4*>(Ab + (g1 / 4) * SUBB + (g1 % 4) * POSB + (qq - 2) * 512);
          }
          if constexpr (qq == 0) {            // B fragments three steps ahead (step grp + 3 has j = (grp + 3) & 3)
            load_b((grp + 3) & 3);
            advance_b((grp + 3) & 3);
          }
          if constexpr (STAGE && m >= S_V && m < S_V + 16) load_v(m - S_V);
          if constexpr (STAGE && m == S_AFF) load_aff();
          if constexpr (STAGE && m >= S_T && m < S_T + 16) affine_piece(m - S_T);
          if constexpr (STAGE && m >= ROWS_AT && m < ROWS_AT + 4) row_piece(m - ROWS_AT);
          if constexpr (STAGE && m >= COLS_AT && m < COLS_AT + 8 / COLS_PER_SLOT) {
#pragma unroll
            for (int u = 0; u < COLS_PER_SLOT; ++u) {
              constexpr int dummy = 0;
              const int piece = (m - COLS_AT) * COLS_PER_SLOT + u + dummy;
              col_piece(buf ^ 1, piece / 2, piece % 2);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      body(std::true_type{});
      __syncthreads();
      buf ^= 1;
    }
  }

  if (DBG & 4) t2 = clock64();
  // ---- output transform.  Along j in registers:  Z[b] = sum_j A^T[b][j] M[i][j],  A^T = (1 1 1 0 / 0 1 -1 -1)
  f32x16 Z[2][MTW][2];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      Z[0][m][nn] = acc[0][m][nn] + acc[1][m][nn] + acc[2][m][nn];
      Z[1][m][nn] = acc[1][m][nn] - acc[2][m][nn] - acc[3][m][nn];
    }
  // Along i across the four waves through LDS (one half b at a time: 4 x BT x 72 floats)
  constexpr int NK = 2 * MTW;                      // final role: couts 4*c4..+3 of tiles tg + 16*k, k = 0..NK-1
  const int c4 = tid & 15, tg = tid >> 4;
  f32x4 Y[NK][2][2];                               // [k][a][b]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          smem[(wave * BT + row) * WZLD + 32 * nn + (lane & 31)] = Z[b][m][nn][r];
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int tile = tg + 16 * k;
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(smem + (0 * BT + tile) * WZLD + 4 * c4);
      const f32x4 z1 = *reinterpret_cast<const f32x4*>(smem + (1 * BT + tile) * WZLD + 4 * c4);
      const f32x4 z2 = *reinterpret_cast<const f32x4*>(smem + (2 * BT + tile) * WZLD + 4 * c4);
      const f32x4 z3 = *reinterpret_cast<const f32x4*>(smem + (3 * BT + tile) * WZLD + 4 * c4);
      Y[k][0][b] = z0 + z1 + z2;
      Y[k][1][b] = z1 - z2 - z3;
    }
  }

  if constexpr (PREC == 3 && MTW == 1) {
    if (p.ksplit > 1) {
      // partial tile -> lane-private float4 slots [tile q][split][8][thread] of the workspace; the last arrival sums all splits in index order
      __shared__ int ks_last;
      int* cnt = reinterpret_cast<int*>(p.ks_ws);
      f32x4* slots = reinterpret_cast<f32x4*>(p.ks_ws + p.ks_cnt_floats);
      f32x4* mine = slots + ((size_t)(q * p.ksplit + kz) * 8) * 256 + tid;
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) mine[(size_t)((k * 2 + a) * 2 + b) * 256] = Y[k][a][b];
      // Visibility without an agent-scope release / acquire (whose L2 write-back + invalidate of a cache full of other blocks' results
      // doubled the kernel's time): every split of a tile runs on the SAME XCD -- the linear block id is blockIdx.x + gridDim.x * kz with
      // gridDim.x a multiple of 8 -- so the partial tiles only have to reach that XCD's L2: the stores are write-through (waited for
      // with vmcnt(0)), the counter is an L2 atomic, and the reader's loads bypass its CU's L1 (glc).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) ks_last = (atomicAdd(cnt + q, 1) == p.ksplit - 1) ? 1 : 0;
      __syncthreads();
      if (!ks_last) return;
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) Y[k][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(slots + (size_t)q * p.ksplit * 8 * 256), 0, 0x7fffffff, 0x00020000);
      for (int z = 0; z < p.ksplit; ++z) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              typedef int i32x4 __attribute__((ext_vector_type(4)));
              const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(((z * 8 + (k * 2 + a) * 2 + b) * 256 + tid) * 16), 0, 1 /* glc */);
              Y[k][a][b] += __builtin_bit_cast(f32x4, v);
            }
      }
      if (tid == 0) cnt[q] = 0;                          // (self-resetting: the workspace is reusable by the next launch on this stream)
    }
  }

  if (DBG & 4) te1 = clock64();
  // ---- batch statistics of the pre-bias result: partial row MTW*mb + m covers tiles [32(MTW*mb+m), +32) = 128 pixels;
  //      (sum, M2 about the group's own mean), merged by dn_bn_finalize
  const int n_first = nb * WBN + 4 * c4;
  if (p.bn_partial != nullptr) {
    float* red = smem + 4 * BT * WZLD;             // [MTW][4 waves][64 couts], then gmean [MTW][64]
    float* gmean = red + MTW * 4 * 64;
    int gcount[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      int left = p.T - (mb * BT + 32 * m);
      gcount[m] = left < 0 ? 0 : (left > 32 ? 32 : left);
    }
    f32x4 s[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
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
    if (lane < 16) {
#pragma unroll
      for (int m = 0; m < MTW; ++m) *reinterpret_cast<f32x4*>(red + (m * 4 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < 64 * MTW) {
      const int m = tid >> 6, col = tid & 63;
      tot = red[(m * 4 + 0) * 64 + col] + red[(m * 4 + 1) * 64 + col] + red[(m * 4 + 2) * 64 + col] + red[(m * 4 + 3) * 64 + col];
      const int gc = m == 0 ? gcount[0] : gcount[MTW - 1];
      gmean[tid] = gc > 0 ? tot / (float)(4 * gc) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
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
      for (int m = 0; m < MTW; ++m) *reinterpret_cast<f32x4*>(red + (m * 4 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    if (tid < 64 * MTW) {
      const int m = tid >> 6, col = tid & 63;
      const float m2 = red[(m * 4 + 0) * 64 + col] + red[(m * 4 + 1) * 64 + col] + red[(m * 4 + 2) * 64 + col] + red[(m * 4 + 3) * 64 + col];
      const int n = nb * WBN + col;
      const int gc = m == 0 ? gcount[0] : gcount[MTW - 1];
      if (n < p.Ntot && gc > 0) {
        float* dst = p.bn_partial + ((long long)(MTW * mb + m) * p.Ntot + n) * 2;
        fold_store(dst, tot);          // (agent scope: the block that arrives last may read them, dn_fold.h)
        fold_store(dst + 1, m2);
      }
    }
  }

  if (p.bnb_y != nullptr)                // (input gradient: the BatchNorm backward's column sums of the layer below)
    wino_bn_bwd_sums<NK, 16, 4, MTW>(p, Y, mb, tg, c4, wave, lane, tid, n_first, smem + 4 * BT * WZLD);

  if (DBG & 4) te2 = clock64();
  // ---- bias, activation, channel-split / accumulating stores: float4 along the channels when the four columns lie in one
  //      float4-addressable result, element-wise otherwise (the 1-channel disparity piece of a concat's input gradient)
  // Every vector instruction of this phase is issued between the PARTNER block's MFMAs (one slot per 64 cycles while it is in its main
  // loop -- measured with DN_WINO_DBG=4/12: 11.4 k of the epilogue's 14 k ticks were spent here, with or without the stores), so the
  // common case is kept to a minimum of instructions: one address per tile, constant offsets for its four pixels, the result
  // selection / activation / accumulate decisions taken once, outside the loops.
  if (n_first < p.Ntot) {
    int seg = 0;
    if (p.n_out > 1 && n_first >= p.out[1].n_begin) seg = 1;
    if (p.n_out > 2 && n_first >= p.out[2].n_begin) seg = 2;
    // (field-wise selects instead of p.out[seg]: a per-lane index into the kernel arguments costs a round trip to memory)
    float* Rp = seg == 0 ? p.out[0].p : (seg == 1 ? p.out[1].p : p.out[2].p);
    const long long sw = seg == 0 ? p.out[0].sw : (seg == 1 ? p.out[1].sw : p.out[2].sw);
    const int Rbeg = seg == 0 ? p.out[0].n_begin : (seg == 1 ? p.out[1].n_begin : p.out[2].n_begin);
    const int RC = seg == 0 ? p.out[0].C : (seg == 1 ? p.out[1].C : p.out[2].C);
    const bool accumulate = (seg == 0 ? p.out[0].accumulate : (seg == 1 ? p.out[1].accumulate : p.out[2].accumulate)) != 0;
    const bool fast4 = (n_first + 3 < Rbeg + RC) && ((Rbeg | RC) & 3) == 0 && (sw & 3) == 0 && (reinterpret_cast<uintptr_t>(Rp) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    float* obase = Rp + (n_first - Rbeg);
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
      if (fast4) bias = *reinterpret_cast<const f32x4*>(p.bias + n_first);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = n_first + e < p.Ntot ? p.bias[n_first + e] : 0.f;
      }
    }
    const bool plain = p.act == DN_ACT_NONE;
    if (fast4) {
      const long long rowB = (long long)p.OW * sw;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int t = mb * BT + tg + 16 * k;
        if (t < p.T) {
          unsigned tx, ty;
          const unsigned r = fastdiv_dev((unsigned)t, (unsigned)p.TW, p.mTW, &tx);
          const int n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
          float* o00 = obase + (((long long)n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx) * sw;
          f32x4 v[2][2];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              v[a][b] = Y[k][a][b] + bias;
              if (!plain) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[a][b][e] = wino_act(v[a][b][e], p.act, p.act_p0, p.act_p1);
              }
            }
          if (accumulate) {                 // second writer of a skip connection: the four loads go out together
            const f32x4 g00 = *reinterpret_cast<const f32x4*>(o00), g01 = *reinterpret_cast<const f32x4*>(o00 + sw);
            const f32x4 g10 = *reinterpret_cast<const f32x4*>(o00 + rowB), g11 = *reinterpret_cast<const f32x4*>(o00 + rowB + sw);
            v[0][0] += g00;
            v[0][1] += g01;
            v[1][0] += g10;
            v[1][1] += g11;
          }
          if (!(DBG & 8) || v[0][0][0] == 12345.678f) {      // DBG 8: ablation without the stores
            *reinterpret_cast<f32x4*>(o00) = v[0][0];
            *reinterpret_cast<f32x4*>(o00 + sw) = v[0][1];
            *reinterpret_cast<f32x4*>(o00 + rowB) = v[1][0];
            *reinterpret_cast<f32x4*>(o00 + rowB + sw) = v[1][1];
          }
        }
      }
    } else {
      // element-wise: the four columns straddle results or are not float4-addressable (the 1-channel disparity piece of a concat's
      // input gradient)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int t = mb * BT + tg + 16 * k;
        if (t < p.T) {
          unsigned tx, ty;
          const unsigned r = fastdiv_dev((unsigned)t, (unsigned)p.TW, p.mTW, &tx);
          const int n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
          const long long pix0 = ((long long)n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const long long pix = pix0 + (long long)a * p.OW + b;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int col = n_first + e;
                if (col < p.Ntot) {
                  const float val = wino_act(Y[k][a][b][e] + bias[e], p.act, p.act_p0, p.act_p1);
                  int sg = 0;
                  if (p.n_out > 1 && col >= p.out[1].n_begin) sg = 1;
                  if (p.n_out > 2 && col >= p.out[2].n_begin) sg = 2;
                  const KResult& Q = p.out[sg];
                  float* o = Q.p + pix * Q.sw + (col - Q.n_begin);
                  *o = Q.accumulate ? *o + val : val;
                }
              }
            }
        }
      }
    }
  }
  if (DBG & 4) {
    t3 = clock64();
    if (tid == 0) {
      long long* o = reinterpret_cast<long long*>(p.ws) + (size_t)blockIdx.x * 8;
      o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = te1; o[5] = te2;
    }
  }
  if constexpr (DBG == 0) {
    if (p.fold_bn | p.fold_bnb) {
      // (tile coordinates recomputed from a laundered block index: keeping nb / MT alive across the whole kernel for this tail cost the
      //  variant with a pending BatchNorm five spilled registers)
      __syncthreads();                   // (the LDS of the epilogue is free from here)
      unsigned bx = blockIdx.x;
      asm volatile("" : "+s"(bx));
      const int MT2 = (p.T + BT - 1) / BT, NT2 = p.Npad / WBN;
      const int per2 = (MT2 * NT2 + 7) >> 3;
      const int q2 = (int)(bx & 7u) * per2 + (int)(bx >> 3);
      wino_fold_tail(p, p.nmajor ? q2 / MT2 : q2 % NT2, MT2, smem, (int)threadIdx.x);
    }
  }
}

// Whether a launch finishes the BatchNorm statistics (forward) / the BatchNorm-backward sums (input gradient) in its last-arriving
// blocks: asked for by the caller, few enough partial rows for one block per 64 channels, counters available.
bool wino_folds_bn_finalize(const IgemmParams& p) {
  return p.bnf.scale != nullptr && p.bn_partial != nullptr && p.fold_cnt != nullptr && (p.T + 31) / 32 <= kFoldMaxRows &&
         wino_npad(p) / WBN <= kFoldCounters && knobs().wino_dbg == 0;
}
bool wino_folds_bn_sums(const IgemmParams& p) {
  return p.bnb_dgamma != nullptr && p.bnb_dbeta != nullptr && p.bnb_partial != nullptr && p.fold_cnt != nullptr && (p.T + 31) / 32 <= kFoldMaxRows &&
         wino_npad(p) / WBN <= kFoldCounters && knobs().wino_dbg == 0;
}

template <int MTW, bool HA, int DBG, int PREC = 0>
static int launch_wino_variant(const IgemmParams& p, hipStream_t stream) {
  using Cfg = WinoCfg<MTW>;
  auto kernel = wino_conv_kernel<MTW, HA, DBG, PREC>;
  // PREC 1: two 16-channel chunks of bf16 planes, or the epilogue's cross-wave exchange + statistics scratch, whichever is larger
  constexpr size_t lds16a = (size_t)2 * 16 * Cfg::BT * W16_ROWB;
  constexpr size_t lds16b = (size_t)(4 * Cfg::BT * WZLD + MTW * 8 * 64) * sizeof(float);
  constexpr size_t lds = PREC == 1 ? (lds16a > lds16b ? lds16a : lds16b) : Cfg::LDS;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino_conv_kernel, %zu): %s", lds, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  const int tiles = ((p.T + Cfg::BT - 1) / Cfg::BT) * (p.Npad / WBN);
  dim3 grid((tiles + 7) / 8 * 8, (PREC == 3 && MTW == 1 && p.ksplit > 1) ? p.ksplit : 1);
  DN_LAUNCH(kernel, grid, dim3(256), lds, stream, p);
  set_last_kernel("dn::wino_conv_kernel<%d, %s, %d, %d>", MTW, HA ? "true" : "false", DBG, PREC);
  return check_launch("wino_conv_kernel");
}

int launch_wino_conv(IgemmParams& p, hipStream_t stream) {
  p.Npad = wino_npad(p);
  p.T = p.M / 4;
  p.TH = p.OH / 2;
  p.TW = p.OW / 2;
  p.mTW = fastdiv_magic((unsigned)p.TW);
  p.mTH = fastdiv_magic((unsigned)p.TH);
  {   // tile order within an XCD (DN_WINO_NMAJOR: 0 cout slice fastest, 1 tile row fastest, 2/3: tile row fastest from 4 / 8 cout slices).
      // Three-piece kernels only: the bf16-rounded and fp32-instruction variants came out 0.2-0.6 % slower with it (profiles/r05_exp28)
    const int nm = knobs().wino_nmajor, nt = p.Npad / WBN;
    p.nmajor = p.compute == DN_COMPUTE_F32X3 && (nm == 1 || (nm == 2 && nt >= 4) || (nm == 3 && nt >= 8));
  }
  p.fold_bn = wino_folds_bn_finalize(p) ? 1 : 0;
  p.fold_bnb = wino_folds_bn_sums(p) ? 1 : 0;
  const int dbg = knobs().wino_dbg;
  if (p.compute == DN_COMPUTE_BF16)      // (wino_layout() has checked that the bf16 variants may be used)
    return p.any_affine ? launch_wino_variant<1, true, 0, 1>(p, stream) : launch_wino_variant<1, false, 0, 1>(p, stream);
  p.ksplit = 1;
  if (p.compute == DN_COMPUTE_F32X3) {
    if ((knobs().wino_dbg == 0 || (knobs().wino_dbg & 4)) && wino8_wanted(p)) return launch_wino_conv8(p, stream);
    if (knobs().wino_dbg == 0) {
      // few blocks, long K: split the input channels.  Needs the caller's workspace (dn_conv_desc.splitk_ws)
      const int ks = wino_splitk_choice(p);
      if (ks > 1 && p.ks_ws != nullptr && p.ks_ws_bytes >= wino_splitk_workspace_bytes(p)) {
        p.ksplit = ks;
        p.ks_chunks = wino_ktot(p) / WKC;
        p.ks_cnt_floats = kSplitKCounterBytes / 4;
      }
    }
    // (the 64-tile / one-block-per-CU form of this variant -- the loop below is written for either tile height -- moves 37 % fewer bytes
    //  through the texture addresser, the weight pieces being fetched once per 64 tiles, and was measured 5-10 % SLOWER on every layer
    //  but one: a single wave per SIMD stalls on every wait)
    if (knobs().wino_dbg == 4) {           // in-kernel timestamps (tools/wino_timing.py)
      p.ws = reinterpret_cast<float*>(knobs().wino_dbgptr);
      return p.any_affine ? launch_wino_variant<1, true, 4, 3>(p, stream) : launch_wino_variant<1, false, 4, 3>(p, stream);
    }
    switch (knobs().wino_dbg) {            // 16 / 32 / 64 / 112: timing ablations of the three-piece variant (wrong results)
      case 16: return p.any_affine ? launch_wino_variant<1, true, 16, 3>(p, stream) : launch_wino_variant<1, false, 16, 3>(p, stream);
      case 32: return p.any_affine ? launch_wino_variant<1, true, 32, 3>(p, stream) : launch_wino_variant<1, false, 32, 3>(p, stream);
      case 64: return p.any_affine ? launch_wino_variant<1, true, 64, 3>(p, stream) : launch_wino_variant<1, false, 64, 3>(p, stream);
      case 112: return p.any_affine ? launch_wino_variant<1, true, 112, 3>(p, stream) : launch_wino_variant<1, false, 112, 3>(p, stream);
      case 128: return p.any_affine ? launch_wino_variant<1, true, 128, 3>(p, stream) : launch_wino_variant<1, false, 128, 3>(p, stream);
      default: return p.any_affine ? launch_wino_variant<1, true, 0, 3>(p, stream) : launch_wino_variant<1, false, 0, 3>(p, stream);
    }
  }
  if (dbg == 12) {                       // timestamps + no result stores (ablation, tools/wino_timing.py 12)
    p.ws = reinterpret_cast<float*>(knobs().wino_dbgptr);
    return p.any_affine ? launch_wino_variant<1, true, 12>(p, stream) : launch_wino_variant<1, false, 12>(p, stream);
  }
  if (dbg == 4) {
    p.ws = reinterpret_cast<float*>(knobs().wino_dbgptr);
    return p.any_affine ? launch_wino_variant<1, true, 4>(p, stream) : launch_wino_variant<1, false, 4>(p, stream);
  }
  return p.any_affine ? launch_wino_variant<1, true, 0>(p, stream) : launch_wino_variant<1, false, 0>(p, stream);
}

}  // namespace dn
