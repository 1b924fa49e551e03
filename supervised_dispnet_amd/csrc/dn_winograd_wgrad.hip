// Winograd F(2x2, 3x3) weight gradient of the 3x3 / stride 1 / pad 1 layers on the CDNA4 matrix cores (gfx950 only):
// wino_wgrad_kernel (fp32 matrix instruction, DN_COMPUTE_F32 / _BF16) and wino_wgrad_x3_kernel (fp32 products from three exact bf16
// pieces per operand on v_mfma_f32_32x32x16_bf16, DN_COMPUTE_F32X3 = the Python layer's default; described where it is defined).
//
//   dU_p[co][ci] = sum over tiles t of  Gt_p[t][co] * V_p[t][ci]        p = 4i + j, the 16 positions of the transform domain
//   Gt = A dY A^T  (2x2 output-gradient tile -> 4x4),   V = B^T d B  (4x4 input patch -> 4x4, the forward's input transform)
//   dW(3x3) = G^T dU G                                                  (wino_wgrad_reduce_kernel, with the split sum)
//
// 16 GEMMs whose reduction runs over the TILES: 2.25x fewer multiply-accumulates than sum over pixels of dy * x.  Both
// operands are transformed on the fly, so the staging work per MFMA is what decides (fp32 MFMAs share the SIMD's FMA lanes
// with every vector instruction, dn_winograd.hip): one block = 64 co x 64 ci x 16 positions, 8 waves (two per SIMD, 128
// accumulator registers each: wave w owns positions 4(w>>1) + 2(w&1) + {0,1}), 8 tiles per chunk.
//   * LDS: [buffer][G | V][position][tile][64 channels]: staging stores are b128 along the channels (a wave writes 1 KiB
//     contiguous), fragment reads are b32 with the 32 lanes of a half-wave on 32 consecutive channels (conflict-free both).
//   * Staging is spread in time and over the waves: waves 0-3 prepare the even chunks, waves 4-7 the odd ones; a wave issues
//     its global loads while chunk c is multiplied, transforms and stores them while chunk c+1 is, and the data is consumed as
//     chunk c+2.  Two waves of each group take input patches (16 loads, BatchNorm-apply + ReLU, 32 adds), two take
//     output-gradient tiles (4 loads, 12 adds); the groups swap these roles so that every SIMD carries one of each.
//   * Zero halo / tile tail through buffer descriptors (loads past num_records return 0); the sign of A's last row is
//     folded into the final 16 -> 9 transform instead of negating in the loop.
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "dn_internal.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void wg_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  wg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int GTC = 8;                        // tiles per chunk
constexpr int G_ROWB = 64 * 4;                // one tile row: 64 channels
constexpr int G_PLANE = GTC * G_ROWB;         // one position: [8 tiles][64 channels] = 2 KiB
constexpr int G_OPB = 16 * G_PLANE;           // one operand (G or V) of one buffer = 32 KiB
constexpr int G_BUFB = 2 * G_OPB;             // G then V
constexpr size_t kWgLds = (size_t)2 * G_BUFB; // 128 KiB

__device__ __forceinline__ f32x4 wg_buffer_load(__amdgpu_buffer_rsrc_t r, int voffset) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

bool wino_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p) {
  if (knobs().no_winograd || knobs().no_winograd_wgrad) return false;
  if (d->kind != DN_CONV_FWD) return false;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->dilation > 1) return false;
  if (d->IH != d->OH || d->IW != d->OW || (d->OH & 1) || (d->OW & 1)) return false;
  if (p.ph[0].ntaps != 9) return false;
  if (p.Ntot % 64 != 0) return false;
  if ((long long)p.M * p.Ntot * 4 + 64 >= (1ll << 31)) return false;
  if (p.M / 4 < 256) return false;
  for (int i = 0; i < p.n_in; ++i) {
    const KOperand& o = p.in[i];
    if (!o.vec || !o.small || o.up != 0 || (o.C % 64) != 0) return false;
    if (o.scale != nullptr && ((reinterpret_cast<uintptr_t>(o.scale) | reinterpret_cast<uintptr_t>(o.shift)) & 15)) return false;
  }
  return true;
}

static int wg_ktot(const IgemmParams& p) {
  int k = 0;
  for (int i = 0; i < p.n_in; ++i) k += p.in[i].C;
  return k;
}

// pixel (tile) splits: whole rounds of 256 resident blocks, at least 16 chunks per split
// Three-piece kernel on the longest reductions (>= 2^18 tiles = 1 M pixels: the 64 -> 64 layer of a 32 x 128 x 416 batch): start at THREE
// rounds.  v_mfma_f32_32x32x16_bf16 rounds its running fp32 sum more often per 8-tile chunk (three instructions, six partial products per
// tile) than the fp32 instruction does (8 FMAs); over 208 chunks per split the accumulated rounding measured 2.1x the fp32-instruction
// kernel's against fp64 (2.0e-6 vs 9.6e-7 relative L2; PyTorch-CPU fp32: 2.2e-6).  A third of the chain (the slab sum that follows is a
// short fixed-order tree) brings it to 1.3x for +0.07 ms on that one layer (tests/test_gpu_f32x3_fp64.py, profiles/r03_wg_rounds.txt).
// A mid-split flush of the accumulators inside one block (same arithmetic, no extra blocks) was measured 26 % SLOWER: dropped.
static void wg_choose_splits(const IgemmParams& p, int* splits, int* tiles_per_split) {
  const int T = p.M / 4, chunks = (T + GTC - 1) / GTC;
  const int tb = (p.Ntot / 64) * (wg_ktot(p) / 64);
  int max_by_work = chunks / 16;
  if (max_by_work < 1) max_by_work = 1;
  int best = 1;
  double best_util = 0.0;
  const int r0 = (p.compute == DN_COMPUTE_F32X3 && T >= (1 << 18)) ? 3 : 1;
  // Blocks of one round the split aims at: every CU -- unless the layer is small (less than 64 chunks per CU when spread over all 256:
  // the 4- and 8-image shards of the metric's batch).  A weight gradient block owns a CU (8 waves, 96-128 KB of LDS), so 256 of them
  // lock the input gradients on the main stream out -- the step's critical path, which the weight gradients on their side streams are
  // not; aiming at 128 leaves half the chip to it (round 5, one box: 4 images 3.713 -> 3.671 ms, 8 images 5.419 -> 5.337; 64: +-0,
  // 32: +10 %; profiles/r05_exp7_wgrad_footprint.txt).  DN_WINO_WG_TARGET overrides.
  const int cus = knobs().wino_wg_target > 0 ? knobs().wino_wg_target : ((long long)chunks * tb < 64 * 256 ? 128 : 256);
  for (int R = r0; R <= 4; ++R) {
    int sp = (R * cus) / tb;
    if (sp < 1) continue;
    if (sp > max_by_work) sp = max_by_work;
    const double util = (double)tb * sp / ((double)((tb * sp + cus - 1) / cus) * cus);
    if (util > best_util + 1e-9) {
      best_util = util;
      best = sp;
    }
    if (util >= 0.92) break;
  }
  const int cps = (chunks + best - 1) / best;
  *tiles_per_split = cps * GTC;
  *splits = (chunks + cps - 1) / cps;
}

constexpr int kWgFold = 16;   // splits summed per thread by the first reduction stage

size_t wino_wgrad_workspace_bytes(const IgemmParams& p) {
  int splits, tps;
  wg_choose_splits(p, &splits, &tps);
  const size_t slab16 = (size_t)16 * p.Ntot * wg_ktot(p);
  const size_t stage2 = splits > kWgFold ? (size_t)((splits + kWgFold - 1) / kWgFold) * slab16 : 0;
  return ((size_t)splits * slab16 + stage2) * sizeof(float);
}

// first stage of the split sum when there are many splits (narrow layers: few (co, ci) blocks, hundreds of tile splits):
// out[g][pos][idx] = sum of splits [16g, 16g+16)  -- fixed order, deterministic
__global__ void wino_wgrad_fold_kernel(const float* __restrict__ ws, float* __restrict__ out, long long slab16, int splits) {
  const int groups = (splits + kWgFold - 1) / kWgFold;
  const long long total = (long long)groups * slab16;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx / slab16);
    const long long e = idx - (long long)g * slab16;
    const int z1 = min(splits, (g + 1) * kWgFold);
    float s = 0.f;
    for (int z = g * kWgFold; z < z1; ++z) s += ws[(long long)z * slab16 + e];
    out[idx] = s;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 wg_buffer_load2s(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {      // soffset: wave-uniform
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voffset, soffset, 0);
  return __builtin_bit_cast(f32x2, v);
}

__device__ __forceinline__ f32x2 wg_buffer_load2(__amdgpu_buffer_rsrc_t r, int voffset) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voffset, 0, 0);
  return __builtin_bit_cast(f32x2, v);
}

template <bool HA, int DBG>
__global__ void __launch_bounds__(512, 1) wino_wgrad_kernel(const IgemmParams p) {
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ktot = p.D1, Cout = p.Ntot;
  const int CB = Cout / 64, KB = Ktot / 64;
  // XCD-aware order: the (co, ci) blocks of one tile split are consecutive logical blocks on one XCD (they share its G / V tiles)
  const int total = CB * KB * p.splits;
  const int per = (total + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= total) return;
  const int cib = q % KB, cob = (q / KB) % CB, split = q / (KB * CB);
  const int T = p.T;
  const int t_begin = split * p.m_per_split;
  const int t_end = min(T, t_begin + p.m_per_split);
  const int nchunks = (t_end - t_begin + GTC - 1) / GTC;

  // the input operand this ci block lies in (operands are multiples of 64 channels wide)
  int s_op = 0;
#pragma unroll
  for (int i = 1; i < DN_MAX_OPERANDS; ++i)
    if (i < p.n_in && cib * 64 >= p.in[i].ch_off) s_op = i;
  const KOperand& S = p.in[s_op];
  const int c_in_op = cib * 64 - S.ch_off;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);

  // ---- staging role (the same for every thread): one input patch AND one gradient tile, two channels wide, of every other
  //      chunk: waves 0-3 own the even chunks, waves 4-7 the odd ones
  const int group = wave >> 2;
  const int item_t = (tid & 255) >> 5;                  // tile within the chunk
  const int item_c = (tid & 31) * 2;                    // channel pair within the 64-wide block
  const int stB = item_t * G_ROWB + item_c * 4;         // staging store offset inside a position plane
  const int pos0 = 4 * (wave >> 1) + 2 * (wave & 1);    // this wave's two positions
  const int frB = pos0 * G_PLANE + (lane >> 5) * G_ROWB + (lane & 31) * 4;   // fragment read offset (+ pp*G_PLANE + ks*2*G_ROWB + h*128)

  f32x16 acc[2][2][2];       // [pp][h (co half)][nn (ci half)]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][c][e] = 0.f;

  // ---- staging state: two patches in flight (vA / vB), one gradient tile (gv)
  f32x2 vA[16], vB[16], gv[8];
  unsigned pmA = 0, pmB = 0;
  f32x2 sc2, sh2;
  const int shB = (int)S.sh * 4, swB = (int)S.sw * 4;
  const int gpixB = Cout * 4, growB = p.OW * Cout * 4;
  if constexpr (HA) {
    const bool op_aff = S.scale != nullptr;
    const f32x2 l1 = *reinterpret_cast<const f32x2*>((op_aff ? S.scale : S.p) + (op_aff ? c_in_op + item_c : 0));
    const f32x2 l2 = *reinterpret_cast<const f32x2*>((op_aff ? S.shift : S.p) + (op_aff ? c_in_op + item_c : 0));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      sc2[e] = op_aff ? l1[e] : 1.f;
      sh2[e] = op_aff ? l2[e] : 0.f;
    }
  }
  const float relu_floor = (HA && S.scale != nullptr) ? 0.f : -__builtin_huge_valf();

  // tile of chunk `target` -> patch offset + validity mask (V) / tile offset (G)
  auto decode_tile = [&](int target, unsigned* tx, unsigned* ty, int* n) -> bool {
    const int t = t_begin + target * GTC + item_t;
    const bool live = t < t_end;
    const unsigned r = fastdiv_dev(live ? (unsigned)t : 0u, (unsigned)p.TW, p.mTW, tx);
    *n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, ty);
    return live;
  };
  auto load_patch = [&](f32x2 (&v)[16], unsigned& pm, int target) {
    unsigned tx, ty;
    int n;
    const bool live = decode_tile(target, &tx, &ty, &n);
    const int py = 2 * (int)ty - 1, px = 2 * (int)tx - 1;
    const int off0 = (n * (int)S.sn + py * (int)S.sh + px * (int)S.sw + c_in_op + item_c) * 4;
    // rows -1 / +2 and columns -1 / +2 of the patch are the only ones that can fall outside
    unsigned cols = 0x6u | (px >= 0 ? 1u : 0u) | (px + 3 < p.IW ? 8u : 0u);
    cols = live ? cols : 0u;
    pm = (py >= 0 ? cols : 0u) | (cols << 4) | (cols << 8) | (py + 3 < p.IH ? cols << 12 : 0u);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int off = off0 + (i >> 2) * shB + (i & 3) * swB;
      asm volatile("" : "+v"(off));
      off = ((pm >> i) & 1u) ? off : -1;
      v[i] = wg_buffer_load2(rsrcX, off);
    }
  };
  auto load_grad = [&](int target) {
    unsigned tx, ty;
    int n;
    const bool live = decode_tile(target, &tx, &ty, &n);
    const int off0 = (((n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx) * Cout + cob * 64 + item_c) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int off = off0 + (i >> 1) * growB + (i & 1) * gpixB;
      asm volatile("" : "+v"(off));
      off = live ? off : -1;
      gv[i] = wg_buffer_load2(rsrcG, off);
    }
  };
  auto affine_piece = [&](f32x2 (&v)[16], unsigned pm, int i) {
    if constexpr (HA) {
      // clamp to [floor, cap]: floor = 0 is the ReLU, cap = 0 re-zeroes a halo pixel the BatchNorm shift lifted (one v_med3 each)
      const float cap = ((pm >> i) & 1u) ? __builtin_huge_valf() : 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) v[i][e] = __builtin_amdgcn_fmed3f(fmaf(v[i][e], sc2[e], sh2[e]), relu_floor, cap);
    }
  };
  auto row_piece = [&](f32x2 (&v)[16], int b) {                      // B^T d, in place
    const f32x2 d0 = v[0 + b], d1 = v[4 + b], d2 = v[8 + b];
    v[0 + b] = d0 - d2;
    v[4 + b] = d1 + d2;
    v[8 + b] = d2 - d1;
    v[12 + b] = d1 - v[12 + b];
  };
  auto col_piece = [&](f32x2 (&v)[16], int b2, int i) {              // (B^T d) B and the LDS stores of transform row i
    char* dst = smemB + b2 * G_BUFB + G_OPB + stB + (4 * i) * G_PLANE;
    *reinterpret_cast<f32x2*>(dst + 0 * G_PLANE) = v[4 * i + 0] - v[4 * i + 2];
    *reinterpret_cast<f32x2*>(dst + 1 * G_PLANE) = v[4 * i + 1] + v[4 * i + 2];
    *reinterpret_cast<f32x2*>(dst + 2 * G_PLANE) = v[4 * i + 2] - v[4 * i + 1];
    *reinterpret_cast<f32x2*>(dst + 3 * G_PLANE) = v[4 * i + 1] - v[4 * i + 3];
  };
  // gradient tile y[a][b] = gv[2a+b] -> u (rows): u0 = y0., u1 = y0. + y1., u2 = y0. - y1., u3 = y1.  (sign of A's last row dropped);
  // g[i][.] = (u_i0, u_i0 + u_i1, u_i0 - u_i1, u_i1)
  auto g_cols = [&](int b2, int i) {
    char* dst = smemB + b2 * G_BUFB + stB + (4 * i) * G_PLANE;
    f32x2 u0, u1;
    if (i == 0) { u0 = gv[0]; u1 = gv[1]; }
    else if (i == 1) { u0 = gv[0] + gv[2]; u1 = gv[1] + gv[3]; }
    else if (i == 2) { u0 = gv[0] - gv[2]; u1 = gv[1] - gv[3]; }
    else { u0 = gv[2]; u1 = gv[3]; }
    *reinterpret_cast<f32x2*>(dst + 0 * G_PLANE) = u0;
    *reinterpret_cast<f32x2*>(dst + 1 * G_PLANE) = u0 + u1;
    *reinterpret_cast<f32x2*>(dst + 2 * G_PLANE) = u0 - u1;
    *reinterpret_cast<f32x2*>(dst + 3 * G_PLANE) = u1;
  };
  auto stage_all = [&](f32x2 (&v)[16], unsigned pm, int b2) {        // affine + transform + store of a loaded patch (prologue)
#pragma unroll
    for (int i = 0; i < 16; ++i) affine_piece(v, pm, i);
#pragma unroll
    for (int b = 0; b < 4; ++b) row_piece(v, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_piece(v, b2, i);
  };

  // ---- pipeline: the patch of target chunk T is requested while chunk T-4 is multiplied, clamped + row-transformed during T-2,
  //      column-transformed + stored during T-1; its gradient tile is requested during T-2 and stored during T-1.
  //      Phase of a wave at chunk c: ph = (c - group) & 3:  0: request patch c+4 -> vA, gradient c+2; finish rows of vB (target c+2)
  //                                                          1: store vB and the gradient (target c+1)
  //                                                          2: request patch c+4 -> vB, gradient c+2; finish rows of vA (target c+2)
  //                                                          3: store vA and the gradient (target c+1)
  // prologue = the phases of chunks -4 .. -1
  if (group == 0) {
    load_patch(vA, pmA, 0);
    load_grad(0);
    stage_all(vA, pmA, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) g_cols(0, i);
    load_patch(vB, pmB, 2);           // phase 2 of chunk -2
  } else {
    load_patch(vA, pmA, 1);           // phase 0 of chunk -3
    load_patch(vB, pmB, 3);           // phase 2 of chunk -1 (requests only; its rows are finished during chunk 1)
    load_grad(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) affine_piece(vA, pmA, i);
#pragma unroll
    for (int b = 0; b < 4; ++b) row_piece(vA, b);
  }
  __syncthreads();

  // ---- main loop: 32 slots per chunk, one MFMA each; fragments (scalar LDS reads) one (k-step, position) group ahead
  float fa[2][2], fb[2][2];
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const int ph = (c - group) & 3;
    const char* Gb = smemB + buf * G_BUFB + frB;
    const char* Vb = Gb + G_OPB;
    fa[0][0] = *reinterpret_cast<const float*>(Gb);
    fa[0][1] = *reinterpret_cast<const float*>(Gb + 128);
    fb[0][0] = *reinterpret_cast<const float*>(Vb);
    fb[0][1] = *reinterpret_cast<const float*>(Vb + 128);
    auto body = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_tag)::value;
      constexpr bool REQ = (PH & 1) == 0;                  // request + finish-rows phase, else store phase
      f32x2 (&vreq)[16] = PH == 0 ? vA : vB;               // PH 0: request into vA;  PH 2: into vB
      unsigned& pmreq = PH == 0 ? pmA : pmB;
      f32x2 (&vfin)[16] = (PH == 0 || PH == 1) ? vB : vA;  // PH 0: finish vB, PH 1: store vB;  PH 2 / 3: vA
      const unsigned pmfin = (PH == 0 || PH == 1) ? pmB : pmA;
      if constexpr (REQ && !(DBG & 32)) {
        load_patch(vreq, pmreq, c + 4);
        load_grad(c + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      wg_static_for<32>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int g = m / 4, qq = m % 4;           // g = (k-step, position): 4 MFMAs
        constexpr int pp = g & 1;
        constexpr int h = qq >> 1, nn = qq & 1;
        constexpr int cur = g & 1, nxt = cur ^ 1;
        acc[pp][h][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][h], fb[cur][nn], acc[pp][h][nn], 0, 0, 0);
        // ---- side work of this slot
        if constexpr (g < 7 && !(DBG & 16)) {
          constexpr int g1 = g + 1;
          constexpr int o = (g1 & 1) * G_PLANE + (g1 >> 1) * 2 * G_ROWB;
          if constexpr (qq == 0) fa[nxt][0] = *reinterpret_cast<const float*>(Gb + o);
          if constexpr (qq == 1) fa[nxt][1] = *reinterpret_cast<const float*>(Gb + o + 128);
          if constexpr (qq == 2) fb[nxt][0] = *reinterpret_cast<const float*>(Vb + o);
          if constexpr (qq == 3) fb[nxt][1] = *reinterpret_cast<const float*>(Vb + o + 128);
        }
        if constexpr (REQ && !(DBG & 2)) {
          if constexpr (m >= 8 && m < 24) affine_piece(vfin, pmfin, m - 8);
          if constexpr (m >= 24 && m < 28) row_piece(vfin, m - 24);
        }
        if constexpr (!REQ && !(DBG & 4)) {
          if constexpr (m < 8 && (m % 2) == 0) col_piece(vfin, buf ^ 1, m / 2);
          if constexpr (m >= 8 && m < 16 && (m % 2) == 0) g_cols(buf ^ 1, (m - 8) / 2);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    switch (ph) {
      case 0: body(std::integral_constant<int, 0>{}); break;
      case 1: body(std::integral_constant<int, 1>{}); break;
      case 2: body(std::integral_constant<int, 2>{}); break;
      default: body(std::integral_constant<int, 3>{}); break;
    }
    if (!(DBG & 64)) __syncthreads();
  }

  // ---- partial dU of this split: ws[split][pos][co][ci]
  float* ws = p.ws + ((size_t)split * 16 + pos0) * (size_t)Cout * Ktot;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = cob * 64 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = cib * 64 + 32 * nn + (lane & 31);
          ws[(size_t)pp * Cout * Ktot + (size_t)row * Ktot + col] = acc[pp][h][nn][r];
        }
}

// ------------------------------------------------------------------------------------------------ DN_COMPUTE_F32X3 variant
// The same block (64 co x 64 ci x 16 positions, 8 waves, 8 tiles per chunk, the same LDS planes, the same epilogue), with every
// fp32 product formed on the bf16 matrix cores from three exact bf16 pieces per operand (dn_winograd.hip, PREC = 3):
//   * one v_mfma_f32_32x32x16_bf16 carries TWO partial products of the chunk's 8 tiles.  Its 16 k-slots are 8 per half-wave, and a
//     half-wave's slots only ever meet the same half-wave's slots of the other operand -- so each half-wave simply uses ITS OWN four
//     tiles (g, 2+g, 4+g, 6+g; g = lane >> 5: the tiles the fp32 kernel reads there) in slots 0-3 and again in slots 4-7, with a
//     different piece:  (x0, x1) . (y0, y0) = x0y0 + x1y0,   (x0, x1) . (y1, y1) = x0y1 + x1y1,   (x0, x2) . (y2, y0) = x0y2 + x2y0,
//     summed over both half-waves = all 8 tiles.  No cross-lane traffic; 3 matrix instructions (96 cycles) per (position, 32 co,
//     32 ci) and chunk instead of 4 x 64 cycles of the fp32 instruction;
//   * a lane splits its four values per fragment (x = x0 + x1 + x2 exactly: round, subtract, round, subtract): 18 vector
//     instructions, issued under the matrix instructions of the previous group, LDS reads two groups ahead; operands are rebuilt in
//     place one 3-instruction group after their last use;
//   * one patch in flight per thread (the chunk is ~2.5x shorter, the vector ALUs are what the kernel is bound by, and the registers
//     are needed for the operands): a wave requests its next chunk's patch + gradient tile at the top of one chunk and transforms +
//     stores them during the next one; ONE barrier per chunk, placed after the last read of the current buffer / the last store
//     into the next one (slot 12 of 24), so that the next chunk's first operands are built under the current chunk's tail.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));

template <bool HA, int DBG = 0>     // DBG (timing ablations, wrong results): 1 no transform + stores, 2 no split arithmetic, 4 no global loads, 8 no LDS fragment reads (the split of the never-changing registers is hoisted too), 16 no LDS stores (arithmetic kept), 32 no barrier
__global__ void __launch_bounds__(512, 1) wino_wgrad_x3_kernel(const IgemmParams p) {
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ktot = p.D1, Cout = p.Ntot;
  const int CB = Cout / 64, KB = Ktot / 64;
  const int total = CB * KB * p.splits;
  const int per = (total + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= total) return;
  const int cib = q % KB, cob = (q / KB) % CB, split = q / (KB * CB);
  const int T = p.T;
  const int t_begin = split * p.m_per_split;
  const int t_end = min(T, t_begin + p.m_per_split);
  const int nchunks = (t_end - t_begin + GTC - 1) / GTC;

  int s_op = 0;
#pragma unroll
  for (int i = 1; i < DN_MAX_OPERANDS; ++i)
    if (i < p.n_in && cib * 64 >= p.in[i].ch_off) s_op = i;
  const KOperand& S = p.in[s_op];
  const int c_in_op = cib * 64 - S.ch_off;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);

  const int group = wave >> 2;                          // waves 0-3 stage the even chunks, waves 4-7 the odd ones
  const int item_t = (tid & 255) >> 5;
  const int item_c = (tid & 31) * 2;
  const int stB = item_t * G_ROWB + item_c * 4;
  const int pos0 = 4 * (wave >> 1) + 2 * (wave & 1);
  const int frB = pos0 * G_PLANE + (lane >> 5) * G_ROWB + (lane & 31) * 4;

  f32x16 acc[2][2][2];       // [pp][h (co half)][nn (ci half)]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][c][e] = 0.f;

  f32x2 v[16], gv[4];
  unsigned pm = 0;
  f32x2 sc2, sh2;
  const int shB = (int)S.sh * 4, swB = (int)S.sw * 4;
  const int gpixB = Cout * 4, growB = p.OW * Cout * 4;
  if constexpr (HA) {
    const bool op_aff = S.scale != nullptr;
    const f32x2 l1 = *reinterpret_cast<const f32x2*>((op_aff ? S.scale : S.p) + (op_aff ? c_in_op + item_c : 0));
    const f32x2 l2 = *reinterpret_cast<const f32x2*>((op_aff ? S.shift : S.p) + (op_aff ? c_in_op + item_c : 0));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      sc2[e] = op_aff ? l1[e] : 1.f;
      sh2[e] = op_aff ? l2[e] : 0.f;
    }
  }
  const float relu_floor = (HA && S.scale != nullptr) ? 0.f : -__builtin_huge_valf();

  auto decode_tile = [&](int target, unsigned* tx, unsigned* ty, int* n) -> bool {
    const int t = t_begin + target * GTC + item_t;
    const bool live = t < t_end;
    const unsigned r = fastdiv_dev(live ? (unsigned)t : 0u, (unsigned)p.TW, p.mTW, tx);
    *n = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, ty);
    return live;
  };
  // Patch addressing: four row offsets AT PATCH COLUMN 1 (always inside the image when the row is) with the row's validity folded in
  // (sign bit = past num_records: the load returns 0); columns 2 and 3 ride in the scalar offset, which the range check ignores -- so
  // the vector offset itself must be a valid one (column 0 of a patch at px = -1 would be negative: it gets its own offsets).  At most
  // two vector instructions per load; selecting every offset from a 16-bit mask cost five.
  auto load_patch = [&](int target) {
    unsigned tx, ty;
    int n;
    const bool live = decode_tile(target, &tx, &ty, &n);
    const int py = 2 * (int)ty - 1, px = 2 * (int)tx - 1;
    const int off1 = (n * (int)S.sn + py * (int)S.sh + (px + 1) * (int)S.sw + c_in_op + item_c) * 4;
    const bool r0 = live && py >= 0, r3 = live && py + 3 < p.IH;
    const bool c0 = px >= 0, c3 = px + 3 < p.IW;
    pm = (live ? 1u : 0u) | (r0 ? 2u : 0u) | (r3 ? 4u : 0u) | (c0 ? 8u : 0u) | (c3 ? 16u : 0u);
    constexpr int kOut = (int)0x80000000;
    const int m3 = c3 ? 0 : kOut;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rv = r == 0 ? r0 : r == 3 ? r3 : live;
      const int rb = off1 + r * shB;
      const int o1 = rv ? rb : kOut;
      v[4 * r + 0] = wg_buffer_load2s(rsrcX, (rv && c0) ? rb - swB : kOut, 0);
      v[4 * r + 1] = wg_buffer_load2s(rsrcX, o1, 0);
      v[4 * r + 2] = wg_buffer_load2s(rsrcX, o1, swB);
      v[4 * r + 3] = wg_buffer_load2s(rsrcX, o1 | m3, 2 * swB);
    }
  };
  auto load_grad = [&](int target) {
    unsigned tx, ty;
    int n;
    const bool live = decode_tile(target, &tx, &ty, &n);
    int off0 = (((n * p.OH + 2 * (int)ty) * p.OW + 2 * (int)tx) * Cout + cob * 64 + item_c) * 4;
    off0 = live ? off0 : (int)0x80000000;
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[i] = wg_buffer_load2s(rsrcG, off0, (i >> 1) * growB + (i & 1) * gpixB);
  };
  auto affine_piece = [&](int i) {
    if constexpr (HA) {
      unsigned pmv = pm;
      asm volatile("" : "+v"(pmv));
      constexpr unsigned kInf = 0x7f800000u;
      const int r = i >> 2, c = i & 3;
      // cap = +inf inside the image, 0 on the zero halo: row bit AND column bit, sign-extended from the 5-bit mask
      int ok = __builtin_amdgcn_sbfe(pmv, r == 0 ? 1 : r == 3 ? 2 : 0, 1);
      if (c == 0) ok &= __builtin_amdgcn_sbfe(pmv, 3, 1);
      if (c == 3) ok &= __builtin_amdgcn_sbfe(pmv, 4, 1);
      const float cap = __builtin_bit_cast(float, (unsigned)ok & kInf);
      const f32x2 t = __builtin_elementwise_fma(v[i], sc2, sh2);      // (one v_pk_fma_f32; fused like fmaf)
#pragma unroll
      for (int e = 0; e < 2; ++e) v[i][e] = __builtin_amdgcn_fmed3f(t[e], relu_floor, cap);
    }
  };
  auto row_piece = [&](int b) {
    const f32x2 d0 = v[0 + b], d1 = v[4 + b], d2 = v[8 + b];
    v[0 + b] = d0 - d2;
    v[4 + b] = d1 + d2;
    v[8 + b] = d2 - d1;
    v[12 + b] = d1 - v[12 + b];
  };
  auto put2 = [&](char* dst, f32x2 val) {
    if constexpr (DBG & 16) asm volatile("" ::"v"(val));      // (ablation: the arithmetic without the LDS store)
    else *reinterpret_cast<f32x2*>(dst) = val;
  };
  auto col_piece = [&](int b2, int i) {
    char* dst = smemB + b2 * G_BUFB + G_OPB + stB + (4 * i) * G_PLANE;
    put2(dst + 0 * G_PLANE, v[4 * i + 0] - v[4 * i + 2]);
    put2(dst + 1 * G_PLANE, v[4 * i + 1] + v[4 * i + 2]);
    put2(dst + 2 * G_PLANE, v[4 * i + 2] - v[4 * i + 1]);
    put2(dst + 3 * G_PLANE, v[4 * i + 1] - v[4 * i + 3]);
  };
  auto g_cols = [&](int b2, int i) {
    char* dst = smemB + b2 * G_BUFB + stB + (4 * i) * G_PLANE;
    f32x2 u0, u1;
    if (i == 0) { u0 = gv[0]; u1 = gv[1]; }
    else if (i == 1) { u0 = gv[0] + gv[2]; u1 = gv[1] + gv[3]; }
    else if (i == 2) { u0 = gv[0] - gv[2]; u1 = gv[1] - gv[3]; }
    else { u0 = gv[2]; u1 = gv[3]; }
    put2(dst + 0 * G_PLANE, u0);
    put2(dst + 1 * G_PLANE, u0 + u1);
    put2(dst + 2 * G_PLANE, u0 - u1);
    put2(dst + 3 * G_PLANE, u1);
  };

  // ---- operands of the matrix instructions: per co half h  A1 = (x0, x1), A3 = (x0, x2);  per ci half nn  B1 = (y0, y0),
  //      B2 = (y1, y1), B3 = (y2, y0): two dwords (four own tiles) of the first piece, two of the second
  wg_u32x4 Aop[2][2], Bop[2][3];
  float raw[2][4];
  unsigned Pk[3][2];                                    // pieces of the fragment being built: [piece][tile pair]
  // build k of a chunk (3 matrix instructions each): what it produces and where it reads
  //   0: B of (pp 0, nn 1)   1: A of (pp 0, h 1)   2: A of (pp 1, h 0)   3: B of (pp 1, nn 0)   4: B of (pp 1, nn 1)   5: A of (pp 1, h 1)
  //   6: A of (pp 0, h 0) of the NEXT chunk       7: B of (pp 0, nn 0) of the NEXT chunk
  auto issue_reads = [&](int k, const char* cur, const char* nxt) {
    const bool isB = (k == 0 || k == 3 || k == 4 || k == 7);
    const int pp = (k >= 2 && k <= 5) ? 1 : 0, half = (k == 0 || k == 1 || k == 4 || k == 5) ? 1 : 0;
    const char* base = (k >= 6 ? nxt : cur) + (isB ? G_OPB : 0) + pp * G_PLANE + half * 128;
    if constexpr (DBG & 8) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) raw[k & 1][ks] = *reinterpret_cast<const float*>(base + ks * 2 * G_ROWB);
  };
  auto split_half = [&](int k, int hf, unsigned (&P)[3][2]) {          // tiles (2 hf, 2 hf + 1) of the fragment: x = P0 + P1 + P2 exactly
    const f32x2 x = f32x2{raw[k & 1][2 * hf], raw[k & 1][2 * hf + 1]};
    if constexpr (DBG & 2) {
      P[0][hf] = __builtin_bit_cast(unsigned, x[0]);
      P[1][hf] = __builtin_bit_cast(unsigned, x[1]);
      P[2][hf] = __builtin_bit_cast(unsigned, x[0]) ^ 0x11u;
      return;
    }
    const wg_bf16x2 h = __builtin_convertvector(x, wg_bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const wg_bf16x2 m = __builtin_convertvector(r1, wg_bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const wg_bf16x2 l = __builtin_convertvector(r2, wg_bf16x2);
    P[0][hf] = __builtin_bit_cast(unsigned, h);
    P[1][hf] = __builtin_bit_cast(unsigned, m);
    P[2][hf] = __builtin_bit_cast(unsigned, l);
  };
  auto finish_build = [&](int k, int part) {                             // part 0 .. 2: spread over the three slots of the group
    const bool isB = (k == 0 || k == 3 || k == 4 || k == 7);
    const int half = (k == 0 || k == 1 || k == 4 || k == 5) ? 1 : 0;
    if (part == 0) {
      split_half(k, 0, Pk);
    } else if (part == 1) {
      split_half(k, 1, Pk);
    } else {
      if (isB) {
        Bop[half][0] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[0][0], Pk[0][1]};      // (y0, y0)
        Bop[half][1] = wg_u32x4{Pk[1][0], Pk[1][1], Pk[1][0], Pk[1][1]};      // (y1, y1)
        Bop[half][2] = wg_u32x4{Pk[2][0], Pk[2][1], Pk[0][0], Pk[0][1]};      // (y2, y0)
      } else {
        Aop[half][0] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[1][0], Pk[1][1]};      // (x0, x1)
        Aop[half][1] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[2][0], Pk[2][1]};      // (x0, x2)
      }
    }
  };

  // ---- prologue: group 0 stages chunk 0 completely; group 1 requests chunk 1 (it transforms + stores it during chunk 0)
  if (group == 0) {
    load_patch(0);
    load_grad(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) affine_piece(i);
#pragma unroll
    for (int b = 0; b < 4; ++b) row_piece(b);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_piece(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) g_cols(0, i);
  } else {
    load_patch(1);
    load_grad(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) affine_piece(i);      // (what the request phase of "chunk -1" would have done)
#pragma unroll
    for (int b = 0; b < 4; ++b) row_piece(b);
  }
  __syncthreads();
  {
    const char* b0 = smemB + frB;                      // the first two operands of chunk 0 (builds 6, 7 of "chunk -1") and the reads of build 0
    issue_reads(6, b0, b0);
    finish_build(6, 0); finish_build(6, 1); finish_build(6, 2);
    issue_reads(7, b0, b0);
    finish_build(7, 0); finish_build(7, 1); finish_build(7, 2);
    issue_reads(0, b0, b0);
    issue_reads(1, b0, b0);
  }

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const char* cur = smemB + buf * G_BUFB + frB;
    const char* nxt = smemB + (buf ^ 1) * G_BUFB + frB;
    auto body = [&](auto req_tag) __attribute__((always_inline)) {
      constexpr bool REQ = decltype(req_tag)::value;       // request phase (loads for chunk c + 2), else transform + store phase (chunk c + 1)
      if constexpr (REQ && !(DBG & 4)) {
        load_patch(c + 2);
        load_grad(c + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      wg_static_for<24>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int pp = m / 12, q12 = m % 12, h = q12 / 6, nn = (q12 / 3) % 2, t = q12 % 3;
        constexpr int grp = m / 3, part = m % 3;
        const wg_u32x4 ao = Aop[h][t == 2 ? 1 : 0], bo = Bop[nn][t];
        acc[pp][h][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, ao), __builtin_bit_cast(wg_bf16x8, bo), acc[pp][h][nn], 0, 0, 0);
        // ---- side work of this slot: finish build `grp`; its last part frees its raw values: request the reads of build grp + 2
        //      (builds 0-5 read the current buffer and are all requested by slot 11; 6, 7 and the next chunk's 0, 1 read the next
        //      buffer and are requested after the barrier of slot 14)
        finish_build(grp, part);
        if constexpr (part == 2 && grp < 4) issue_reads(grp + 2, cur, nxt);
        if constexpr (part == 2 && grp == 5) issue_reads(7, cur, nxt);
        if constexpr (part == 2 && grp >= 6) issue_reads(grp - 6, nxt, nxt);
        // staging of a wave's next chunk, balanced over its two phases: the request phase clamps + row-transforms the patch it asked
        // for at the top of this chunk in its last nine slots (after the barrier, when the loads have had ~13 slots), the store phase
        // column-transforms + stores it (and the gradient tile) in the slots before the barrier
        if constexpr (REQ && !(DBG & 1)) {
          if constexpr (m >= 13 && m < 21) {
            affine_piece(2 * (m - 13));
            affine_piece(2 * (m - 13) + 1);
          }
          if constexpr (m == 21 || m == 22) {
            row_piece(2 * (m - 21));
            row_piece(2 * (m - 21) + 1);
          }
        }
        if constexpr (!REQ && !(DBG & 1)) {
          if constexpr (m < 4) g_cols(buf ^ 1, m);
          if constexpr (m >= 4 && m < 12 && (m % 2) == 0) col_piece(buf ^ 1, (m - 4) / 2);
        }
        if constexpr (m == 12 && !(DBG & 32)) __syncthreads();                   // all reads of `cur` are out (build 5's at slot 11), all stores into `nxt` are done (slot 10)
        if constexpr (m == 14) issue_reads(6, cur, nxt);          // (its raw set is build 4's until that build's last part)
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    if (((c - group) & 1) == 0) body(std::true_type{});
    else body(std::false_type{});
  }

  float* ws = p.ws + ((size_t)split * 16 + pos0) * (size_t)Cout * Ktot;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = cob * 64 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = cib * 64 + 32 * nn + (lane & 31);
          ws[(size_t)pp * Cout * Ktot + (size_t)row * Ktot + col] = acc[pp][h][nn][r];
        }
}

// ------------------------------------------------------------------------------------------------ 128 co x 64 ci x 8 positions (round 4)
// wino_wgrad_x3_kernel splits (64 + 64) x 16 positions x 8 tiles values per chunk and spends twelve vector instructions per matrix
// instruction doing it (profiles/r04_sq_counters_wgrad.txt: vector issue port 0.59 busy, matrix pipe 0.40).  A 128 x 64 block over all
// 16 positions would need 256 accumulator registers per lane; this variant takes 128 output channels x 64 input channels for HALF of
// the positions -- transform rows i = 2 ih, 2 ih + 1, block pairs (ih = 0, 1) adjacent in the grid so that the second finds its inputs
// in L2.  Per block and chunk: (128 + 64) x 8 x 8 values to split instead of (64 + 64) x 16 x 8 (-25 %), the input transform only for
// the rows it needs (three of the four patch rows are fetched: the same number of loads per matrix instruction as before), half the
// LDS bytes.  A wave owns ONE position: 4 x 2 tiles of 32 x 32 = 128 accumulators, 24 matrix instructions per chunk as before, but 6
// operand builds (four A, two B) instead of 8.  Same workspace layout and reduce kernel; needs C_out % 128 == 0.
//   chunk schedule of a wave (24 slots, matrix instruction m: h = m / 6, nn = (m / 3) % 2, piece pairing t = m % 3):
//     build 0 = B(nn 1) slots 0-2 | 1 = A(h 1) 3-5 | 2 = A(h 2) 6-8 | 3 = A(h 3) 12-14 | 4 = A(h 0) of the NEXT chunk 18-20 | 5 = B(nn 0) of the
//     next chunk 21-23; the fragment reads of the current buffer are all out by slot 5, those of the next buffer follow the barrier (slot 12)
constexpr int W_GROWB = 128 * 4;                  // one tile row of G: 128 output channels
constexpr int W_VROWB = 64 * 4;
constexpr int W_GPLANE = GTC * W_GROWB;           // one local position of G: 4 KiB
constexpr int W_VPLANE = GTC * W_VROWB;           // 2 KiB
constexpr int W_GOPB = 8 * W_GPLANE;              // 32 KiB
constexpr int W_VOPB = 8 * W_VPLANE;              // 16 KiB
constexpr int W_BUFB = W_GOPB + W_VOPB;
constexpr size_t kWgwLds = (size_t)2 * W_BUFB;    // 96 KiB

template <bool HA, int DBG = 0>     // DBG (timing ablations, wrong results; DN_WINO_WG_DBG = 2000 + bits): 1 no fragment reads (the split runs on registers the compiler
                                    // cannot see through), 2 no split arithmetic, 4 no staging (loads, clamp, transforms, stores), 8 no barrier
__global__ void __launch_bounds__(512, 1) wino_wgrad_x3w_kernel(const IgemmParams p) {
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ktot = p.D1, Cout = p.Ntot;
  const int CB = Cout / 128, KB = Ktot / 64;
  const int total = CB * KB * 2 * p.splits;
  const int per = (total + 7) >> 3;
  const int q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= total) return;
  const int ih = q & 1, q2 = q >> 1;
  const int cib = q2 % KB, cob = (q2 / KB) % CB, split = q2 / (KB * CB);
  const int T = p.T;
  const int t_begin = split * p.m_per_split;
  const int t_end = min(T, t_begin + p.m_per_split);
  const int nchunks = (t_end - t_begin + GTC - 1) / GTC;

  int s_op = 0;
#pragma unroll
  for (int i = 1; i < DN_MAX_OPERANDS; ++i)
    if (i < p.n_in && cib * 64 >= p.in[i].ch_off) s_op = i;
  const KOperand& S = p.in[s_op];
  const int c_in_op = cib * 64 - S.ch_off;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, 0x80000000u, 0x00020000);

  const int group = wave >> 2;                          // waves 0-3 stage the even chunks, waves 4-7 the odd ones
  const int item_t = (tid & 255) >> 5;
  const int item_c = (tid & 31) * 2;
  const int stV = W_GOPB + item_t * W_VROWB + item_c * 4;
  const int stG = item_t * W_GROWB + item_c * 4;          // (+ 256 for the thread's second gradient item: channels 64 + item_c)
  const int frA = wave * W_GPLANE + (lane >> 5) * W_GROWB + (lane & 31) * 4;
  const int frB = W_GOPB + wave * W_VPLANE + (lane >> 5) * W_VROWB + (lane & 31) * 4;

  f32x16 acc[4][2];          // [h (32 output channels)][nn (32 input channels)]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  f32x2 v[3][4], gv[2][4];
  unsigned pm = 0;
  f32x2 sc2, sh2;
  const int shB = (int)S.sh * 4, swB = (int)S.sw * 4;
  const int gpixB = Cout * 4, growB = p.OW * Cout * 4;
  if constexpr (HA) {
    const bool op_aff = S.scale != nullptr;
    const f32x2 l1 = *reinterpret_cast<const f32x2*>((op_aff ? S.scale : S.p) + (op_aff ? c_in_op + item_c : 0));
    const f32x2 l2 = *reinterpret_cast<const f32x2*>((op_aff ? S.shift : S.p) + (op_aff ? c_in_op + item_c : 0));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      sc2[e] = op_aff ? l1[e] : 1.f;
      sh2[e] = op_aff ? l2[e] : 0.f;
    }
  }
  const float relu_floor = (HA && S.scale != nullptr) ? 0.f : -__builtin_huge_valf();

  // The scalars of the address arithmetic, pinned in SGPRs: left to itself the compiler re-reads some of them from the kernel-argument
  // segment inside the loop (58 of 102 SGPRs in use), and the s_waitcnt lgkmcnt(0) behind such a load also drains the LDS reads in flight.
  int kTW = p.TW, kTH = p.TH, kIH = p.IH, kIW = p.IW, kOH = p.OH, kOW = p.OW, kSn = (int)S.sn, kSh = (int)S.sh, kSw = (int)S.sw;
  unsigned kmTW = p.mTW, kmTH = p.mTH;
  asm volatile("" : "+s"(kTW), "+s"(kTH), "+s"(kIH), "+s"(kIW), "+s"(kOH), "+s"(kOW), "+s"(kSn), "+s"(kSh), "+s"(kSw), "+s"(kmTW), "+s"(kmTH));
  // one tile per thread and chunk: its input patch rows ih .. ih + 2 (2 channels) and its output-gradient tile (2 x 2 channels)
  auto load_chunk = [&](int target) {
    const int t = t_begin + target * GTC + item_t;
    const bool live = t < t_end;
    unsigned tx, ty;
    const unsigned r = fastdiv_dev(live ? (unsigned)t : 0u, (unsigned)kTW, kmTW, &tx);
    const int n = (int)fastdiv_dev(r, (unsigned)kTH, kmTH, &ty);
    const int py = 2 * (int)ty - 1 + ih, px = 2 * (int)tx - 1;        // py: image row of the first FETCHED patch row
    const int off1 = (n * kSn + py * kSh + (px + 1) * kSw + c_in_op + item_c) * 4;
    const bool rv0 = live && py >= 0, rv2 = live && py + 2 < kIH;         // (the middle row is inside the image whenever the tile exists)
    const bool c0 = px >= 0, c3 = px + 3 < kIW;
    pm = (live ? 1u : 0u) | (rv0 ? 2u : 0u) | (rv2 ? 4u : 0u) | (c0 ? 8u : 0u) | (c3 ? 16u : 0u);
    constexpr int kOut = (int)0x80000000;
    const int m3 = c3 ? 0 : kOut;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool rv = k == 0 ? rv0 : k == 2 ? rv2 : live;
      const int rb = off1 + k * shB;
      const int o1 = rv ? rb : kOut;
      v[k][0] = wg_buffer_load2s(rsrcX, (rv && c0) ? rb - swB : kOut, 0);
      v[k][1] = wg_buffer_load2s(rsrcX, o1, 0);
      v[k][2] = wg_buffer_load2s(rsrcX, o1, swB);
      v[k][3] = wg_buffer_load2s(rsrcX, o1 | m3, 2 * swB);
    }
    int goff = (((n * kOH + 2 * (int)ty) * kOW + 2 * (int)tx) * Cout + cob * 128 + item_c) * 4;
    goff = live ? goff : kOut;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) gv[e][i] = wg_buffer_load2s(rsrcG, goff, (i >> 1) * growB + (i & 1) * gpixB + e * 256);
  };
  auto affine_piece = [&](int k, int c) {
    if constexpr (HA) {
      unsigned pmv = pm;
      asm volatile("" : "+v"(pmv));
      int ok = __builtin_amdgcn_sbfe(pmv, k == 0 ? 1 : k == 2 ? 2 : 0, 1);
      if (c == 0) ok &= __builtin_amdgcn_sbfe(pmv, 3, 1);
      if (c == 3) ok &= __builtin_amdgcn_sbfe(pmv, 4, 1);
      const float cap = __builtin_bit_cast(float, (unsigned)ok & 0x7f800000u);
      const f32x2 t = __builtin_elementwise_fma(v[k][c], sc2, sh2);
#pragma unroll
      for (int e = 0; e < 2; ++e) v[k][c][e] = __builtin_amdgcn_fmed3f(t[e], relu_floor, cap);
    }
  };
  // rows 2 ih, 2 ih + 1 of B^T d (fetched rows e0 e1 e2 = d[ih .. ih + 2]):  ih 0: d0 - d2, d1 + d2;  ih 1: d2 - d1, d1 - d3
  auto row_piece = [&](int c) {
    const f32x2 e0 = v[0][c], e1 = v[1][c], e2 = v[2][c];
    if (ih == 0) {                                        // (block-uniform: a scalar branch)
      v[0][c] = e0 - e2;
      v[1][c] = e1 + e2;
    } else {
      v[0][c] = e1 - e0;
      v[1][c] = e0 - e2;
    }
  };
  auto col_piece = [&](int b2, int il) {               // local positions 4 il .. 4 il + 3 of V
    char* dst = smemB + b2 * W_BUFB + stV + (4 * il) * W_VPLANE;
    *reinterpret_cast<f32x2*>(dst + 0 * W_VPLANE) = v[il][0] - v[il][2];
    *reinterpret_cast<f32x2*>(dst + 1 * W_VPLANE) = v[il][1] + v[il][2];
    *reinterpret_cast<f32x2*>(dst + 2 * W_VPLANE) = v[il][2] - v[il][1];
    *reinterpret_cast<f32x2*>(dst + 3 * W_VPLANE) = v[il][1] - v[il][3];
  };
  auto g_cols = [&](int b2, int e, int il) {           // gradient item e, local positions 4 il .. 4 il + 3 of G (row i = 2 ih + il of A dY A^T)
    char* dst = smemB + b2 * W_BUFB + stG + e * 256 + (4 * il) * W_GPLANE;
    f32x2 u0, u1;
    if (ih == 0) {
      if (il == 0) { u0 = gv[e][0]; u1 = gv[e][1]; }
      else { u0 = gv[e][0] + gv[e][2]; u1 = gv[e][1] + gv[e][3]; }
    } else {
      if (il == 0) { u0 = gv[e][0] - gv[e][2]; u1 = gv[e][1] - gv[e][3]; }
      else { u0 = gv[e][2]; u1 = gv[e][3]; }
    }
    *reinterpret_cast<f32x2*>(dst + 0 * W_GPLANE) = u0;
    *reinterpret_cast<f32x2*>(dst + 1 * W_GPLANE) = u0 + u1;
    *reinterpret_cast<f32x2*>(dst + 2 * W_GPLANE) = u0 - u1;
    *reinterpret_cast<f32x2*>(dst + 3 * W_GPLANE) = u1;
  };

  // ---- operands: A1 = (x0, x1), A3 = (x0, x2) of 32 output channels; B1 = (y0, y0), B2 = (y1, y1), B3 = (y2, y0) of 32 input channels
  wg_u32x4 Aop[2][2], Bop[2][3];
  float raw[4][4];
  unsigned Pk[3][2];
  // builds of a chunk: id 0 = B(nn 1), 1 = A(h 1), 2 = A(h 2), 3 = A(h 3) read the current buffer; 4 = A(h 0), 5 = B(nn 0) of the next chunk read
  // the next buffer; raw register set = id & 3.
  // Every fragment read is issued in the second half of a chunk (after the barrier), away from the store-phase waves' 12 x 1 KB stores
  // of the first half (measured: -1 %; the reads cost what they cost wherever they are issued, ablation 2001)
  auto issue_reads = [&](int id, const char* cur, const char* nxt) {
    if constexpr (DBG & 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(raw[id & 3][ks]));
      return;
    }
    const bool isB = id == 0 || id == 5;
    const char* buf = id >= 4 ? nxt : cur;
    if (isB) {
      const char* base = buf + frB + (id == 0 ? 128 : 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) raw[id & 3][ks] = *reinterpret_cast<const float*>(base + ks * 2 * W_VROWB);
    } else {
      const int h = id == 4 ? 0 : id;
      const char* base = buf + frA + h * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) raw[id & 3][ks] = *reinterpret_cast<const float*>(base + ks * 2 * W_GROWB);
    }
  };
  // The split of a fragment's four values (two tile pairs) runs LEVEL by level over the three slots of its build, both pairs side by side:
  // a wave alone on its SIMD issues a dependent vector instruction every 8 cycles and an independent one every 4
  // (profiles/r02_valu_rates_ubench.txt), and one pair after the other is a single chain of ten.  Residuals stay in the raw registers.
  auto split_level = [&](int id, int level) {
    float* x = raw[id & 3];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if constexpr (DBG & 2) {
        Pk[level][hf] = __builtin_bit_cast(unsigned, x[2 * hf]) ^ (unsigned)level;
        continue;
      }
      const f32x2 v2 = f32x2{x[2 * hf], x[2 * hf + 1]};
      const wg_bf16x2 pc = __builtin_convertvector(v2, wg_bf16x2);
      Pk[level][hf] = __builtin_bit_cast(unsigned, pc);
      if (level < 2) {
        const f32x2 r = v2 - __builtin_convertvector(pc, f32x2);
        x[2 * hf] = r[0];
        x[2 * hf + 1] = r[1];
      }
    }
  };
  auto finish_build = [&](int id, int part) {
    split_level(id, part);
    if (part < 2) return;
    if (id == 0 || id == 5) {
      const int nn = id == 0 ? 1 : 0;
      Bop[nn][0] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[0][0], Pk[0][1]};      // (y0, y0)
      Bop[nn][1] = wg_u32x4{Pk[1][0], Pk[1][1], Pk[1][0], Pk[1][1]};      // (y1, y1)
      Bop[nn][2] = wg_u32x4{Pk[2][0], Pk[2][1], Pk[0][0], Pk[0][1]};      // (y2, y0)
    } else {
      const int a = id == 4 ? 0 : (id & 1);                                // A(h) lives in Aop[h & 1]
      Aop[a][0] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[1][0], Pk[1][1]};       // (x0, x1)
      Aop[a][1] = wg_u32x4{Pk[0][0], Pk[0][1], Pk[2][0], Pk[2][1]};       // (x0, x2)
    }
  };

  // ---- prologue: group 0 stages chunk 0 completely; group 1 requests chunk 1 (it column-transforms + stores it during chunk 0)
  load_chunk(group);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) affine_piece(k, c);
#pragma unroll
  for (int c = 0; c < 4; ++c) row_piece(c);
  if (group == 0) {
#pragma unroll
    for (int il = 0; il < 2; ++il) {
      col_piece(0, il);
      g_cols(0, 0, il);
      g_cols(0, 1, il);
    }
  }
  __syncthreads();
  {
    const char* b0 = smemB;
    issue_reads(4, b0, b0);
    finish_build(4, 0); finish_build(4, 1); finish_build(4, 2);
    issue_reads(5, b0, b0);
    finish_build(5, 0); finish_build(5, 1); finish_build(5, 2);
    issue_reads(0, b0, b0);
    issue_reads(1, b0, b0);
    issue_reads(2, b0, b0);
    issue_reads(3, b0, b0);
  }

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const char* cur = smemB + buf * W_BUFB;
    const char* nxt = smemB + (buf ^ 1) * W_BUFB;
    auto body = [&](auto req_tag) __attribute__((always_inline)) {
      constexpr bool REQ = decltype(req_tag)::value;       // request phase (loads for chunk c + 2), else transform + store phase (chunk c + 1)
      if constexpr (REQ && !(DBG & 4)) load_chunk(c + 2);
      __builtin_amdgcn_sched_barrier(0);
      wg_static_for<24>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int h = m / 6, nn = (m / 3) % 2, t = m % 3;
        const wg_u32x4 ao = Aop[h & 1][t == 2 ? 1 : 0], bo = Bop[nn][t];
        acc[h][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, ao), __builtin_bit_cast(wg_bf16x8, bo), acc[h][nn], 0, 0, 0);
        // ---- side work: operand builds (three slots each) and the fragment reads that follow them
        constexpr int bid = m < 9 ? m / 3 : (m >= 12 && m < 15) ? 3 : (m >= 18 && m < 21) ? 4 : m >= 21 ? 5 : -1;
        constexpr int part = m % 3;
        if constexpr (bid >= 0) finish_build(bid, part);
        if constexpr (m == 13) issue_reads(4, cur, nxt);
        if constexpr (m == 15) issue_reads(5, cur, nxt);
        if constexpr (m == 17) issue_reads(2, nxt, nxt);          // (the next chunk's builds 2, 3, 0, 1: its current buffer is this one's next)
        if constexpr (m == 19) issue_reads(3, nxt, nxt);
        if constexpr (m == 20) issue_reads(0, nxt, nxt);
        if constexpr (m == 23) issue_reads(1, nxt, nxt);
        // ---- staging of the wave's next chunk
        // ---- staging of the wave's next chunk: the request phase clamps + row-transforms what it asked for at the top of this chunk in
        //      its second half, the store phase column-transforms + stores before the barrier.  (Measured: everything in the store phase
        //      +3 %; the clamp in the build-free slots 9-11 / 15-17 +-0.)
        if constexpr (DBG & 4) {
        } else if constexpr (REQ) {
          if constexpr (m >= 13 && m < 19) {
            affine_piece((2 * (m - 13)) / 4, (2 * (m - 13)) % 4);
            affine_piece((2 * (m - 13) + 1) / 4, (2 * (m - 13) + 1) % 4);
          }
          if constexpr (m >= 19 && m < 23) row_piece(m - 19);
        } else {
          if constexpr (m == 0) g_cols(buf ^ 1, 0, 0);
          if constexpr (m == 2) g_cols(buf ^ 1, 1, 0);
          if constexpr (m == 4) g_cols(buf ^ 1, 0, 1);
          if constexpr (m == 6) g_cols(buf ^ 1, 1, 1);
          if constexpr (m == 8) col_piece(buf ^ 1, 0);
          if constexpr (m == 10) col_piece(buf ^ 1, 1);
        }
        if constexpr (m == 12 && !(DBG & 8)) __syncthreads();      // all reads of `cur` are long out (previous chunk), all stores into `nxt` are done (slot 10)
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    if (((c - group) & 1) == 0) body(std::true_type{});
    else body(std::false_type{});
  }

  const int pos = 4 * (2 * ih + (wave >> 2)) + (wave & 3);
  float* ws = p.ws + ((size_t)split * 16 + pos) * (size_t)Cout * Ktot;
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cob * 128 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = cib * 64 + 32 * nn + (lane & 31);
        ws[(size_t)row * Ktot + col] = acc[h][nn][r];
      }
}

// dU = sum over splits (with the sign of A's last row restored), dW = G^T dU G, written in the framework layout [co][ci][3][3].
// Four threads per 4 consecutive ci, one per transform row i: each sums its 4 positions over the splits (the 16 positions x splits
// reads are independent float4 streams; with one thread per quad the big layers ran 256 blocks of 64 dependent-free but serial loads
// at 1.1 TB/s), applies G along j, and the rows meet through LDS for G along i.  Fixed summation order: deterministic.
__global__ void __launch_bounds__(256) wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Ktot, int splits, int dw_vec,
                                                                int cin_total) {
  // 16 quads per block; thread = (quad ql, transform row i, split slice zs): the splits z = zs, zs + 4, ... of the row's 4 positions
  // are summed by the thread, the four slices meet through LDS in slice order (fixed order: deterministic).  (One thread per (quad,
  // row) walking all splits ran the 64-channel layers as 16 blocks of 192 serial loads: 54 us at the very end of the backward pass.)
  __shared__ f32x4 zl[16][4][4][4];      // [quad][row i][j][slice]
  __shared__ f32x4 wl[16][4][3];
  const long long slab = (long long)Cout * Ktot, quads = slab >> 2;
  const int zs = threadIdx.x & 3, i = (threadIdx.x >> 2) & 3, ql = threadIdx.x >> 4;
  for (long long q0 = blockIdx.x * 16ll; q0 < quads; q0 += (long long)gridDim.x * 16) {
    const long long qd = q0 + ql;
    const bool live = qd < quads;
    const long long idx = (live ? qd : 0) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
      const int pos = 4 * i + j;
      for (int z = zs; z < splits; z += 4) sacc += *reinterpret_cast<const f32x4*>(ws + ((long long)z * 16 + pos) * slab + idx);
      zl[ql][i][j][zs] = sacc;
    }
    __syncthreads();
    if (zs == 0) {
      f32x4 u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 sacc = ((zl[ql][i][j][0] + zl[ql][i][j][1]) + zl[ql][i][j][2]) + zl[ql][i][j][3];
        u[j] = ((i == 3) != (j == 3)) ? -sacc : sacc;
      }
      // along j:  w[c] = sum_j u[j] G[j][c],  G = (1 0 0 / .5 .5 .5 / .5 -.5 .5 / 0 0 1)
      wl[ql][i][0] = u[0] + 0.5f * (u[1] + u[2]);
      wl[ql][i][1] = 0.5f * (u[1] - u[2]);
      wl[ql][i][2] = 0.5f * (u[1] + u[2]) + u[3];
    }
    __syncthreads();
    if (live && zs == 0) {
      // along i:  o[r][c] = sum_i G[i][r] w_i[c]; the 4 x 9 results are 36 consecutive floats of dw (idx * 9 is a multiple of 4 floats);
      // thread i stores the float4s i, i + 4 (and 8 for i == 0)
      float o[4][9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x4 w0 = wl[ql][0][c], w1 = wl[ql][1][c], w2 = wl[ql][2][c], w3 = wl[ql][3][c];
        const f32x4 o0 = w0 + 0.5f * (w1 + w2), o1 = 0.5f * (w1 - w2), o2 = 0.5f * (w1 + w2) + w3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e][0 * 3 + c] = o0[e];
          o[e][1 * 3 + c] = o1[e];
          o[e][2 * 3 + c] = o2[e];
        }
      }
      // (cin_total > Ktot: these are the leading channels of a wider layer -- dn_conv2d_wgrad's split for a trailing 1-channel piece --
      //  and a row of dw is cin_total * 9 floats; the quad's 4 channels never straddle a row, Ktot being a multiple of 64)
      const long long co = idx / Ktot, ci = idx - co * Ktot;
      float* dst = dw + (co * cin_total + ci) * 9;
#pragma unroll
      for (int v = 0; v < 9; ++v) {
        if ((v & 3) != i) continue;
        f32x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = o[(v * 4 + e) / 9][(v * 4 + e) % 9];
        if (dw_vec) *reinterpret_cast<f32x4*>(dst + 4 * v) = w4;
        else {                               // a gradient slice of the optimizer arena need not be 16-byte aligned
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * v + e] = w4[e];
        }
      }
    }
    __syncthreads();
  }
}

int launch_wino_wgrad(IgemmParams& p, float* dw, hipStream_t stream) {
  int splits, tps;
  wg_choose_splits(p, &splits, &tps);
  p.splits = splits;
  p.m_per_split = tps;
  p.T = p.M / 4;
  p.TH = p.GH / 2;
  p.TW = p.GW / 2;
  p.OH = p.GH;
  p.OW = p.GW;
  p.mTW = fastdiv_magic((unsigned)p.TW);
  p.mTH = fastdiv_magic((unsigned)p.TH);
  const int dbg = knobs().wino_wg_dbg;
  const bool x3 = p.compute == DN_COMPUTE_F32X3 && (dbg == 0 || dbg >= 1000);
  auto kernel = x3 ? (p.any_affine ? wino_wgrad_x3_kernel<true> : wino_wgrad_x3_kernel<false>)
                   : (p.any_affine ? wino_wgrad_kernel<true, 0> : wino_wgrad_kernel<false, 0>);
  if (x3) {
    switch (dbg) {            // DN_WINO_WG_DBG = 1000 + bits: timing ablations of the three-piece variant
      case 1001: kernel = wino_wgrad_x3_kernel<true, 1>; break;
      case 1002: kernel = wino_wgrad_x3_kernel<true, 2>; break;
      case 1004: kernel = wino_wgrad_x3_kernel<true, 4>; break;
      case 1005: kernel = wino_wgrad_x3_kernel<true, 5>; break;
      case 1008: kernel = wino_wgrad_x3_kernel<true, 8>; break;
      case 1015: kernel = wino_wgrad_x3_kernel<true, 15>; break;
      case 1016: kernel = wino_wgrad_x3_kernel<true, 16>; break;
      case 1032: kernel = wino_wgrad_x3_kernel<true, 32>; break;
      case 1024: kernel = wino_wgrad_x3_kernel<true, 24>; break;
      default: break;
    }
  }
  switch (dbg) {
    case 2: kernel = wino_wgrad_kernel<true, 2>; break;
    case 6: kernel = wino_wgrad_kernel<true, 6>; break;
    case 22: kernel = wino_wgrad_kernel<true, 22>; break;
    case 54: kernel = wino_wgrad_kernel<true, 54>; break;
    case 118: kernel = wino_wgrad_kernel<true, 118>; break;
    default: break;
  }
  // 128 x 64 x 8-position blocks when the output channels allow it (DN_WINO_WGW=0 keeps the 64 x 64 x 16 kernel)
  bool wide = x3 && dbg == 0 && (p.Ntot % 128) == 0 && knobs().wino_wgw != 0;
  if (wide) kernel = p.any_affine ? wino_wgrad_x3w_kernel<true> : wino_wgrad_x3w_kernel<false>;
  if (x3 && dbg >= 2000 && (p.Ntot % 128) == 0) {            // ablations of the wide block (tools/wgw_ablate.sh)
    wide = true;
    switch (dbg - 2000) {
      case 1: kernel = wino_wgrad_x3w_kernel<true, 1>; break;
      case 2: kernel = wino_wgrad_x3w_kernel<true, 2>; break;
      case 3: kernel = wino_wgrad_x3w_kernel<true, 3>; break;
      case 4: kernel = wino_wgrad_x3w_kernel<true, 4>; break;
      case 5: kernel = wino_wgrad_x3w_kernel<true, 5>; break;
      case 6: kernel = wino_wgrad_x3w_kernel<true, 6>; break;
      case 7: kernel = wino_wgrad_x3w_kernel<true, 7>; break;
      case 8: kernel = wino_wgrad_x3w_kernel<true, 8>; break;
      case 15: kernel = wino_wgrad_x3w_kernel<true, 15>; break;
      default: kernel = wino_wgrad_x3w_kernel<true>; break;
    }
  }
  const size_t lds = wide ? kWgwLds : kWgLds;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino_wgrad_kernel, %zu): %s", lds, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  const int Ktot = wg_ktot(p);
  const int total = (p.Ntot / 64) * (Ktot / 64) * splits;          // (the wide kernel: Ntot / 128 x 2 position halves -- the same count)
  DN_LAUNCH(kernel, dim3((total + 7) / 8 * 8), dim3(512), lds, stream, p);
  if (wide) set_last_kernel("dn::wino_wgrad_x3w_kernel<%s>", p.any_affine ? "true" : "false");
  else if (x3) set_last_kernel("dn::wino_wgrad_x3_kernel<%s, 0>", p.any_affine ? "true" : "false");
  else set_last_kernel("dn::wino_wgrad_kernel<%s, %d>", p.any_affine ? "true" : "false", (dbg == 2 || dbg == 6 || dbg == 22 || dbg == 54 || dbg == 118) ? dbg : 0);
  int rc = check_launch("wino_wgrad_kernel");
  if (rc != DN_OK) return rc;
  const long long slab = (long long)p.Ntot * Ktot;
  const float* src = p.ws;
  int nsrc = splits;
  if (splits > kWgFold) {
    float* folded = p.ws + (size_t)splits * 16 * slab;
    const int groups = (splits + kWgFold - 1) / kWgFold;
    const long long total = (long long)groups * 16 * slab;
    int fb = (int)((total + 255) / 256);
    if (fb > 16384) fb = 16384;
    DN_LAUNCH(wino_wgrad_fold_kernel, dim3(fb), dim3(256), 0, stream, p.ws, folded, 16 * slab, splits);
    src = folded;
    nsrc = groups;
  }
  int blocks = (int)(((slab >> 2) + 15) / 16);       // 16 quads (x 4 transform rows x 4 split slices) per block
  if (blocks > 16384) blocks = 16384;
  const int cin_total = p.dw_cin_total > Ktot ? p.dw_cin_total : Ktot;
  DN_LAUNCH(wino_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, src, dw, p.Ntot, Ktot, nsrc,
                     ((reinterpret_cast<uintptr_t>(dw) & 15) == 0 && cin_total == Ktot) ? 1 : 0, cin_total);
  return check_launch("wino_wgrad_reduce_kernel");
}

}  // namespace dn
