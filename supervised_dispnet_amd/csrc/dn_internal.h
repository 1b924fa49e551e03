// Internal declarations shared by the HIP translation units of libdispnet_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <functional>
#include <tuple>
#include <type_traits>

#include "dispnet_hip.h"

namespace dn {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* fmt, ...);   // name (as rocprofv3 prints it) of the main kernel the last conv-family call launched
int check_launch(const char* what);

static inline hipStream_t as_stream(dn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- launch tape (dn_tape.hip; dn_tape_* in dispnet_hip.h).  Every kernel launch of this library goes through DN_LAUNCH.  While a
// tape is being recorded the launch is ALSO kept -- kernel, grid, LDS bytes, stream and the by-value kernel arguments -- so that the
// whole launch sequence of a training step can be re-issued later by one C call (dn_tape_replay) without any of the host work that
// produced it (Python, descriptor marshalling, planning): at 4 images per GPU the step is ~210 dependent launches of ~20 us of host
// time each.  Replays are only valid while every pointer baked into the recorded arguments is (the caller records under a private
// memory pool and replays into the same buffers).
struct LaunchTape;
extern std::atomic<LaunchTape*> g_tape_rec;          // the tape being recorded, or nullptr
// A recorded launch is re-issued as op(stop): stop == nullptr is the plain launch; with an event, the launch carries it as its STOP event
// (hipExtLaunchKernel: the event is the dispatch packet's own completion signal, no marker packet behind the kernel -- what a fence that
// follows the launch on its stream is turned into, dn_tape.hip).
void tape_push(LaunchTape* t, std::function<void(hipEvent_t)>&& op, const void* kernel, hipStream_t stream);

template <typename... KArgs, typename... Args>
static inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
  kernel<<<grid, block, lds, stream>>>(args...);
  if (LaunchTape* t = g_tape_rec.load(std::memory_order_relaxed)) {
    std::tuple<std::decay_t<KArgs>...> held(args...);            // the kernel's own parameter types: hipExtLaunchKernel takes their addresses
    tape_push(t, [=](hipEvent_t stop) mutable {
      if (stop == nullptr) {
        std::apply([&](auto&... a) { kernel<<<grid, block, lds, stream>>>(a...); }, held);
      } else {
        std::apply([&](auto&... a) {
          void* argv[] = {static_cast<void*>(&a)...};
          if (hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, lds, stream, nullptr, stop, 0) != hipSuccess) {
            // never leave the event unrecorded (a wait on it would not wait): the plain launch and an event record of its own
            (void)hipGetLastError();
            kernel<<<grid, block, lds, stream>>>(a...);
            (void)hipEventRecord(stop, stream);
          }
        }, held);
      }
    }, reinterpret_cast<const void*>(kernel), stream);
  }
}
#define DN_LAUNCH(...) ::dn::launch(__VA_ARGS__)

// Tuning / test switches (DN_* environment variables), read ONCE and again only on dn_reload_knobs(): the conv entry points
// consult a dozen of them per launch.
struct Knobs {
  bool no_winograd, no_winograd_wgrad;   // DN_NO_WINOGRAD / DN_NO_WINOGRAD_WGRAD: the tiled kernels take the 3x3 / stride 1 layers (tests compare the two)
  bool no_direct;                        // DN_NO_DIRECT: no one-channel head kernels
  bool no_thin, no_thin_conv;            // DN_NO_THIN / DN_NO_THIN_CONV: the thin full-resolution layers on the tiled kernels (tests compare the three forms)
  bool no_lds3;                          // DN_NO_LDS3: no LDS-resident family (dn_lds3*.hip, dn_stemk.hip, stem3)
  bool no_splitk, no_wino_splitk, no_x3_splitk, no_wino8_tail;   // DN_NO_SPLITK / DN_NO_WINO_SPLITK / DN_NO_X3_SPLITK / DN_NO_WINO8_TAIL: K splits of small grids off
  bool no_x3_direct, no_x3_wgrad;        // DN_NO_X3_DIRECT / DN_NO_X3_WGRAD: the fp32 matrix instruction in the direct forward family / tiled weight gradient
  bool no_tap_windows;                   // DN_NO_TAP_WINDOWS: > 32-tap weight gradients on the unscheduled kernel
  int wino_wgw;                          // DN_WINO_WGW: 0 = keep the 64 x 64 x 16-position weight-gradient block where 128 x 64 x 8 would run (bitwise test)
  int wino_dbg, wino_wg_dbg, lds3_dbg;   // DN_WINO_DBG / DN_WINO_WG_DBG / DN_LDS3_DBG: timing / ablation instantiations (tools/wino_timing.py, tools/lds3_timing.py)
  unsigned long long wino_dbgptr;        // DN_WINO_DBGPTR: where they write their time stamps
  int wino_min_tiles;                    // DN_WINO_MIN_TILES: fewest 2x2 output tiles the Winograd kernels take (192)
  int wino_splitk_target, wino_splitk_maxblocks;   // DN_WINO_SPLITK_TARGET (512 blocks) / _MAXBLOCKS (208): the 4-wave kernel's K split of small grids
  int pack_blocks;                       // DN_PACK_BLOCKS: blocks per table entry of the batched weight re-lay (x2 for the Winograd entries)
  bool no_riding_fences;                 // DN_NO_RIDING_FENCES: the launch tape's device-scope fences as event records of their own, not as the stop event of the launch in front
  int wino_nmajor;                       // DN_WINO_NMAJOR (1): Winograd forward/dgrad tile order within an XCD: 1 tile row fastest (one 64-cout weight slice per XCD), 0 cout slice fastest, 2/3 by slice count
  int wino_wg_target;                    // DN_WINO_WG_TARGET (0 = by rule: 128 for small layers, else 256): blocks per round the Winograd weight gradient's tile split aims at
  int wino8;                             // DN_WINO8 (0 never / 1 always / -1 = by rule): 8-wave three-piece Winograd kernel
  int wino8_var;                         // DN_WINO8_VAR: main-loop variant of the 8-wave kernel (A/B measurements)
};
const Knobs& knobs();

// dn_conv_desc.compute as the kernels see it: 0 (DN_COMPUTE_DEFAULT, a zeroed descriptor) is the three-piece arithmetic, unknown values
// the fp32 matrix instruction
inline int norm_compute(int c) {
  return c == DN_COMPUTE_BF16 ? DN_COMPUTE_BF16 : ((c == DN_COMPUTE_F32X3 || c == DN_COMPUTE_DEFAULT) ? DN_COMPUTE_F32X3 : DN_COMPUTE_F32);
}

#define DN_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      ::dn::set_error(__VA_ARGS__);      \
      return (code);                     \
    }                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------- igemm plan
constexpr int kMaxTaps = 64;     // 7x7 = 49 taps (single phase); 4 phases of a 4x4/stride-2 = 16
constexpr int kMaxPhases = 4;
constexpr int kChunk = 32;       // K elements staged per main-loop step

struct KOperand {                // device view of a dn_operand
  const float* p;
  const float* scale;
  const float* shift;
  long long sn, sh, sw, sc;
  int C;
  int up;
  int vec;                       // C % 4 == 0 && sc == 1 && 16-byte aligned rows  -> float4 path
  int ch_off;                    // first channel of this operand inside the concatenated K axis
  unsigned mC;                   // fastdiv magic of C
  int small;                     // every element offset fits int32 (fast path requirement)
};

struct KResult {
  float* p;
  long long sn, sh, sw;
  int C;
  int n_begin;                   // first global output column of this segment
  int accumulate;
  int linear;                    // pixel offset == m * sw (dense, un-phased)  -> no div/mod in the epilogue
};

struct KPhase {
  int ntaps;
  int tap0;                      // first entry in the tap tables
  int ooy, oox;                  // output pixel = grid * ostride + (ooy, oox)
  int nchunks;                   // K chunks of this phase (sum over operands of ceil(ntaps*C/32))
  long long w_off;               // element offset of this phase's [Npad][nchunks*32] block in w_packed
};

// what dn_bn_finalize takes per layer; also carried by a conv launch that finishes the statistics itself (dn_conv_desc.bnf_*, dn_fold.h)
struct BnFinalizeArgs {
  const float* conv_bias;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  float momentum, eps;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
};

struct IgemmParams {
  KOperand in[DN_MAX_OPERANDS];
  KResult out[DN_MAX_OPERANDS];
  KPhase ph[kMaxPhases];
  int8_t tdy[kMaxTaps], tdx[kMaxTaps], tr[kMaxTaps], ts[kMaxTaps];
  int n_in, n_out, nphases;
  int N, GH, GW, M;              // grid (per phase); M = N*GH*GW
  int sy, sx;                    // input pixel = grid*s + tap offset
  int osy, osx;                  // output pixel stride per grid step
  int IH, IW, OH, OW;
  int Ntot, Npad;                // output columns (all segments), padded to the N tile
  int BN;                        // N tile chosen (32 / 64 / 128)
  int R, S;
  int n_is_dim0;                 // framework weight layout: [n][c][R][S] (1) or [c][n][R][S] (0)
  int D0, D1;                    // framework weight dims 0 and 1
  const float* w;
  const float* bias;
  float* bn_partial;
  int act;
  float act_p0, act_p1;
  // weight-gradient use only
  const float* g;                // dy (or x for conv-transpose) [M][Ntot]
  float* ws;                     // [splits][Npad][Kp]
  int splits, m_per_split;
  unsigned mGW, mGH;             // fastdiv magics of GW, GH
  int allvec;                    // every operand takes the float4 fast path (and g, for wgrad)
  int uni32;                     // fast plan: <= 32 taps per phase, zero padding, >= 1 operand the scheduled loaders take
                                 //   (C % 32 == 0 or C in {4,8,16}, float4-addressable, no upsample, < 2 GiB span)
  int wg_uniform;                // every operand: C % 32 == 0, float4-addressable, no upsample (weight-gradient fast path)
  int any_affine;                // some operand carries a pending BN-apply + ReLU
  int reflect;                   // gather with ReflectionPad2d index mapping instead of zero fill
  int compute;                   // DN_COMPUTE_F32 / _BF16 / _F32X3 (descriptor field; honoured by the Winograd forward / input gradient)
  int tile_store;                // epilogue may stage the result tile in LDS and store whole pixels
  // Winograd F(2x2,3x3) launches only (dn_winograd.hip)
  // BatchNorm-backward column sums of the producer, taken in the input gradient's epilogue (dn_conv_desc.bnb_*; Winograd kernels only)
  const float *bnb_y, *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
  float* bnb_partial;
  int ksplit, ks_chunks, ks_cnt_floats;   // input-channel split of small grids (dn_winograd.hip): splits, chunks of the K axis, float offset of the partial tiles
  int dw_cin_total;                       // weight gradient of the LEADING operands of a wider layer: channels per row of dw (0: this plan's own)
  int ks_reg, ks_tail;                    // 8-wave kernel: tiles [0, ks_reg) run whole, the ks_tail tiles after them are split (dn_winograd8.hip)
  float* recip_out;              // one-channel heads: 1 / result, dense [N][OH][OW] (dn_conv_desc.recip_out), or nullptr
  float* ks_ws;                  // caller workspace (dn_conv_desc.splitk_ws): zeroed int counters, then the partial tiles
  size_t ks_ws_bytes;
  int T, TH, TW;                 // 2x2 output tiles: total, per image column / row
  unsigned mTW, mTH;             // fastdiv magics
  // last-arrival epilogues (round 5, dn_fold.h; Winograd kernels): requested through dn_conv_desc.bnf_* / bnb_dgamma, bnb_dbeta; the
  // launcher sets fold_bn / fold_bnb when the launch qualifies (few partial rows, counters available)
  BnFinalizeArgs bnf;
  float *bnb_dgamma, *bnb_dbeta;
  int* fold_cnt;                 // kFoldCounters zeroed, self-resetting ints: the last 256 bytes of dn_conv_desc.splitk_ws
  int fold_bn, fold_bnb;
  int nmajor;                // Winograd tile order: q -> (mb, nb) = (q % MT, q / MT) instead of (q / NT, q % NT)
};

// floor(n / d) for 0 <= n < 2^31 with a precomputed magic (see fastdiv_magic); branch-free
static inline unsigned fastdiv_magic(unsigned d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / d); }

int build_plan(const dn_conv_desc* d, bool for_wgrad, IgemmParams* p);

// one row of the batched weight re-lay table (dn_pack_entry_fill / dn_pack_many): everything a pack kernel takes as arguments
struct PackEntry {
  IgemmParams p;
  const float* w;
  float* wp;
  long long total;      // elements the kernel walks (direct: packed elements; Winograd: wino_packed_elems)
  int wino;             // 0: direct [phase][Npad][K] layout, 1: Winograd fragment-order transform (fp32), 2: the same as bf16, 3: as three bf16 pieces
  int NS;               // Winograd: Npad / 32
};
int launch_wino_pack_many(const PackEntry* tab_dev, int first, int n, hipStream_t stream);
int launch_wino_pack16_many(const PackEntry* tab_dev, int first, int n, int pieces, hipStream_t stream);
int launch_wino_pack16(const IgemmParams& p, const float* w, float* wp, int pieces, hipStream_t stream);
long long wino_packed_floats(const IgemmParams& p, int layout);
int wino_layout(const dn_conv_desc* d, const IgemmParams& p);

// floor(n/d) on the device with the plan's magic (estimate is exact or one low; branch-free fix-up); *rem = n - q*d
__device__ __forceinline__ unsigned fastdiv_dev(unsigned n, unsigned d, unsigned M, unsigned* rem) {
  unsigned q = __umulhi(n, M);
  unsigned r = n - q * d;
  const bool fix = r >= d;
  q += fix ? 1u : 0u;
  r -= fix ? d : 0u;
  *rem = r;
  return q;
}

// dn_direct.hip: matrix-core-free kernels for the one-channel disparity heads, dispatched from the conv entry points
bool head_fwd_eligible(const dn_conv_desc* d, const IgemmParams& p);
int launch_head_fwd(const IgemmParams& p, hipStream_t stream);
bool head_fwd_fuses_reciprocal(const dn_conv_desc* d, const IgemmParams& p);
bool head_dgrad_eligible(const dn_conv_desc* d, const IgemmParams& p);
int launch_head_dgrad(const IgemmParams& p, hipStream_t stream);
bool head_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p);
size_t head_wgrad_workspace_bytes(const IgemmParams& p);
int launch_head_wgrad(const IgemmParams& p, float* dw, float* workspace, hipStream_t stream);

// dn_winograd.hip: Winograd F(2x2,3x3) forward / input-gradient of the 3x3 stride-1 pad-1 layers with 16-aligned channels
bool wino_eligible(const dn_conv_desc* d, const IgemmParams& p);
long long wino_packed_elems(const IgemmParams& p);
int launch_wino_pack(const IgemmParams& p, const float* w, float* wp, hipStream_t stream);
int launch_wino_conv(IgemmParams& p, hipStream_t stream);
bool wino_folds_bn_finalize(const IgemmParams& p);   // the launch will finish the BatchNorm statistics / the BatchNorm-backward sums itself
bool wino_folds_bn_sums(const IgemmParams& p);
int wino_splitk_choice(const IgemmParams& p);
size_t conv_x3_splitk_workspace_upper_bytes(const IgemmParams& p);   // dn_conv.hip: K split of the three-piece direct kernel
size_t wino_splitk_workspace_bytes(const IgemmParams& p);
// dn_winograd_wgrad.hip: Winograd weight gradient of the same layers (operands and output channels multiples of 64)
bool wino_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p);
size_t wino_wgrad_workspace_bytes(const IgemmParams& p);
int launch_wino_wgrad(IgemmParams& p, float* dw, hipStream_t stream);

// dn_thin.hip: weight gradient of the thin full-resolution 3x3 layers (first encoder layer, iconv0) on 16x16x4 MFMAs from global memory
bool thin_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p);
size_t thin_wgrad_workspace_bytes(const IgemmParams& p);
int launch_thin_wgrad(IgemmParams& p, float* dw, hipStream_t stream);
bool thin_conv_eligible(const dn_conv_desc* d, const IgemmParams& p);
int launch_thin_conv(const IgemmParams& p, hipStream_t stream);

// dn_lds3.hip: LDS-resident three-piece direct convolutions of the thin full-resolution decoder layers (iconv0 / upconv0 forward + input gradient)
bool lds3_conv_eligible(const dn_conv_desc* d, const IgemmParams& p);
int launch_lds3_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream);
// dn_lds3_wgrad.hip: weight gradients of iconv0 (16 [+ 1] -> 16) and the first encoder layer (<= 3 -> 64) from channel-planar LDS tiles
bool lds3_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p);
size_t lds3_wgrad_workspace_bytes(const IgemmParams& p);
int launch_lds3_wgrad(const dn_conv_desc* fwd, IgemmParams& p, float* dw, hipStream_t stream);
bool lds3k_wgrad_eligible(const dn_conv_desc* fwd, const IgemmParams& p);      // 96-channel form (iconv1), eight waves
size_t lds3k_wgrad_workspace_bytes(const IgemmParams& p);
int launch_lds3k_wgrad(const dn_conv_desc* fwd, IgemmParams& p, float* dw, hipStream_t stream);
int launch_wgrad_reduce(const IgemmParams& p, float* dw, hipStream_t stream);   // dn_conv.hip: fixed-order sum of p.splits slabs of p.ws -> dw
// dn_wgrad_x3.hip: the tiled weight gradient with three-piece arithmetic on the bf16 matrix cores (64 / 128-wide n tiles, float4 operands)
bool wgrad_x3_eligible(const IgemmParams& p);
int launch_wgrad_x3(const IgemmParams& p, hipStream_t stream);
// dn_lds3k.hip: the 8-wave form (roles = phase x M tile x K quarter) for iconv1 / upconv1
bool lds3k_conv_eligible(const dn_conv_desc* d, const IgemmParams& p);
int launch_lds3k_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream);
bool stem3_conv_eligible(const dn_conv_desc* d, const IgemmParams& p);
bool stemk_conv_eligible(const dn_conv_desc* d, const IgemmParams& p);      // dn_stemk.hip: 7x7 / stride-2 first layers
int launch_stemk_conv(const dn_conv_desc* d, const IgemmParams& p, hipStream_t stream);
bool stemk_wgrad_eligible(const dn_conv_desc* d, const IgemmParams& p);
size_t stemk_wgrad_workspace_bytes(const dn_conv_desc* d, const IgemmParams& p);
int launch_stemk_wgrad(const dn_conv_desc* d, IgemmParams& p, float* dw, hipStream_t stream);
int launch_stem3_conv(const IgemmParams& p, hipStream_t stream);

}  // namespace dn
