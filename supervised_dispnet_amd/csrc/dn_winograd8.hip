// Winograd F(2x2, 3x3) forward / input gradient, three-piece arithmetic (DN_COMPUTE_F32X3), 8-wave form (gfx950 only).
//
// Same mathematics, weight layout, staging and epilogue contract as dn::wino_conv_kernel<1, HA, 0, 3> (dn_winograd.hip); what differs
// is how the 16 GEMMs of a 16-channel chunk are dealt to waves.  The 4-wave kernel gives a wave 4 positions x 32 tiles x 64 output
// channels and runs two blocks per CU: per chunk the CU pulls 2 x 96 KB of weight pieces + 2 x 32 KB of patches through its one
// vector-memory path (40 vector-memory instructions per wave and chunk), which is what bounds it (the ablation
// without the weight stream is 23 % faster, the matrix instructions alone 54 %).  Here ONE block of 8 waves owns 64 tiles x 64 output
// channels and a wave takes 2 positions x 64 tiles x 64 output channels:
//   * every weight piece is still fetched by exactly one wave of the CU, straight into registers in fragment order, but now feeds
//     64 tiles: 96 KB of weights + 64 KB of patches per chunk and CU (28 vector-memory instructions per wave and chunk) for the same
//     48 matrix instructions per wave;
//   * every transformed input value is still read from LDS (and split into its three bf16 pieces) by exactly one wave;
//   * two waves per SIMD as before (128 accumulator registers + <= 128 others).
// The price is the output transform: a row of the 4x4 transform domain is spread over two waves, so the transform along j needs one
// more exchange through LDS (lane-private float4 slots, conflict-free) before the cross-row exchange of the 4-wave kernel, and with
// one block per CU no partner block runs under the prologue / epilogue.  launch_wino_conv picks per layer (wino8_wanted).
//
// Main-loop order of a wave: unit a = 2 * (local position) + (tile half), 12 matrix instructions each (output-channel half nn, then
// the six partial products); weight fragment (position, nn, piece) is used by both tile halves, so the six fragments of a position
// live in a ring of six registers that is refilled in release order during the second tile half (10-12 instructions of lead).
#include <stdlib.h>

#include "dn_internal.h"
#include "dn_wino_common.h"

namespace dn {

namespace {
constexpr int BT8 = 64;                                   // tiles per block
constexpr size_t kXchgBytes = (size_t)8 * BT8 * WZLD * sizeof(float);        // 2 halves b x 4 transform rows x [64 tiles][72]
constexpr size_t kStatBytes = (size_t)(2 * 8 * 2 * 64) * sizeof(float);    // [2 tile groups][8 waves][2][64 couts]: statistics (+ group means) or the BatchNorm-backward sums
constexpr size_t kLds8 = kXchgBytes + kStatBytes;
static_assert(kLds8 >= WinoCfg<2>::LDS, "the epilogue's exchange area must cover the main loop's two chunk buffers");
static_assert(kLds8 <= 160 * 1024, "one CU has 160 KB of LDS");
}  // namespace

// Scalar fp32 add / subtract / fma that the SLP vectoriser cannot fuse into v_pk_*_f32 (a packed fp32 instruction beside matrix
// instructions costs far more than the two scalar ones it replaces: MI355X_MICROARCH.md, "price of one filler beside MFMAs").
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Experiment bits of DBG (round 6): 4096 scalar subtractions in the split, 8192 the two cout halves of a unit interleaved (consecutive
// matrix instructions on different accumulators), 16384 scalar staging arithmetic, 32768 two split chains side by side
template <bool HA, int DBG>
__global__ void __launch_bounds__(512, 1) wino_conv8_kernel(const IgemmParams p) {
  constexpr bool XS = (DBG & 4096) != 0, XI = (DBG & 8192) != 0, XT = (DBG & 16384) != 0, XL = (DBG & 32768) != 0;   // (XL: the four pairs of a unit's split side by side in two slots)
  constexpr bool XD = (DBG & 262144) != 0;      // patch loads one chunk deeper: issued in slots 24..39 of chunk c for chunk c + 2, transformed + stored in slots 0..13 of c + 1
  using Cfg = WinoCfg<2>;
  constexpr int BT = BT8, HALFB = Cfg::HALFB, POSB = Cfg::POSB, SUBB = Cfg::SUBB, BUFB = Cfg::BUFB;
  extern __shared__ __align__(16) float smem[];
  char* smemB = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: wave-uniform branches below)
  const int wi = wave >> 1, jh = wave & 1;              // transform row i, half of its four positions (j = 2 jh, 2 jh + 1)
  // Staging schedules: the eight waves run the same 48-slot chunk between two barriers; if they also ran the same side work in the same
  // slots, the CU's vector-memory path, vector ALUs and LDS store port would each be busy in one third of the chunk and idle in the
  // other two (measured: the non-matrix work of a chunk alone took 4850 cycles against 3070 of matrix time).  So the two waves of a SIMD
  // (w and w + 4) are half a chunk out of phase: group A loads its patch early in chunk c and transforms + stores it late in c; group B
  // transforms + stores early in c what it loaded late in c - 1 (two chunks ahead of its use).
  // XCD-aware tile order (see igemm_conv_u32_kernel): contiguous logical tile ranges per XCD, N tile fastest
  const int MT = (p.T + BT - 1) / BT, NT = p.Npad / WBN;
  // Tail split (p.ksplit > 1, launch_wino8_variant): one block per CU means a grid of R + tail blocks (R a multiple of 256) runs R / 256
  // full rounds and then a round with only `tail` CUs busy -- 832 blocks = 3.25 rounds cost 4.  The tiles of that last partial round
  // are split `ksplit` ways along the input channels instead (block ids R + kz * tail + t: dispatched after the full rounds, all
  // splits of a tile on XCD t % 8); the partial accumulators meet in the caller's workspace, the last arrival sums them in index order
  // and runs the epilogue (the same scheme as the small-grid split of the 4-wave kernel).
  int q, kz = 0, g0 = 0, g1 = 0x7fffffff;
  if (p.ksplit > 1 && (int)blockIdx.x >= p.ks_reg) {
    const int r = (int)blockIdx.x - p.ks_reg;
    kz = r / p.ks_tail;
    if (kz >= p.ksplit) return;
    q = p.ks_reg + r % p.ks_tail;
    const int cps = (p.ks_chunks + p.ksplit - 1) / p.ksplit;
    g0 = kz * cps;
    g1 = g0 + cps < p.ks_chunks ? g0 + cps : p.ks_chunks;
  } else {
    const int nreg = p.ksplit > 1 ? p.ks_reg : MT * NT;
    const int per = (nreg + 7) >> 3;
    q = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || q >= nreg) return;
  }
  const bool split_tile = g1 != 0x7fffffff;
  const int mb = p.nmajor ? q % MT : q / NT, nb = p.nmajor ? q / MT : q % NT;
  long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, te1 = 0, te2 = 0;
  // DBG & 2048 (with 4): per-wave phase sums over the chunks of the main loop -- head (barrier release -> first matrix instruction), the
  // four 12-slot quarters, the wait at the chunk's barrier (tools/wino_timing.py prints them per wave)
  unsigned ph_head = 0, ph_q[4] = {0, 0, 0, 0}, ph_bar = 0, ph_n = 0;
  long long ph_t = 0;
  if (DBG & 4) t0 = clock64();

  // ---- staging role: one 4x4 patch of 2 channels per thread and chunk (64 tiles x 8 channel pairs)
  constexpr int VW = 2;
  typedef f32x2 fV;
  const int st_tile = tid >> 3, cg = tid & 7;
  const int k0 = cg * VW;
  unsigned pmask = 0;            // bit 4a+b: patch pixel (a, b) lies inside the image (and the tile exists)
  int pn, py, px;
  {
    const int t = mb * BT + st_tile;
    unsigned tx, ty;
    const unsigned r = fastdiv_dev(t < p.T ? (unsigned)t : 0u, (unsigned)p.TW, p.mTW, &tx);
    pn = (int)fastdiv_dev(r, (unsigned)p.TH, p.mTH, &ty);
    py = 2 * (int)ty - 1;
    px = 2 * (int)tx - 1;
    unsigned colm = 0, rowm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) colm |= ((unsigned)(px + b) < (unsigned)p.IW) ? (1u << b) : 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) rowm |= ((unsigned)(py + a) < (unsigned)p.IH) ? (1u << (4 * a)) : 0u;
    pmask = t < p.T ? rowm * colm : 0u;
  }

  f32x16 acc[2][2][2];            // [local position][tile half][cout half]

  // B fragments straight from the packed weights wp16[chunk][pos][n/32][piece][lane][8 bf16] (1 KiB per fragment)
  const int NS = p.Npad / 32;
  const char* wcur16 = reinterpret_cast<const char*>(p.w) + ((size_t)(4 * wi + 2 * jh) * NS + 2 * nb) * 3072 + lane * 16;
  const size_t wchunk16B = (size_t)16 * NS * 3072, wj16B = (size_t)NS * 3072;
  wcur16 += (size_t)g0 * wchunk16B;                     // (tail split: the stream starts at this block's first chunk)
  // stream index g (relative to the current chunk): chunk g / 12, k = g % 12 = 6 posl + 3 nn + (2 - piece); register g % 6
  constexpr int WRING = 6;
  bf16x8 bq[WRING];
  auto load_b3 = [&](int g) __attribute__((always_inline)) {
    const int ch = g / 12, k = g % 12, posl = k / 6, nn = (k % 6) / 3, piece = 2 - k % 3;
    const size_t off = (size_t)ch * wchunk16B + (size_t)posl * wj16B + (size_t)nn * 3072 + (size_t)piece * 1024;
    bq[g % WRING] = *reinterpret_cast<const bf16x8*>(wcur16 + off);     // (round 6: as __builtin_nontemporal_load the kernel is 8-12 % SLOWER, profiles/r06_exp5)
  };
#pragma unroll
  for (int g = 0; g < WRING; ++g) load_b3(g);

  const int stA = (k0 >> 3) * SUBB + ((k0 >> 2) & 1) * HALFB + st_tile * 16 + (k0 & 3) * 4;   // staging store offset in a buffer (+ pos * POSB)
  // fragment read: lane (tile, g) takes k 8g .. 8g+7 = 8-k group g, both 4-k planes
  const int frA3 = (4 * wi + 2 * jh) * POSB + (lane >> 5) * SUBB + (lane & 31) * 16;

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
        v[i] = fV{x, 0.f};
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
    auto load_aff = [&]() __attribute__((always_inline)) {
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
    auto affine_piece = [&](int i) __attribute__((always_inline)) {
      if constexpr (HA) {
        // clamp to [floor, cap]: floor = 0 is the ReLU, cap = 0 re-zeroes a halo pixel the BatchNorm shift lifted
        unsigned pm = pmask;
        asm volatile("" : "+v"(pm));
        const int msk = __builtin_amdgcn_sbfe((int)pm, i, 1);
        const float cap = __builtin_bit_cast(float, msk & 0x7f800000);
        fV t;
        if constexpr (XT) { t[0] = s_fma(v[i][0], sc4[0], sh4[0]); t[1] = s_fma(v[i][1], sc4[1], sh4[1]); }
        else t = __builtin_elementwise_fma(v[i], sc4, sh4);
#pragma unroll
        for (int e = 0; e < VW; ++e) v[i][e] = __builtin_amdgcn_fmed3f(t[e], relu_floor, cap);
      }
    };
    auto row_piece = [&](int b) __attribute__((always_inline)) {        // B^T d, in place: rows (0,1,2,3) <- (d0-d2, d1+d2, d2-d1, d1-d3)
      const fV d0 = v[0 + b], d1 = v[4 + b], d2 = v[8 + b];
      if constexpr (XT) {
        const fV d3 = v[12 + b];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          v[0 + b][e] = s_sub(d0[e], d2[e]);
          v[4 + b][e] = s_add(d1[e], d2[e]);
          v[8 + b][e] = s_sub(d2[e], d1[e]);
          v[12 + b][e] = s_sub(d1[e], d3[e]);
        }
        return;
      }
      v[0 + b] = d0 - d2;
      v[4 + b] = d1 + d2;
      v[8 + b] = d2 - d1;
      v[12 + b] = d1 - v[12 + b];
    };
    auto col_piece = [&](int b2, int i, int half) __attribute__((always_inline)) {    // (B^T d) B and the LDS stores of transform row i
      char* dst = smemB + b2 * BUFB + stA + (4 * i) * POSB;
      if constexpr (XT) {
        const fV c0 = v[4 * i + 0], c1 = v[4 * i + 1], c2 = v[4 * i + 2], c3 = v[4 * i + 3];
        if (half == 0) {
          *reinterpret_cast<fV*>(dst + 0 * POSB) = fV{s_sub(c0[0], c2[0]), s_sub(c0[1], c2[1])};
          *reinterpret_cast<fV*>(dst + 1 * POSB) = fV{s_add(c1[0], c2[0]), s_add(c1[1], c2[1])};
        } else {
          *reinterpret_cast<fV*>(dst + 2 * POSB) = fV{s_sub(c2[0], c1[0]), s_sub(c2[1], c1[1])};
          *reinterpret_cast<fV*>(dst + 3 * POSB) = fV{s_sub(c1[0], c3[0]), s_sub(c1[1], c3[1])};
        }
        return;
      }
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
#pragma unroll
      for (int i = 0; i < 16; ++i) load_v(i);
      load_aff();
      if (first_op) {
        first_op = false;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[a][m][nn][e] = 0.f;
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
      if constexpr (XD) {                          // second stage of the fill: the next chunk's patch is in flight when the loop starts
        cnB = (c_lo + 1 < nch ? c_lo + 1 : c_lo) * (WKC * 4);
        const unsigned lm0 = scalar1 ? 0u : pmask;
#pragma unroll
        for (int i = 0; i < 16; ++i) load_v_t(i, std::false_type{}, lm0);
        load_aff();
      }
    }
    __syncthreads();
    if (DBG & 4) t1 = clock64();

    f32x4 raw[2];
    if constexpr (DBG & 512) raw[0] = raw[1] = f32x4{1.f, 2.f, 3.f, 4.f};
    bf16x8 fa3[2][3];
    auto read_raw = [&](const char* Ab, int a) __attribute__((always_inline)) {       // unit a = 2 posl + mm
      raw[0] = *reinterpret_cast<const f32x4*>(Ab + (a >> 1) * POSB + (a & 1) * 512);
      raw[1] = *reinterpret_cast<const f32x4*>(Ab + (a >> 1) * POSB + HALFB + (a & 1) * 512);
    };
    auto split_pair = [&](int slot, int q) __attribute__((always_inline)) {           // channels 2q, 2q+1 of the fragment: x = h + m + l exactly
      const f32x2 x = f32x2{raw[q >> 1][2 * (q & 1)], raw[q >> 1][2 * (q & 1) + 1]};
      const bf16x2 h = __builtin_convertvector(x, bf16x2);
      if constexpr (DBG & 16) {                         // ablation (timing only, wrong results): no split arithmetic
        fa3[slot][0][2 * q] = h[0]; fa3[slot][0][2 * q + 1] = h[1];
        fa3[slot][1][2 * q] = h[1]; fa3[slot][1][2 * q + 1] = h[0];
        fa3[slot][2][2 * q] = h[0]; fa3[slot][2][2 * q + 1] = h[0];
        return;
      }
      f32x2 r1, r2;
      const f32x2 hf = __builtin_convertvector(h, f32x2);
      if constexpr (XS) r1 = f32x2{s_sub(x[0], hf[0]), s_sub(x[1], hf[1])};
      else r1 = x - hf;
      const bf16x2 m = __builtin_convertvector(r1, bf16x2);
      const f32x2 mf = __builtin_convertvector(m, f32x2);
      if constexpr (XS) r2 = f32x2{s_sub(r1[0], mf[0]), s_sub(r1[1], mf[1])};
      else r2 = r1 - mf;
      const bf16x2 l = __builtin_convertvector(r2, bf16x2);
      fa3[slot][0][2 * q] = h[0]; fa3[slot][0][2 * q + 1] = h[1];
      fa3[slot][1][2 * q] = m[0]; fa3[slot][1][2 * q + 1] = m[1];
      fa3[slot][2][2 * q] = l[0]; fa3[slot][2][2 * q + 1] = l[1];
    };
    // (a 1-channel piece is a single chunk: what the loop "re-fetches" for it is never used, so its loads are masked off)
    const unsigned lmask = scalar1 ? 0u : pmask;
    for (int c = c_lo; c < nch; ++c) {
      const bool more = c + 1 < nch;
      cnB = (more ? c + 1 : c) * (WKC * 4);            // the last chunk re-fetches itself into the idle buffer: no branch
      if constexpr (XD) cnB = (c + 2 < nch ? c + 2 : nch - 1) * (WKC * 4);
      const char* Ab = smemB + buf * BUFB + frA3;
      if constexpr ((DBG & 2048) != 0) ph_t = clock64();
      if constexpr (!(DBG & 512)) read_raw(Ab, 0);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) split_pair(0, q4);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DBG & 2048) != 0) {
        const long long tn = clock64();
        ph_head += (unsigned)(tn - ph_t);
        ph_t = tn;
        __builtin_amdgcn_sched_barrier(0);
      }
      // slot m = 12 a + 6 nn + t: x0y2, x0y1, x1y1, x0y0, x1y0, x2y0
      static_for<48>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int a = m / 12, q12 = m % 12, nn = XI ? q12 % 2 : q12 / 6, t = XI ? q12 / 2 : q12 % 6, posl = a >> 1, mm = a & 1;
        constexpr int AS[6] = {0, 0, 1, 0, 1, 2}, BS[6] = {2, 1, 1, 0, 0, 0};
        if constexpr (!(DBG & 256)) {
          acc[posl][mm][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[a & 1][AS[t]], bq[3 * nn + 2 - BS[t]], acc[posl][mm][nn], 0, 0, 0);
        } else {                                          // ablation: everything but the matrix instructions (operands kept alive)
          asm volatile("" :: "v"(fa3[a & 1][AS[t]]), "v"(bq[3 * nn + 2 - BS[t]]));
        }
        // ---- side work of this slot
        if constexpr (mm == 1 && !(DBG & 32)) {           // second (last) tile half: the fragments of this position are released
          if constexpr (t == 0) load_b3(6 * posl + 3 * nn + 0 + WRING);
          if constexpr (t == 2) load_b3(6 * posl + 3 * nn + 1 + WRING);
          if constexpr (t == 5) load_b3(6 * posl + 3 * nn + 2 + WRING);
        }                                                  // (XI: the same release points, two slots apart instead of six)
        if constexpr (a < 3 && q12 == 1 && !(DBG & 512)) read_raw(Ab, a + 1);
        if constexpr (!XL && a < 3 && q12 >= 6 && q12 < 10) split_pair((a + 1) & 1, q12 - 6);
        if constexpr (XL && a < 3 && (q12 == 6 || q12 == 8)) {
          split_pair((a + 1) & 1, q12 - 6);
          split_pair((a + 1) & 1, q12 - 5);
        }
        constexpr bool STG = !(DBG & 64);                 // (DBG 64: ablation without the staging of the next chunk)
        constexpr int SCHED = XD ? 2 : ((DBG & 1024) ? 0 : 1);   // staging slot schedules (1 = shipped; others: measurements)
        // 0 (DN_WINO_DBG=1028, the round-2 schedule, kept for A/B timing): loads 2..17, clamp 22..29, rows 30..33, cols 34..41
        // 1 (shipped): loads on odd slots 1..31, clamp 34..37 (4 per slot), rows 38..39, cols 40..47: 5686 -> 5354 cycles per chunk.
        // Also measured: loads 0..15 + transform 28..47 (5607), two loads per slot 0..7 (5697), s_setprio asymmetry between the two
        // waves of a SIMD (no change)
        if constexpr (SCHED == 0) {
          if constexpr (STG && m >= 2 && m < 18) load_v_t(m - 2, std::false_type{}, lmask);
          if constexpr (STG && m == 18) load_aff();
          constexpr int A0 = 22, R0 = A0 + 8, C0 = R0 + 4;
          if constexpr (STG && m >= A0 && m < R0) {
            affine_piece(2 * (m - A0));
            affine_piece(2 * (m - A0) + 1);
          }
          if constexpr (STG && m >= R0 && m < R0 + 4) row_piece(m - R0);
          if constexpr (STG && m >= C0 && m < C0 + 8) col_piece(buf ^ 1, (m - C0) / 2, (m - C0) % 2);
        } else if constexpr (SCHED == 1) {
          // (timing ablations, wrong results: 65536 = only 4 of the 16 patch loads are issued, the others copy them -- what a raw input
          //  region fetched once per block would leave of the vector-memory instructions; 131072 = 16 extra ds_read_b64 per thread and
          //  chunk -- what reading the patches from such a region would add)
          if constexpr (STG && m >= 1 && m < 33 && (m & 1)) {
            constexpr int li = (m - 1) / 2;
            if constexpr ((DBG & 65536) != 0 && li >= 4) v[li] = v[li & 3];
            else load_v_t(li, std::false_type{}, lmask);
            if constexpr ((DBG & 131072) != 0) {
              const fV dz = *reinterpret_cast<const fV*>(smemB + buf * BUFB + stA + li * POSB);
              asm volatile("" :: "v"(dz));
            }
          }
          if constexpr (STG && m == 2) load_aff();
          if constexpr (STG && m >= 34 && m < 38) {
#pragma unroll
            for (int u = 0; u < 4; ++u) affine_piece(4 * (m - 34) + u);
          }
          if constexpr (STG && m >= 38 && m < 40) {
            row_piece(2 * (m - 38));
            row_piece(2 * (m - 38) + 1);
          }
          if constexpr (STG && m >= 40 && m < 48) col_piece(buf ^ 1, (m - 40) / 2, (m - 40) % 2);
        } else if constexpr (SCHED == 2) {
          if constexpr (STG && m < 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) affine_piece(4 * m + u);
          }
          if constexpr (STG && m >= 4 && m < 6) {
            row_piece(2 * (m - 4));
            row_piece(2 * (m - 4) + 1);
          }
          if constexpr (STG && m >= 6 && m < 14) col_piece(buf ^ 1, (m - 6) / 2, (m - 6) % 2);
          if constexpr (STG && m >= 24 && m < 40) load_v_t(m - 24, std::false_type{}, lmask);
          if constexpr (STG && m == 40) load_aff();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((DBG & 2048) != 0 && m % 12 == 11) {
          const long long tn = clock64();
          ph_q[m / 12] += (unsigned)(tn - ph_t);
          ph_t = tn;
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      wcur16 += wchunk16B;
      __syncthreads();
      if constexpr ((DBG & 2048) != 0) {
        ph_bar += (unsigned)(clock64() - ph_t);
        ++ph_n;
      }
      buf ^= 1;
    }
  }

  if (split_tile) {
    // partial accumulators -> lane-private float4 slots [tail tile][split][32][thread]; the last arrival sums the splits in index order.
    // Visibility without an agent-scope fence as in wino_conv_kernel: the splits of a tile share an XCD (its L2): write-through stores
    // waited for with vmcnt(0), an L2 atomic counter, reader loads that bypass the CU's L1 (glc).
    __shared__ int ks_last;
    const int t = q - p.ks_reg;
    int* cnt = reinterpret_cast<int*>(p.ks_ws);
    f32x4* slots = reinterpret_cast<f32x4*>(p.ks_ws + p.ks_cnt_floats);
    f32x4* mine = slots + ((size_t)(t * p.ksplit + kz) * 32) * 512 + tid;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x16 v16 = acc[a][m][nn];
            mine[(size_t)((((a * 2 + m) * 2 + nn) * 4) + e) * 512] = f32x4{v16[4 * e], v16[4 * e + 1], v16[4 * e + 2], v16[4 * e + 3]};
          }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ks_last = (atomicAdd(cnt + t, 1) == p.ksplit - 1) ? 1 : 0;
    __syncthreads();
    if (!ks_last) return;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][m][nn][e] = 0.f;
    const __amdgpu_buffer_rsrc_t rws =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(slots + (size_t)t * p.ksplit * 32 * 512), 0, 0x7fffffff, 0x00020000);
    for (int z = 0; z < p.ksplit; ++z) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              typedef int i32x4 __attribute__((ext_vector_type(4)));
              const i32x4 vi = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(((z * 32 + ((a * 2 + m) * 2 + nn) * 4 + e) * 512 + tid) * 16), 0, 1 /* glc */);
              const f32x4 v = __builtin_bit_cast(f32x4, vi);
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[a][m][nn][4 * e + u] += v[u];
            }
    }
    if (tid == 0) cnt[t] = 0;                            // (self-resetting: the workspace is reusable by the next launch on this stream)
  }
  if (DBG & 4) t2 = clock64();
  // ---- output transform along j: Z[b] = sum_j A^T[b][j] M[i][j], A^T = (1 1 1 0 / 0 1 -1 -1).  This wave holds j = 2 jh, 2 jh + 1 and
  //      completes Z[b = jh]; in place: acc[0] = its partial of Z[jh] (kept), acc[1] = its partial of Z[1 - jh] (sent to the partner):
  //      jh = 0: keeps M0 + M1, sends M1;  jh = 1: keeps -M2 - M3, sends M2
  if (jh == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) acc[0][m][nn] = acc[0][m][nn] + acc[1][m][nn];
  } else {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const f32x16 m2 = acc[0][m][nn];
        acc[0][m][nn] = -m2 - acc[1][m][nn];
        acc[1][m][nn] = m2;
      }
  }
  // exchange 1 (within a transform row) through lane-private float4 slots [wave][16][lane] (16 KB per wave, conflict-free)
  {
    f32x4* mine = reinterpret_cast<f32x4*>(smemB + wave * 16384) + lane;
    const f32x4* theirs = reinterpret_cast<const f32x4*>(smemB + (wave ^ 1) * 16384) + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x16 src = acc[1][m][nn];
          mine[((m * 2 + nn) * 4 + r4) * 64] = f32x4{src[4 * r4], src[4 * r4 + 1], src[4 * r4 + 2], src[4 * r4 + 3]};
        }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 g = theirs[((m * 2 + nn) * 4 + r4) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[0][m][nn][4 * r4 + e] += g[e];
        }
    __syncthreads();
  }
  // exchange 2 (across the rows i): plane [b = jh][i][tile][72]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        smem[((jh * 4 + wi) * BT + row) * WZLD + 32 * nn + (lane & 31)] = acc[0][m][nn][r];
      }
  __syncthreads();
  constexpr int NK = 2;                            // final role: couts 4*c4..+3 of tiles tg + 32*k
  const int c4 = tid & 15, tg = tid >> 4;
  f32x4 Y[NK][2][2];                               // [k][a][b]
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int tile = tg + 32 * k;
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(smem + ((b * 4 + 0) * BT + tile) * WZLD + 4 * c4);
      const f32x4 z1 = *reinterpret_cast<const f32x4*>(smem + ((b * 4 + 1) * BT + tile) * WZLD + 4 * c4);
      const f32x4 z2 = *reinterpret_cast<const f32x4*>(smem + ((b * 4 + 2) * BT + tile) * WZLD + 4 * c4);
      const f32x4 z3 = *reinterpret_cast<const f32x4*>(smem + ((b * 4 + 3) * BT + tile) * WZLD + 4 * c4);
      Y[k][0][b] = z0 + z1 + z2;
      Y[k][1][b] = z1 - z2 - z3;
    }

  if (DBG & 4) te1 = clock64();
  // ---- batch statistics of the pre-bias result: partial row 2*mb + k covers tiles [32(2 mb + k), +32) = 128 pixels;
  //      (sum, M2 about the group's own mean), merged by dn_bn_finalize
  const int n_first = nb * WBN + 4 * c4;
  if (p.bn_partial != nullptr) {
    float* red = smem + 8 * BT * WZLD;             // [2 groups][8 waves][64 couts], then gmean [2][64]
    float* gmean = red + 2 * 8 * 64;
    int gcount[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      int left = p.T - (mb * BT + 32 * m);
      gcount[m] = left < 0 ? 0 : (left > 32 ? 32 : left);
    }
    f32x4 s[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      s[m] = Y[m][0][0] + Y[m][0][1] + Y[m][1][0] + Y[m][1][1];       // tiles past T hold exact zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[m][e] += __shfl_xor(s[m][e], 16);
        s[m][e] += __shfl_xor(s[m][e], 32);
      }
    }
    if (lane < 16) {
#pragma unroll
      for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(red + (m * 8 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < 128) {
      const int m = tid >> 6, col = tid & 63;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[(m * 8 + w) * 64 + col];
      const int gc = m == 0 ? gcount[0] : gcount[1];
      gmean[tid] = gc > 0 ? tot / (float)(4 * gc) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(gmean + m * 64 + 4 * c4);
      f32x4 s2 = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool live = tg < gcount[m];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x4 dv = Y[m][a][b] - mu;
#pragma unroll
          for (int e = 0; e < 4; ++e) s2[e] += live ? dv[e] * dv[e] : 0.f;
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
      for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(red + (m * 8 + wave) * 64 + 4 * c4) = s[m];
    }
    __syncthreads();
    if (tid < 128) {
      const int m = tid >> 6, col = tid & 63;
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) m2 += red[(m * 8 + w) * 64 + col];
      const int n = nb * WBN + col;
      const int gc = m == 0 ? gcount[0] : gcount[1];
      if (n < p.Ntot && gc > 0) {
        float* dst = p.bn_partial + ((long long)(2 * mb + m) * p.Ntot + n) * 2;
        fold_store(dst, tot);          // (agent scope: the block that arrives last may read them, dn_fold.h)
        fold_store(dst + 1, m2);
      }
    }
  }

  if (p.bnb_y != nullptr)                // (input gradient: the BatchNorm backward's column sums of the layer below)
    wino_bn_bwd_sums<NK, 32, 8, 2>(p, Y, mb, tg, c4, wave, lane, tid, n_first, smem + 8 * BT * WZLD);

  if (DBG & 4) te2 = clock64();
  // ---- bias, activation, channel-split / accumulating stores (as in the 4-wave kernel)
  if (n_first < p.Ntot) {
    int seg = 0;
    if (p.n_out > 1 && n_first >= p.out[1].n_begin) seg = 1;
    if (p.n_out > 2 && n_first >= p.out[2].n_begin) seg = 2;
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
        const int t = mb * BT + tg + 32 * k;
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
          *reinterpret_cast<f32x4*>(o00) = v[0][0];
          *reinterpret_cast<f32x4*>(o00 + sw) = v[0][1];
          *reinterpret_cast<f32x4*>(o00 + rowB) = v[1][0];
          *reinterpret_cast<f32x4*>(o00 + rowB + sw) = v[1][1];
        }
      }
    } else {
      // element-wise: the four columns straddle results or are not float4-addressable (the 1-channel disparity piece of a concat's
      // input gradient)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int t = mb * BT + tg + 32 * k;
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
    if constexpr ((DBG & 2048) != 0) {
      if (lane == 0) {
        long long* o = reinterpret_cast<long long*>(p.ws) + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = ph_head; o[1] = ph_q[0]; o[2] = ph_q[1]; o[3] = ph_q[2]; o[4] = ph_q[3]; o[5] = ph_bar; o[6] = ph_n; o[7] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // (HW_ID: SIMD id in bits 5:4)
      }
    }
  }
  if constexpr ((DBG & 4095) == 0) {
    __syncthreads();                     // (the LDS of the epilogue is free from here)
    wino_fold_tail(p, nb, MT, smem, tid);
  }
}

template <bool HA, int DBG>
static int launch_wino8_variant(const IgemmParams& p, hipStream_t stream) {
  auto kernel = wino_conv8_kernel<HA, DBG>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds8);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino_conv8_kernel, %zu): %s", kLds8, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  const int tiles = ((p.T + BT8 - 1) / BT8) * (p.Npad / WBN);
  IgemmParams q = p;
  q.ksplit = 1;
  int blocks = (tiles + 7) / 8 * 8;
  if ((DBG & 4095) == 0 && !knobs().no_wino8_tail && p.ks_ws != nullptr) {
    // tail split (see the kernel): the last partial round of one-block-per-CU rounds
    const int cus = 256, reg = tiles / cus * cus, tail = tiles - reg;
    int chunks = 0;
    for (int i = 0; i < p.n_in; ++i) chunks += (p.in[i].C + WKC - 1) / WKC;
    int ks = tail > 0 ? cus / tail : 1;
    if (ks > chunks / 8) ks = chunks / 8;
    if (ks > 8) ks = 8;
    const size_t need = 4096 + (size_t)tail * ks * 32 * 512 * 16;
    // (measured at 32 images, tools/conv_microbench.py: 512 -> 512 @16x52 = 832 blocks = 3 rounds + 64: forward 0.428 -> 0.413 ms, input
    //  gradient 0.395 -> 0.371; 256 -> 256 @32x104 = 6 rounds + 128: +3..5 % SLOWER -- the rounds are not in lock step, so the partial
    //  round costs less than a round, and 128 tiles x 2 x 256 KB of partial accumulators cost more than it: few rounds, small tails only)
    if (reg > 0 && reg <= 3 * cus && tail > 0 && tail % 8 == 0 && tail <= 64 && ks >= 2 && p.ks_ws_bytes >= need) {
      q.ksplit = ks;
      q.ks_chunks = chunks;
      q.ks_cnt_floats = 4096 / 4;
      q.ks_reg = reg;
      q.ks_tail = tail;
      blocks = reg + tail * ks;
    }
  }
  dim3 grid(blocks);
  DN_LAUNCH(kernel, grid, dim3(512), kLds8, stream, q);
  set_last_kernel("dn::wino_conv8_kernel<%s, %d>", HA ? "true" : "false", DBG);
  return check_launch("wino_conv8_kernel");
}

// Which layers take the 8-wave form (DN_WINO8: 0 never, 1 always, unset: by the rule below).  Measured per layer at batch 32
// (profiles/r03_wino8_microbench.txt): 512 -> 512 @16x52 forward 0.466 -> 0.422 ms, @8x26 0.123 -> 0.106; 256 -> 256 -5 %; 128 -> 128
// -2 %; 64 -> 64 +-0 (four chunks: the exposed prologue / epilogue eat the main loop's gain); 768 -> 256 @8x26 (104 blocks of 64 tiles
// for 256 CUs) +30 %: the 4-wave kernel's 208 half-size blocks fill the chip better.
bool wino8_wanted(const IgemmParams& p) {
  if (p.compute != DN_COMPUTE_F32X3) return false;
  const int mode = knobs().wino8;
  if (mode == 0) return false;
  if (mode == 1) return true;
  int k = 0;
  for (int i = 0; i < p.n_in; ++i) k += (p.in[i].C + WKC - 1) / WKC * WKC;
  const long long blocks = (long long)((p.M / 4 + BT8 - 1) / BT8) * ((p.Ntot + WBN - 1) / WBN);
  // (round 5 measured a FULL split for grids below 192 blocks -- every 64-tile block split 2-8 ways along K, ks_reg = 0: 256 KB of partial
  //  accumulators per split block against the 4-wave kernel's 32 KB of transformed partial tiles; 512 -> 512 @16x52 with 4 images 0.084 ->
  //  0.082 ms, @8x26 0.044 -> 0.055, the 4-image step 3.78 -> 3.89 ms: dropped, profiles/r05_exp1.txt)
  return k >= 128 && blocks >= 192;
}

int launch_wino_conv8(const IgemmParams& p, hipStream_t stream) {
  const int dbg = knobs().wino_dbg;
  if (dbg & 4) {                        // in-kernel timestamps (tools/wino_timing.py), optionally with ablation bits (timing only)
    IgemmParams q = p;
    q.ws = reinterpret_cast<float*>(knobs().wino_dbgptr);
    switch (dbg) {
#define DN_W8_CASE(D) case D: return launch_wino8_variant<false, D>(q, stream);
      DN_W8_CASE(4 + 16) DN_W8_CASE(4 + 32) DN_W8_CASE(4 + 64) DN_W8_CASE(4 + 256) DN_W8_CASE(4 + 512) DN_W8_CASE(4 + 16 + 64) DN_W8_CASE(4 + 16 + 32 + 64)
      DN_W8_CASE(4 + 16 + 32 + 64 + 512) DN_W8_CASE(4 + 256 + 16) DN_W8_CASE(4 + 256 + 64) DN_W8_CASE(4 + 256 + 16 + 64) DN_W8_CASE(4 + 256 + 16 + 32 + 64) DN_W8_CASE(4 + 256 + 16 + 32 + 64 + 512) DN_W8_CASE(4 + 1024) DN_W8_CASE(4 + 2048) DN_W8_CASE(4 + 65536) DN_W8_CASE(4 + 131072) DN_W8_CASE(4 + 65536 + 131072) DN_W8_CASE(4 + 2048 + 262144)
#undef DN_W8_CASE
      default: break;
    }
    return q.any_affine ? launch_wino8_variant<true, 4>(q, stream) : launch_wino8_variant<false, 4>(q, stream);
  }
  switch (dbg == 0 ? knobs().wino8_var : 0) {   // round-6 experiment instantiations (results are right; DN_WINO8_VAR = 4096 | 8192 | 16384 bits)
#define DN_W8_X(D) case D: return p.any_affine ? launch_wino8_variant<true, D>(p, stream) : launch_wino8_variant<false, D>(p, stream);
    DN_W8_X(28672 + 32768) DN_W8_X(262144)      // (the single bits and other combinations measured the same: profiles/r06_exp1_wino8_variants.txt)
#undef DN_W8_X
    default: break;
  }
  return p.any_affine ? launch_wino8_variant<true, 0>(p, stream) : launch_wino8_variant<false, 0>(p, stream);
}

}  // namespace dn
