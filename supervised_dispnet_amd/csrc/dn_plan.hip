// Host-side planning for the implicit-GEMM convolution family: turns a dn_conv_desc into the tap / phase tables
// the kernels consume.  Pure host logic (also exported through dn_debug_conv_plan for CPU-side tests).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "dn_internal.h"

namespace dn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[160] = "";

void set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  return DN_OK;
}

static int pick_bn(int ntot) { return ntot <= 32 ? 32 : (ntot <= 64 ? 64 : 128); }

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

// floor division for possibly negative numerators
static int floor_div(int a, int b) {
  int q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

static void fill_operand(KOperand* k, const dn_operand* o, int ch_off) {
  k->p = o->data;
  k->scale = o->scale;
  k->shift = o->shift;
  k->sn = o->stride_n;
  k->sh = o->stride_h;
  k->sw = o->stride_w;
  k->sc = o->stride_c;
  k->C = o->C;
  k->up = o->up_shift;
  bool aligned = ((reinterpret_cast<uintptr_t>(o->data) & 15) == 0) && (o->stride_n % 4 == 0) &&
                 (o->stride_h % 4 == 0) && (o->stride_w % 4 == 0);
  k->vec = (o->C % 4 == 0 && o->stride_c == 1 && aligned) ? 1 : 0;
  k->ch_off = ch_off;
  k->mC = fastdiv_magic((unsigned)o->C);
}

// largest element offset the kernel may form for an operand stored as [N][Hs][Ws][C] with the given strides
static bool offsets_fit_int32(const KOperand& k, int N, int IH, int IW) {
  const long long hs = (IH >> k.up) + 1, ws = (IW >> k.up) + 1;
  const long long span = (long long)N * (k.sn < 0 ? -k.sn : k.sn) + hs * (k.sh < 0 ? -k.sh : k.sh) +
                         ws * (k.sw < 0 ? -k.sw : k.sw) + (long long)k.C * (k.sc < 0 ? -k.sc : k.sc);
  return span < (1ll << 31);
}

// ------------------------------------------------------------------------------------------------------ knobs
static Knobs g_knobs;
static std::atomic<int> g_knobs_state{0};      // 0 = unread, 1 = valid
static std::mutex g_knobs_mu;

static void read_knobs(Knobs* k) {
  auto on = [](const char* n) { return getenv(n) != nullptr; };
  auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
  k->no_winograd = on("DN_NO_WINOGRAD");
  k->no_winograd_wgrad = on("DN_NO_WINOGRAD_WGRAD");
  k->no_direct = on("DN_NO_DIRECT");
  k->no_thin = on("DN_NO_THIN");
  k->no_thin_conv = on("DN_NO_THIN_CONV");
  k->no_splitk = on("DN_NO_SPLITK");
  k->wino_dbg = num("DN_WINO_DBG", 0);
  k->wino_wg_dbg = num("DN_WINO_WG_DBG", 0);
  k->wino_wgw = num("DN_WINO_WGW", 1);
  k->lds3_dbg = num("DN_LDS3_DBG", 0);
  k->wino_dbgptr = getenv("DN_WINO_DBGPTR") ? strtoull(getenv("DN_WINO_DBGPTR"), nullptr, 0) : 0ull;
  k->wino_min_tiles = num("DN_WINO_MIN_TILES", 192);
  k->no_x3_direct = on("DN_NO_X3_DIRECT");
  k->no_wino_splitk = on("DN_NO_WINO_SPLITK");
  k->wino_splitk_target = num("DN_WINO_SPLITK_TARGET", 512);   // (r05: 256 -> 512 and 128 -> 208 blocks: the 208-block layers of a 4-image shard take two co-resident half-K blocks per CU; b4 3.83 -> 3.74 ms, b8 5.50 -> 5.45, profiles/r05_exp2.txt)
  k->wino_splitk_maxblocks = num("DN_WINO_SPLITK_MAXBLOCKS", 208);
  k->no_wino8_tail = on("DN_NO_WINO8_TAIL");
  k->no_x3_splitk = on("DN_NO_X3_SPLITK");
  k->pack_blocks = num("DN_PACK_BLOCKS", 512);
  if (k->pack_blocks < 1) k->pack_blocks = 1;
  k->no_lds3 = on("DN_NO_LDS3");
  k->no_x3_wgrad = on("DN_NO_X3_WGRAD");
  k->no_tap_windows = on("DN_NO_TAP_WINDOWS");
  k->wino_wg_target = num("DN_WINO_WG_TARGET", 0);
  k->wino_nmajor = num("DN_WINO_NMAJOR", 1);
  k->no_riding_fences = on("DN_NO_RIDING_FENCES");
  k->wino8 = num("DN_WINO8", -1);
  k->wino8_var = num("DN_WINO8_VAR", 0);
}

const Knobs& knobs() {
  if (g_knobs_state.load(std::memory_order_acquire) == 0) {
    std::lock_guard<std::mutex> lock(g_knobs_mu);
    if (g_knobs_state.load(std::memory_order_relaxed) == 0) {
      read_knobs(&g_knobs);
      g_knobs_state.store(1, std::memory_order_release);
    }
  }
  return g_knobs;
}

int build_plan(const dn_conv_desc* d, bool for_wgrad, IgemmParams* p) {
  memset(p, 0, sizeof(*p));
  DN_REQUIRE(d != nullptr, DN_ERR_BAD_ARG, "null conv descriptor");
  DN_REQUIRE(d->kind >= DN_CONV_FWD && d->kind <= DN_CONVT_DGRAD, DN_ERR_BAD_ARG, "bad conv kind %d", d->kind);
  DN_REQUIRE(d->R >= 1 && d->S >= 1 && d->R * d->S <= 49, DN_ERR_UNSUPPORTED, "kernel %dx%d unsupported", d->R, d->S);
  DN_REQUIRE(d->stride >= 1 && d->stride <= 2, DN_ERR_UNSUPPORTED, "stride %d unsupported", d->stride);
  DN_REQUIRE(d->n_in >= 1 && d->n_in <= DN_MAX_OPERANDS, DN_ERR_BAD_ARG, "n_in %d", d->n_in);
  DN_REQUIRE(d->N > 0 && d->IH > 0 && d->IW > 0 && d->OH > 0 && d->OW > 0, DN_ERR_BAD_ARG, "bad sizes");
  int cin_total = 0;
  for (int i = 0; i < d->n_in; ++i) {
    DN_REQUIRE(d->in[i].C > 0 && (d->in[i].up_shift == 0 || d->in[i].up_shift == 1), DN_ERR_BAD_ARG, "operand %d", i);
    cin_total += d->in[i].C;
  }
  p->N = d->N;
  p->R = d->R;
  p->S = d->S;
  p->act = d->act;
  p->act_p0 = d->act_p0;
  p->act_p1 = d->act_p1;
  p->bias = d->bias;
  p->w = d->w_packed;
  p->bn_partial = d->bn_partial;
  if (d->kind == DN_CONV_DGRAD && !for_wgrad && d->bnb_y != nullptr && d->bnb_partial != nullptr) {
    DN_REQUIRE(d->bnb_scale && d->bnb_shift && d->bnb_mean && d->bnb_invstd, DN_ERR_BAD_ARG, "bnb_y needs scale / shift / mean / invstd");
    p->bnb_y = d->bnb_y;
    p->bnb_scale = d->bnb_scale;
    p->bnb_shift = d->bnb_shift;
    p->bnb_mean = d->bnb_mean;
    p->bnb_invstd = d->bnb_invstd;
    p->bnb_partial = d->bnb_partial;
  }
  if (!for_wgrad && d->kind == DN_CONV_FWD) p->recip_out = d->recip_out;
  if (!for_wgrad && d->splitk_ws != nullptr && d->splitk_ws_bytes > 0) {
    p->ks_ws = reinterpret_cast<float*>(d->splitk_ws);
    p->ks_ws_bytes = (size_t)d->splitk_ws_bytes;
    if (d->splitk_ws_bytes >= 8192 && (d->splitk_ws_bytes & 3) == 0) {
      // the last 256 bytes: counters of the last-arrival epilogues (dn_fold.h), out of reach of the K splits' partial tiles
      p->ks_ws_bytes -= 256;
      p->fold_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(d->splitk_ws) + d->splitk_ws_bytes - 256);
    }
  }
  if (!for_wgrad && d->kind == DN_CONV_FWD && d->bn_partial != nullptr && d->bnf_scale != nullptr) {
    DN_REQUIRE(d->bnf_gamma && d->bnf_beta && d->bnf_mean && d->bnf_invstd && d->bnf_shift, DN_ERR_BAD_ARG,
               "bnf_scale needs gamma / beta / mean / invstd / shift");
    DN_REQUIRE((d->bnf_running_mean == nullptr) == (d->bnf_running_var == nullptr), DN_ERR_BAD_ARG, "bnf_running_mean and bnf_running_var go together");
    p->bnf.conv_bias = d->bias;
    p->bnf.gamma = d->bnf_gamma;
    p->bnf.beta = d->bnf_beta;
    p->bnf.running_mean = d->bnf_running_mean;
    p->bnf.running_var = d->bnf_running_var;
    p->bnf.num_batches_tracked = reinterpret_cast<long long*>(d->bnf_num_batches_tracked);
    p->bnf.momentum = d->bnf_momentum;
    p->bnf.eps = d->bnf_eps;
    p->bnf.mean = d->bnf_mean;
    p->bnf.invstd = d->bnf_invstd;
    p->bnf.scale = d->bnf_scale;
    p->bnf.shift = d->bnf_shift;
  }
  if (!for_wgrad && d->kind == DN_CONV_DGRAD && p->bnb_y != nullptr && d->bnb_dgamma != nullptr && d->bnb_dbeta != nullptr) {
    p->bnb_dgamma = d->bnb_dgamma;
    p->bnb_dbeta = d->bnb_dbeta;
  }
  const int st = d->stride, pad = d->pad;
  const int dil = d->dilation > 1 ? d->dilation : 1;
  DN_REQUIRE(d->pad_mode == 0 || d->pad_mode == 1, DN_ERR_BAD_ARG, "bad pad_mode %d", d->pad_mode);
  if (dil > 1) {
    DN_REQUIRE(st == 1 && d->pad_mode == 0 && (d->kind == DN_CONV_FWD || d->kind == DN_CONV_DGRAD), DN_ERR_UNSUPPORTED,
               "dilation %d: stride-1 zero-padded conv forward / input gradient / weight gradient only", dil);
    DN_REQUIRE((d->R - 1) * dil - pad <= 127 && (d->S - 1) * dil - pad <= 127 && pad <= 127, DN_ERR_UNSUPPORTED, "dilation %d: tap offsets beyond +-127", dil);
  }
  if (d->pad_mode == 1) {
    DN_REQUIRE(for_wgrad ? d->kind == DN_CONV_FWD : d->kind == DN_CONV_FWD, DN_ERR_UNSUPPORTED,
               "reflection padding is supported by the conv forward and its weight gradient only");
    DN_REQUIRE(pad < d->IH && pad < d->IW, DN_ERR_BAD_ARG, "reflection pad %d needs a larger input than %dx%d", pad, d->IH, d->IW);
    p->reflect = 1;
  }

  if (for_wgrad) {
    DN_REQUIRE(d->kind == DN_CONV_FWD || d->kind == DN_CONVT_FWD, DN_ERR_BAD_ARG, "wgrad needs a forward descriptor");
    p->nphases = 1;
    p->osy = p->osx = 1;
    p->sy = p->sx = st;
    p->n_is_dim0 = 1;
    int nt = 0;
    for (int r = 0; r < d->R; ++r)
      for (int s = 0; s < d->S; ++s) {
        p->tdy[nt] = (int8_t)(r * dil - pad);
        p->tdx[nt] = (int8_t)(s * dil - pad);
        p->tr[nt] = (int8_t)r;
        p->ts[nt] = (int8_t)s;
        ++nt;
      }
    p->ph[0].ntaps = nt;
    if (d->kind == DN_CONV_FWD) {
      // dW[co][ci][r][s] = sum_{oy,ox} dy[oy,ox][co] * x[oy*st - pad + r, ..][ci]
      int cout = 0;
      for (int i = 0; i < d->n_out; ++i) cout += d->out[i].C;
      DN_REQUIRE(cout > 0, DN_ERR_BAD_ARG, "wgrad: forward descriptor has no output channels");
      p->GH = d->OH;
      p->GW = d->OW;
      p->IH = d->IH;
      p->IW = d->IW;
      p->Ntot = cout;
      p->D0 = cout;
      p->D1 = cin_total;
      p->n_in = d->n_in;
      int off = 0;
      for (int i = 0; i < d->n_in; ++i) {
        fill_operand(&p->in[i], &d->in[i], off);
        off += d->in[i].C;
      }
    } else {
      // dWt[ci][co][r][s] = sum_{iy,ix} x[iy,ix][ci] * dy[iy*st - pad + r, ..][co]   (operand filled by caller)
      DN_REQUIRE(d->n_in == 1 && d->in[0].scale == nullptr && d->in[0].up_shift == 0, DN_ERR_UNSUPPORTED,
                 "conv-transpose wgrad needs one plain input operand");
      int cout = 0;
      for (int i = 0; i < d->n_out; ++i) cout += d->out[i].C;
      p->GH = d->IH;
      p->GW = d->IW;
      p->IH = d->OH;
      p->IW = d->OW;
      p->Ntot = cin_total;
      p->D0 = cin_total;
      p->D1 = cout;
      p->n_in = 1;   // the dy operand; data pointer set by the caller
      p->in[0].C = cout;
      p->in[0].ch_off = 0;
    }
  } else {
    int ntot = 0;
    DN_REQUIRE(d->n_out >= 1 && d->n_out <= DN_MAX_OPERANDS, DN_ERR_BAD_ARG, "n_out %d", d->n_out);
    for (int i = 0; i < d->n_out; ++i) {
      DN_REQUIRE(d->out[i].C > 0 && d->out[i].data != nullptr, DN_ERR_BAD_ARG, "result %d", i);
      ntot += d->out[i].C;
    }
    p->Ntot = ntot;
    p->n_in = d->n_in;
    p->n_out = d->n_out;
    p->IH = d->IH;
    p->IW = d->IW;
    p->OH = d->OH;
    p->OW = d->OW;
    int off = 0;
    for (int i = 0; i < d->n_in; ++i) {
      fill_operand(&p->in[i], &d->in[i], off);
      off += d->in[i].C;
    }
    const bool gather = (d->kind == DN_CONV_FWD || d->kind == DN_CONVT_DGRAD);
    if (gather) {
      p->nphases = 1;
      p->GH = d->OH;
      p->GW = d->OW;
      p->sy = p->sx = st;
      p->osy = p->osx = 1;
      p->n_is_dim0 = 1;
      p->D0 = ntot;
      p->D1 = cin_total;
      int nt = 0;
      for (int r = 0; r < d->R; ++r)
        for (int s = 0; s < d->S; ++s) {
          p->tdy[nt] = (int8_t)(r * dil - pad);
          p->tdx[nt] = (int8_t)(s * dil - pad);
          p->tr[nt] = (int8_t)r;
          p->ts[nt] = (int8_t)s;
          ++nt;
        }
      p->ph[0].ntaps = nt;
    } else {
      // scatter family: out pixel o = g*st + ph; contributing kernel rows r == (ph + pad) mod st; in pixel = g + (ph+pad-r)/st
      p->nphases = st * st;
      p->GH = ceil_div(d->OH, st);
      p->GW = ceil_div(d->OW, st);
      p->sy = p->sx = 1;
      p->osy = p->osx = st;
      p->n_is_dim0 = 0;
      p->D0 = cin_total;
      p->D1 = ntot;
      int nt = 0;
      for (int py = 0; py < st; ++py)
        for (int px = 0; px < st; ++px) {
          KPhase& ph = p->ph[py * st + px];
          ph.tap0 = nt;
          ph.ooy = py;
          ph.oox = px;
          for (int r = 0; r < d->R; ++r) {
            if (((py + pad - r * dil) % st + st) % st != 0) continue;
            for (int s = 0; s < d->S; ++s) {
              if (((px + pad - s * dil) % st + st) % st != 0) continue;
              DN_REQUIRE(nt < kMaxTaps, DN_ERR_UNSUPPORTED, "too many taps");
              p->tdy[nt] = (int8_t)floor_div(py + pad - r * dil, st);
              p->tdx[nt] = (int8_t)floor_div(px + pad - s * dil, st);
              p->tr[nt] = (int8_t)r;
              p->ts[nt] = (int8_t)s;
              ++nt;
            }
          }
          ph.ntaps = nt - ph.tap0;
        }
    }
    for (int i = 0; i < d->n_out; ++i) {
      KResult& r = p->out[i];
      r.p = d->out[i].data;
      r.C = d->out[i].C;
      r.accumulate = d->out[i].accumulate;
      r.sn = d->out[i].stride_n;
      r.sh = d->out[i].stride_h;
      r.sw = d->out[i].stride_w;
      r.n_begin = (i == 0) ? 0 : p->out[i - 1].n_begin + p->out[i - 1].C;
      r.linear = (p->nphases == 1 && r.sh == (long long)d->OW * r.sw && r.sn == (long long)d->OH * r.sh) ? 1 : 0;
    }
    DN_REQUIRE(d->bn_partial == nullptr || (d->n_out == 1 && p->nphases == 1), DN_ERR_UNSUPPORTED,
               "bn_partial needs a single un-phased result");
  }
  long long m = (long long)p->N * p->GH * p->GW;
  DN_REQUIRE(m > 0 && m < (1ll << 31), DN_ERR_UNSUPPORTED, "grid too large");
  p->M = (int)m;
  p->mGW = fastdiv_magic((unsigned)p->GW);
  p->mGH = fastdiv_magic((unsigned)p->GH);
  p->allvec = 1;
  p->uni32 = 1;          // "fast plan": <= 32 taps per phase, zero padding, at least one operand the fast loaders take
  p->any_affine = 0;
  for (int z = 0; z < p->nphases; ++z)
    if (p->ph[z].ntaps > 32) p->uni32 = 0;
  int n_fast = 0, n_uniform = 0;
  for (int i = 0; i < p->n_in; ++i) {
    const KOperand& o = p->in[i];
    if (o.scale != nullptr) p->any_affine = 1;
    // (the conv-transpose wgrad operand is filled in by the caller, which re-evaluates these flags)
    p->in[i].small = offsets_fit_int32(p->in[i], p->N, p->IH, p->IW) && ((long long)p->N * (o.sn < 0 ? -o.sn : o.sn) + 16) * 4 < (1ll << 31) ? 1 : 0;
    if (!(p->in[i].vec && p->in[i].small)) p->allvec = 0;
    if (p->in[i].vec && p->in[i].small && o.up == 0 && (o.C % 32 == 0 || o.C == 4 || o.C == 8 || o.C == 16)) ++n_fast;
    if (p->in[i].vec && p->in[i].small && o.up == 0 && o.C % 32 == 0) ++n_uniform;
  }
  (void)n_fast;
  if (p->reflect) p->uni32 = 0;
  for (int i = 0; i < p->n_in; ++i)
    if (!p->in[i].small) p->uni32 = 0;
  // weight-gradient fast path: <= 32 taps, zero padding, every operand within 2 GiB; float4 operands of any C % 4 == 0,
  // scalar operands only without a pending affine, taps within a signed byte
  p->wg_uniform = (!p->reflect && p->ph[0].ntaps <= 32) ? 1 : 0;
  for (int i = 0; i < p->n_in; ++i) {
    const KOperand& o = p->in[i];
    if (!o.small || (!o.vec && o.scale != nullptr) || o.C >= 32768) p->wg_uniform = 0;
  }
  (void)n_uniform;
  p->tile_store = false ? 0 : (false ? 1 : 2);
  // (0 = DN_COMPUTE_DEFAULT = the three-piece arithmetic; unknown values: the fp32 instruction)
  p->compute = norm_compute(d->compute);
  p->BN = pick_bn(p->Ntot);
  p->Npad = ceil_div(p->Ntot, p->BN) * p->BN;
  long long woff = 0;
  for (int z = 0; z < p->nphases; ++z) {
    int nch = 0;
    for (int i = 0; i < p->n_in; ++i) nch += ceil_div(p->ph[z].ntaps * p->in[i].C, kChunk);
    p->ph[z].nchunks = nch;
    p->ph[z].w_off = woff;
    woff += (long long)p->Npad * nch * kChunk;
  }
  return DN_OK;
}

}  // namespace dn

extern "C" {

int dn_version(void) { return 17; }

void dn_reload_knobs(void) {
  std::lock_guard<std::mutex> lock(dn::g_knobs_mu);
  dn::read_knobs(&dn::g_knobs);
  dn::g_knobs_state.store(1, std::memory_order_release);
}

const char* dn_last_error(void) { return dn::g_err; }

const char* dn_last_kernel(void) { return dn::g_kernel; }

int dn_device_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    dn::set_error("hipGetDevice failed (no HIP device?)");
    return DN_ERR_LAUNCH;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    dn::set_error("hipGetDeviceProperties failed");
    return DN_ERR_LAUNCH;
  }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int64_t dn_conv_packed_weight_elems(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (dn::build_plan(d, false, &p) != DN_OK) return -1;
  if (const int wl = dn::wino_layout(d, p)) return dn::wino_packed_floats(p, wl);
  const dn::KPhase& last = p.ph[p.nphases - 1];
  return last.w_off + (int64_t)p.Npad * last.nchunks * dn::kChunk;
}

int32_t dn_conv_weight_layout(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (dn::build_plan(d, false, &p) != DN_OK) return -1;
  return dn::wino_layout(d, p);
}

int32_t dn_conv_bn_partial_rows(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (dn::build_plan(d, false, &p) != DN_OK) return -1;
  return (p.M + 127) / 128;
}

int32_t dn_conv_fwd_fuses_reciprocal(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (d == nullptr || d->kind != DN_CONV_FWD || dn::build_plan(d, false, &p) != DN_OK) return 0;
  return dn::head_fwd_fuses_reciprocal(d, p) ? 1 : 0;
}

int32_t dn_conv_dgrad_fuses_bn_sums(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (d == nullptr || dn::build_plan(d, false, &p) != DN_OK) return -1;
  if (d->kind != DN_CONV_DGRAD || false) return 0;
  if (dn::wino_layout(d, p) == 0) return 0;
  const dn_result& r = d->out[0];
  const bool dense = d->n_out == 1 && !r.accumulate && (r.C & 3) == 0 && r.stride_w == r.C && r.stride_h == (int64_t)d->OW * r.C &&
                     r.stride_n == (int64_t)d->OH * d->OW * r.C;
  return (dense && d->bias == nullptr && d->act == DN_ACT_NONE) ? 1 : 0;
}

int32_t dn_conv_fwd_folds_bn_finalize(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (d == nullptr || dn::build_plan(d, false, &p) != DN_OK) return -1;
  if (d->kind != DN_CONV_FWD || dn::wino_layout(d, p) == 0) return 0;
  p.T = p.M / 4;
  return dn::wino_folds_bn_finalize(p) ? 1 : 0;
}

int32_t dn_conv_dgrad_folds_bn_sums(const dn_conv_desc* d) {
  if (dn_conv_dgrad_fuses_bn_sums(d) != 1) return 0;
  dn::IgemmParams p;
  if (dn::build_plan(d, false, &p) != DN_OK) return -1;
  p.T = p.M / 4;
  return dn::wino_folds_bn_sums(p) ? 1 : 0;
}

int64_t dn_conv_splitk_workspace_bytes(const dn_conv_desc* d) {
  dn::IgemmParams p;
  if (d == nullptr || dn::build_plan(d, false, &p) != DN_OK) return -1;
  if (dn::wino_layout(d, p) == 3) return (int64_t)dn::wino_splitk_workspace_bytes(p);
  if (dn::wino_layout(d, p) != 0) return 0;
  return (int64_t)dn::conv_x3_splitk_workspace_upper_bytes(p);
}

// Test/diagnostic hook (host only, no device work): dump the plan as int32s.
//  [0]=nphases [1]=GH [2]=GW [3]=sy [4]=osy [5]=Ntot [6]=Npad [7]=BN [8]=n_is_dim0 [9]=D0 [10]=D1 [11]=M
//  then per phase: ntaps, ooy, oox, nchunks, w_off, followed by ntaps x (dy, dx, r, s)
int dn_debug_conv_plan(const dn_conv_desc* d, int for_wgrad, int32_t* out, int cap) {
  dn::IgemmParams p;
  int rc = dn::build_plan(d, for_wgrad != 0, &p);
  if (rc != DN_OK) return rc;
  int n = 0;
  auto put = [&](long long v) {
    if (n < cap) out[n] = (int32_t)v;
    ++n;
  };
  put(p.nphases); put(p.GH); put(p.GW); put(p.sy); put(p.osy); put(p.Ntot); put(p.Npad); put(p.BN);
  put(p.n_is_dim0); put(p.D0); put(p.D1); put(p.M);
  for (int z = 0; z < p.nphases; ++z) {
    const dn::KPhase& ph = p.ph[z];
    put(ph.ntaps); put(ph.ooy); put(ph.oox); put(ph.nchunks); put(ph.w_off);
    for (int t = 0; t < ph.ntaps; ++t) {
      put(p.tdy[ph.tap0 + t]); put(p.tdx[ph.tap0 + t]); put(p.tr[ph.tap0 + t]); put(p.ts[ph.tap0 + t]);
    }
  }
  return n;
}

}  // extern "C"
