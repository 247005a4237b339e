// capi_model.hip -- errors and device selection, model construction (arrays -> the packed device blob),
// accessors, and the operand exponents of the f16 modes.  Host C++ over the HIP runtime; no CPU compute path.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <cmath>

#include <algorithm>
#include <string>
#include <mutex>
#include <utility>
#include <unordered_set>
#include <vector>

#include "pk_host.h"

using namespace pkmi;
using namespace pkhost;

// ------------------------------------------------------------------ errors

namespace {
thread_local char g_err[512] = "";
thread_local int g_device = 0;
}  // namespace

namespace pkhost {

int Fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
const char *LastError() { return g_err; }
int CurrentDevice() { return g_device; }

int UseDevice(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return Fail(PK_MI355_E_DEVICE, "no HIP device available (libpk_mi355 has no CPU fallback)");
  if (device < 0 || device >= n) return Fail(PK_MI355_E_INVALID, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  return 0;
}

ModelKnobs ReadModelKnobs() {
  ModelKnobs k;
  if (const char *c = getenv("PK_MI355_FUSED_TAIL32")) k.wave_tail32 = atoi(c) != 0;
  if (const char *c = getenv("PK_MI355_FUSED_TAIL_MIN_TILES")) k.fused_tail_min_tiles = std::max(384, atoi(c));   // (below 384 tiles the small-tile kernel runs: no tail variant)
  if (const char *c = getenv("PK_MI355_TAIL_WALK")) k.tail_walk = std::max(0, atoi(c));
  if (const char *c = getenv("PK_MI355_TAIL_STRIP")) k.tail_strip = atoi(c) != 0;
  if (const char *c = getenv("PK_MI355_L1_RING")) k.l1_ring = atoi(c) == 3 ? 3 : 2;
  return k;
}

// host mirror of the exponent words <- device (after a broadcast, or a write through the blob's device pointer)
int RefreshExps(pk_mi355_am *am) {
  if (!IsF16(am->precision) || !am->d_blob) { am->exps_stale = false; return 0; }
  HIP_TRY(hipMemcpy(am->h_exps.data(), am->d_blob + am->exp_off, sizeof(int32_t) * am->h_exps.size(), hipMemcpyDeviceToHost));
  am->exps_stale = false;
  return 0;
}
int UploadExps(pk_mi355_am *am) {
  HIP_TRY(hipMemcpy(am->d_blob + am->exp_off, am->h_exps.data(), sizeof(int32_t) * am->h_exps.size(), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace pkhost

extern "C" {

const char *pk_mi355_last_error(void) { return g_err; }
const char *pk_mi355_version(void) { return "pk_mi355 0.1 (gfx950)"; }

int pk_mi355_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pk_mi355_set_device(int device) {
  int rc = UseDevice(device);
  if (rc == 0) g_device = device;
  return rc;
}

// ------------------------------------------------------------------ model

pk_mi355_am_t *pk_mi355_am_create(void) {
  pk_mi355_am *am = new pk_mi355_am();
  am->device = g_device;
  am->knobs = ReadModelKnobs();
  return am;
}

void pk_mi355_am_destroy(pk_mi355_am_t *am) {
  if (!am) return;
  hipSetDevice(am->device);
  if (am->proc) pk_mi355_batch_destroy(am->proc);
  FreeWorkspace(am->ws);
  hipFree(am->d_blob);
  hipFree(am->d_tid2pdf);
  delete am;
}

int pk_mi355_am_add_linear(pk_mi355_am_t *am, int in_dim, int out_dim, const float *W,
                           const float *b) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "model is finalized");
  if (in_dim <= 0 || out_dim <= 0 || !W || !b) return Fail(PK_MI355_E_INVALID, "bad linear layer");
  HostLayer L;
  L.type = PK_NNET_LINEAR_LAYER;
  L.in_dim = in_dim;
  L.out_dim = out_dim;
  L.W.assign(W, W + (size_t)in_dim * out_dim);
  L.b.assign(b, b + out_dim);
  am->layers.push_back(std::move(L));
  return 0;
}

int pk_mi355_am_add_layer(pk_mi355_am_t *am, int layer_type) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "model is finalized");
  if (layer_type != PK_NNET_RELU_LAYER && layer_type != PK_NNET_NORMALIZE_LAYER &&
      layer_type != PK_NNET_SOFTMAX_LAYER)   // nnet.cc:106-127 accepts only kinds 0..3
    return Fail(PK_MI355_E_INVALID, "unexpected layer type: %d", layer_type);
  HostLayer L;
  L.type = layer_type;
  am->layers.push_back(L);
  return 0;
}

int pk_mi355_am_set_precision(pk_mi355_am_t *am, int precision) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "set the precision before finalizing the model");
  if (precision != PK_MI355_PRECISION_F32 && precision != PK_MI355_PRECISION_F16X3 && precision != PK_MI355_PRECISION_F16)
    return Fail(PK_MI355_E_INVALID, "unknown precision %d", precision);
  am->precision = precision;
  return 0;
}

int pk_mi355_am_precision(const pk_mi355_am_t *am) { return am ? am->precision : 0; }

int pk_mi355_am_set_softmax(pk_mi355_am_t *am, int mode) {
  if (!am) return Fail(PK_MI355_E_INVALID, "null model");
  if (mode != PK_MI355_SOFTMAX_STABLE && mode != PK_MI355_SOFTMAX_REFERENCE)
    return Fail(PK_MI355_E_INVALID, "unknown softmax mode %d", mode);
  am->softmax_reference = mode == PK_MI355_SOFTMAX_REFERENCE;
  return 0;
}

int pk_mi355_am_softmax(const pk_mi355_am_t *am) {
  return am && am->softmax_reference ? PK_MI355_SOFTMAX_REFERENCE : PK_MI355_SOFTMAX_STABLE;
}

int pk_mi355_am_finalize(pk_mi355_am_t *am, const float *prior, int num_pdfs, int left_context,
                         int right_context, const int32_t *tid2pdf, int num_tids) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "model already finalized");
  if (left_context < 0 || right_context < 0) return Fail(PK_MI355_E_INVALID, "negative context");
  int rc = UseDevice(am->device);
  if (rc) return rc;

  // dimension chain
  const bool f16 = IsF16(am->precision);
  if (f16) {
    // the split-fp16 path covers (Linear [ReLU] [Normalize])+ [Softmax]: the BASELINE model family and the
    // relu + renormalize stacks tool/convert_am.py writes (a Normalize must be followed by another Linear)
    const int nl = (int)am->layers.size();
    bool ok = nl > 0 && am->layers[0].type == PK_NNET_LINEAR_LAYER;
    for (int i = 0; ok && i < nl; ++i) {
      const int t = am->layers[i].type;
      if (t == PK_NNET_RELU_LAYER) ok = i > 0 && am->layers[i - 1].type == PK_NNET_LINEAR_LAYER;
      else if (t == PK_NNET_NORMALIZE_LAYER)
        ok = i > 0 && i + 1 < nl && am->layers[i + 1].type == PK_NNET_LINEAR_LAYER &&
             (am->layers[i - 1].type == PK_NNET_LINEAR_LAYER ||
              (am->layers[i - 1].type == PK_NNET_RELU_LAYER && i >= 2 && am->layers[i - 2].type == PK_NNET_LINEAR_LAYER));
      else if (t == PK_NNET_SOFTMAX_LAYER) ok = (i == nl - 1);
      else if (t != PK_NNET_LINEAR_LAYER) ok = false;
    }
    if (!ok) return Fail(PK_MI355_E_INVALID, "f16x3 / f16 precision supports (Linear [ReLU] [Normalize])+ [Softmax] networks only");
  }
  int first_in = 0, dim = 0;
  am->lin.clear();
  am->flops_per_frame = 0;
  size_t off = 0;
  int max_pad = 0;
  for (auto &L : am->layers) {
    if (L.type != PK_NNET_LINEAR_LAYER) continue;
    if (first_in == 0) { first_in = L.in_dim; }
    else if (L.in_dim != dim)
      return Fail(PK_MI355_E_INVALID, "layer dimension mismatch: %d after %d", L.in_dim, dim);
    dim = L.out_dim;
    DevLinear D;
    D.K = L.in_dim; D.N = L.out_dim;
    if (f16) {
      D.Kpad = (int)RoundUp(D.K, kBKF16);
      D.Npad = (int)RoundUp(D.N, kTileF16);
      D.wt_off = off; off += (size_t)D.Kpad * D.Npad;         // 2 Kpad halves per row = Kpad floats
    } else {
      D.Kpad = (int)RoundUp(D.K, kBK);
      D.Npad = (int)RoundUp(D.N, kTile);
      D.wt_off = off; off += (size_t)D.Kpad * D.Npad;
    }
    D.b_off = off;  off += D.Npad;
    am->lin.push_back(D);
    am->flops_per_frame += 2.0 * D.K * D.N;
    max_pad = std::max(max_pad, D.Npad);
    max_pad = std::max(max_pad, (int)RoundUp(D.K, f16 ? kTileF16 : kTile));
  }
  if (am->lin.empty()) {
    // layer-only networks (the nnet_test.cc micro-tests): the width is the pdf count
    if (num_pdfs <= 0) return Fail(PK_MI355_E_INVALID, "cannot infer the width of a network without linear layers");
    first_in = dim = num_pdfs;
    max_pad = (int)RoundUp(dim, kTile);
  }
  if (num_pdfs > 0 && num_pdfs != dim)
    return Fail(PK_MI355_E_INVALID, "num_pdfs = %d but the network outputs %d", num_pdfs, dim);
  am->input_dim = first_in;
  am->output_dim = dim;
  am->num_pdfs = dim;
  am->left = left_context;
  am->right = right_context;
  const int ctx = left_context + right_context + 1;
  if (first_in % ctx != 0)
    return Fail(PK_MI355_E_INVALID, "input width %d is not a multiple of the context %d", first_in, ctx);
  am->feat_dim = first_in / ctx;
  am->max_dim_pad = max_pad;
  am->logprior_off = off;
  off += RoundUp(dim, 4);
  am->exp_off = off;
  const int nlin = (int)am->lin.size();
  am->h_exps.assign(2 * nlin + 1, 0);
  if (f16) off += RoundUp(2 * nlin + 1, 4);
  am->blob_floats = off;

  // pack: W^T zero-padded to [Kpad][Npad] (nnet.cc:16-17 keeps the transpose),
  // bias padded, log prior (am.cc:43: logf of the probabilities)
  std::vector<float> blob(off, 0.0f);
  size_t li = 0;
  for (auto &L : am->layers) {
    if (L.type != PK_NNET_LINEAR_LAYER) continue;
    const DevLinear &D = am->lin[li++];
    if (f16) {
      // W stays [out][in] (k contiguous), split into fp16 hi and lo = fp16(w - hi), the
      // pairs interleaved in chunks of 8 k's (see gemm_f16.hip).
      // Range safety: the layer's weights are first multiplied by 2^w_exp -- exact in fp32 -- chosen so that
      // max |W| lands in [2^13, 2^14): every weight within 2^-16 of the largest then has a NORMAL lo half
      // (lo = fp16(w - hi) needs |w| >= 2^-3 for that), whatever the scale the model was trained at; the GEMM's
      // epilogue multiplies by 2^-w_exp again.  Unscaled, He-normal weights of a K = 1024 layer (max ~0.2) already
      // have subnormal lo halves, and at 2^-8 of that scale the mode is no better than plain fp16.
      float wmax = 0.0f;
      for (size_t i = 0; i < (size_t)D.N * D.K; ++i) {
        if (!std::isfinite(L.W[i]))
          return Fail(PK_MI355_E_INVALID, "affine layer %d holds a non-finite weight (f16x3 / f16 precision cannot carry it)", (int)li - 1);
        wmax = std::max(wmax, fabsf(L.W[i]));
      }
      // PK_MI355_NO_PRESCALE=1: measurement switch only (what the prescale costs in clock: normal lo halves toggle
      // more bits than subnormal ones, and this mode is power-limited) -- never set it in production
      static const bool no_prescale = [] { const char *e = getenv("PK_MI355_NO_PRESCALE"); return e && atoi(e) != 0; }();
      const int w_exp = (wmax > 0.0f && !no_prescale) ? std::min(60, std::max(-60, 13 - ilogbf(wmax))) : 0;
      am->h_exps[li - 1] = w_exp;
      const float w_scale = ldexpf(1.0f, w_exp);
      _Float16 *w2 = reinterpret_cast<_Float16 *>(blob.data() + D.wt_off);
      for (int o = 0; o < D.N; ++o)
        for (int k = 0; k < D.K; ++k) {
          float v = L.W[(size_t)o * D.K + k] * w_scale;
          v = std::min(std::max(v, -65504.0f), 65504.0f);
          const _Float16 hi = static_cast<_Float16>(v);
          _Float16 *dst = w2 + (size_t)o * 2 * D.Kpad + (k >> 3) * 16 + (k & 7);
          dst[0] = hi;
          dst[8] = static_cast<_Float16>(v - static_cast<float>(hi));
        }
    } else {
      float *wt = blob.data() + D.wt_off;
      for (int o = 0; o < D.N; ++o)
        for (int k = 0; k < D.K; ++k) wt[(size_t)k * D.Npad + o] = L.W[(size_t)o * D.K + k];
    }
    memcpy(blob.data() + D.b_off, L.b.data(), sizeof(float) * D.N);
  }
  for (int i = 0; i < dim; ++i)
    blob[am->logprior_off + i] = prior ? logf(prior[i]) : 0.0f;
  if (f16) memcpy(blob.data() + am->exp_off, am->h_exps.data(), sizeof(int32_t) * am->h_exps.size());
  HIP_TRY(hipMalloc(&am->d_blob, sizeof(float) * off));
  HIP_TRY(hipMemcpy(am->d_blob, blob.data(), sizeof(float) * off, hipMemcpyHostToDevice));

  am->tid2pdf.clear();
  if (tid2pdf && num_tids > 0) {
    am->tid2pdf.assign(tid2pdf, tid2pdf + num_tids);
    HIP_TRY(hipMalloc(&am->d_tid2pdf, sizeof(int32_t) * num_tids));
    HIP_TRY(hipMemcpy(am->d_tid2pdf, tid2pdf, sizeof(int32_t) * num_tids, hipMemcpyHostToDevice));
  }
  am->finalized = true;
  return 0;
}

int pk_mi355_am_num_pdfs(const pk_mi355_am_t *am) { return am ? am->num_pdfs : 0; }
int pk_mi355_am_input_dim(const pk_mi355_am_t *am) { return am ? am->input_dim : 0; }
int pk_mi355_am_transition_to_pdf(const pk_mi355_am_t *am, int trans_id) {
  if (am->tid2pdf.empty()) return trans_id;
  return am->tid2pdf[trans_id];
}
void *pk_mi355_am_blob_device_ptr(pk_mi355_am_t *am) {
  if (!am) return nullptr;
  if (IsF16(am->precision)) am->exps_stale = true;   // the caller may write the blob (a broadcast through another
                                                     // library): the host mirror of its exponent words is re-read on need
  return am->d_blob;
}
size_t pk_mi355_am_blob_bytes(const pk_mi355_am_t *am) { return am ? am->blob_floats * sizeof(float) : 0; }
double pk_mi355_am_flops_per_frame(const pk_mi355_am_t *am) { return am ? am->flops_per_frame : 0; }

// ------------------------------------------------------------------ f16 modes: operand exponents

int pk_mi355_am_get_exponents(pk_mi355_am_t *am, int32_t *w_exp, int32_t *x_exp, int capacity) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  const int nlin = (int)am->lin.size();
  if (capacity < nlin) return Fail(PK_MI355_E_INVALID, "room for %d exponents, the model has %d affine layers", capacity, nlin);
  int rc = UseDevice(am->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(am->mu);
  if (am->exps_stale && (rc = RefreshExps(am))) return rc;
  for (int l = 0; l < nlin; ++l) {
    if (w_exp) w_exp[l] = am->h_exps[l];
    if (x_exp) x_exp[l] = am->h_exps[nlin + l];
  }
  return nlin;
}

int pk_mi355_am_set_input_exponents(pk_mi355_am_t *am, const int32_t *x_exp, int count) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  if (!IsF16(am->precision)) return Fail(PK_MI355_E_STATE, "operand exponents exist in the f16x3 / f16 precisions only");
  const int nlin = (int)am->lin.size();
  if (!x_exp || count != nlin) return Fail(PK_MI355_E_INVALID, "%d exponents expected (one per affine layer)", nlin);
  for (int l = 0; l < nlin; ++l)
    if (x_exp[l] < -kMaxXExp || x_exp[l] > kMaxXExp) return Fail(PK_MI355_E_INVALID, "exponent %d out of [-%d, %d]", x_exp[l], kMaxXExp, kMaxXExp);
  int rc = UseDevice(am->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(am->mu);
  if (am->exps_stale && (rc = RefreshExps(am))) return rc;
  for (int l = 0; l < nlin; ++l) am->h_exps[nlin + l] = x_exp[l];
  return UploadExps(am);
}

void *pk_mi355_device_malloc(size_t bytes) {
  if (UseDevice(g_device)) return nullptr;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) { Fail(PK_MI355_E_DEVICE, "hipMalloc: %s", hipGetErrorString(e)); return nullptr; }
  return p;
}

void pk_mi355_device_free(void *p) { hipFree(p); }

void *pk_mi355_host_malloc(size_t bytes) {
  if (UseDevice(g_device)) return nullptr;
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { Fail(PK_MI355_E_DEVICE, "hipHostMalloc: %s", hipGetErrorString(e)); return nullptr; }
  return p;
}

void pk_mi355_host_free(void *p) { hipHostFree(p); }

int pk_mi355_memcpy(void *dst, const void *src, size_t bytes, int kind) {
  hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (kind < 1 || kind > 3) return Fail(PK_MI355_E_INVALID, "memcpy kind must be 1 (H2D), 2 (D2H) or 3 (D2D)");
  HIP_TRY(hipMemcpy(dst, src, bytes, k));
  return 0;
}

}  // extern "C"
