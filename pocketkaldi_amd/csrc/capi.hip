// capi.hip -- the C ABI of libpk_mi355.so (include/pk_mi355.h): model management,
// the layer executor, the batched scorer and the reference-compatible
// pk_decodable_* functions.  Host C++ over the HIP runtime; no CPU compute path.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <mutex>
#include <utility>
#include <unordered_set>
#include <vector>

#include "../../include/pk_mi355.h"
#include "pk_kernels.h"
#include "pk_tables.h"

using namespace pkmi;

// ------------------------------------------------------------------ errors

namespace {

thread_local char g_err[512] = "";
// Like hipSetDevice, the selected device is a per-thread setting (a worker thread that never
// called pk_mi355_set_device creates its objects on device 0).
thread_local int g_device = 0;

int Fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return Fail(PK_MI355_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

inline int64_t RoundUp(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// the two precisions that run on the fp16 matrix cores share layouts and code paths
inline bool IsF16(int precision) { return precision == PK_MI355_PRECISION_F16X3 || precision == PK_MI355_PRECISION_F16; }

int UseDevice(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return Fail(PK_MI355_E_DEVICE, "no HIP device available (libpk_mi355 has no CPU fallback)");
  if (device < 0 || device >= n) return Fail(PK_MI355_E_INVALID, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  return 0;
}

// ------------------------------------------------------------------ timing

struct Timer {
  bool enabled = false;
  struct Rec { int kind; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t Get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  void Reset() {
    for (auto &r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
    recs.clear();
  }
  int Begin(int kind, hipStream_t s) {
    if (!enabled) return -1;
    Rec r{kind, Get(), Get()};
    hipEventRecord(r.a, s);
    recs.push_back(r);
    return (int)recs.size() - 1;
  }
  void End(int id, hipStream_t s) {
    if (id >= 0) hipEventRecord(recs[id].b, s);
  }
  ~Timer() {
    Reset();
    for (auto e : pool) hipEventDestroy(e);
  }
};

struct Scoped {
  Timer *t; int id; hipStream_t s;
  Scoped(Timer *t_, int kind, hipStream_t s_) : t(t_), id(t_ ? t_->Begin(kind, s_) : -1), s(s_) {}
  ~Scoped() { if (t) t->End(id, s); }
};

// ------------------------------------------------------------------ section files
// "VEC0" i32 bytes(=4n+4) i32 n, n x 4 bytes (vector.cc:393-425);
// "MAT0" i32 8, i32 rows, i32 cols, rows x VEC0 (matrix.cc:288-319);
// "NNT0" i32 4, i32 layers; "LAY0" i32 4, i32 type [+ MAT0 W, VEC0 b] (nnet.cc:80-147)

struct FileBuf {
  std::vector<unsigned char> d;
  size_t pos = 0;
  std::string path;
  int Open(const char *p) {
    path = p;
    FILE *f = fopen(p, "rb");
    if (!f) return Fail(PK_MI355_E_IO, "cannot open %s", p);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    d.resize(n > 0 ? n : 0);
    size_t got = n > 0 ? fread(d.data(), 1, n, f) : 0;
    fclose(f);
    if ((long)got != n) return Fail(PK_MI355_E_IO, "short read on %s", p);
    return 0;
  }
  bool Tag(const char *t) {
    if (pos + 4 > d.size() || memcmp(&d[pos], t, 4) != 0) return false;
    pos += 4;
    return true;
  }
  bool I32(int32_t *v) {
    if (pos + 4 > d.size()) return false;
    memcpy(v, &d[pos], 4);
    pos += 4;
    return true;
  }
  template <typename T>
  int Vec(std::vector<T> *out) {
    int32_t bytes, n;
    if (!Tag("VEC0") || !I32(&bytes) || !I32(&n))
      return Fail(PK_MI355_E_IO, "VEC0 section expected in %s", path.c_str());
    if (n < 0 || bytes != n * 4 + 4 || pos + (size_t)n * 4 > d.size())
      return Fail(PK_MI355_E_IO, "corrupted VEC0 section in %s", path.c_str());
    out->resize(n);
    if (n) memcpy(out->data(), &d[pos], (size_t)n * 4);
    pos += (size_t)n * 4;
    return 0;
  }
};

}  // namespace

// ------------------------------------------------------------------ model

struct HostLayer {
  int type = 0;
  int in_dim = 0, out_dim = 0;
  std::vector<float> W;   // [out][in]
  std::vector<float> b;
};

struct DevLinear {
  int K = 0, N = 0, Kpad = 0, Npad = 0;
  size_t wt_off = 0, b_off = 0;    // float offsets into the blob (f16x3: wt_off = interleaved
                                   // (hi, lo) rows [Npad][2 Kpad] halves)
};

struct Workspace;   // forward

struct pk_mi355_am {
  int device = 0;
  std::vector<HostLayer> layers;
  bool finalized = false;
  int precision = PK_MI355_PRECISION_F32;
  bool softmax_reference = false;  // PK_MI355_SOFTMAX_REFERENCE: the reference's softmax operations one by one
  int left = 0, right = 0, num_pdfs = 0;
  int input_dim = 0, output_dim = 0, feat_dim = 0;
  int max_dim_pad = 0;             // widest activation, rounded to the tile
  std::vector<int32_t> tid2pdf;
  int32_t *d_tid2pdf = nullptr;    // device copy for the on-GPU gather
  std::vector<DevLinear> lin;      // one per linear layer, in order
  float *d_blob = nullptr;
  size_t blob_floats = 0;
  size_t logprior_off = 0;
  // f16x3 / f16: the operand exponents, int32 words INSIDE the blob (so the one broadcast carries them):
  // [w_exp of linear layer 0 .. n-1 | x_exp of the operand of linear layer 0 .. n-1 | 0].  The kernels read the
  // device words; h_exps mirrors them on the host (refreshed from the device after a broadcast).
  size_t exp_off = 0;
  std::vector<int32_t> h_exps;
  bool exps_stale = false;
  double flops_per_frame = 0;
  // The reference's pk_decodable_init / AcousticModel::Compute allocate per call and are re-entrant
  // for a shared model (nnet.cc:149-163 is const); here the single-utterance entry points share
  // one device workspace per model, so they serialise on this mutex instead.
  std::mutex mu;
  Workspace *ws = nullptr;         // single-utterance workspace of pk_decodable_init (under mu)
  struct pk_mi355_batch *proc = nullptr;   // cached 1-utterance scorer of pk_mi355_process_acoustic (under mu)
  int64_t proc_cap = 0;
  float proc_stats[41] = {0};
};

namespace {

// Activation buffers for one chunk of at most `rows_cap` frames.
struct ExecBufs {
  float *in = nullptr;    // plain (non-spliced) feature-major input, [Kpad0][rows_cap]
  float *a = nullptr;     // ping
  float *b = nullptr;     // pong
  int64_t rows_cap = 0;
  int64_t in_floats = 0, act_floats = 0;
  // f16x3 mode: interleaved (hi, lo) activation rows [rows_cap][2 max Npad], ping/pong; the
  // plain input rows [rows_cap][2 Kpad0]; `a` holds the fp32 logits, `b` softmax probabilities
  _Float16 *h[2] = {nullptr, nullptr};
  _Float16 *xin = nullptr;
  // f16x3 / f16: range words of the operand of every linear layer ([num linear][kRangeSlots], gemm_f16.hip:
  // PublishRange) and their page-locked host mirror, read after the call (EvalRange)
  uint32_t *range = nullptr, *h_range = nullptr;
  int range_words = 0;
  unsigned *row_done = nullptr;   // fused tail (gemm.hip, TAIL variant): one arrival counter per 128-row tile, zero between launches
};

int AllocExec(const pk_mi355_am *am, int64_t rows_cap, ExecBufs *e) {
  e->rows_cap = rows_cap;
  if (IsF16(am->precision)) {
    const size_t act = sizeof(_Float16) * 2 * (size_t)am->max_dim_pad * rows_cap;
    const size_t xin = sizeof(_Float16) * 2 * (size_t)RoundUp(am->input_dim, kBKF16) * (rows_cap + 16);
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipMalloc(&e->h[i], act));
      HIP_TRY(hipMemset(e->h[i], 0, act));
    }
    HIP_TRY(hipMalloc(&e->xin, xin));
    HIP_TRY(hipMemset(e->xin, 0, xin));
    e->range_words = (int)am->lin.size() * kRangeSlots;
    HIP_TRY(hipMalloc(&e->range, sizeof(uint32_t) * e->range_words));
    HIP_TRY(hipMemset(e->range, 0, sizeof(uint32_t) * e->range_words));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_range), sizeof(uint32_t) * e->range_words, hipHostMallocDefault));
    memset(e->h_range, 0, sizeof(uint32_t) * e->range_words);
  }
  {
    const size_t nd = sizeof(unsigned) * (size_t)(rows_cap / kTile + 2);
    HIP_TRY(hipMalloc(&e->row_done, nd));
    HIP_TRY(hipMemset(e->row_done, 0, nd));
  }
  e->act_floats = (int64_t)am->max_dim_pad * rows_cap;
  e->in_floats = RoundUp(am->input_dim, kBK) * rows_cap;
  HIP_TRY(hipMalloc(&e->a, e->act_floats * sizeof(float)));
  HIP_TRY(hipMalloc(&e->b, e->act_floats * sizeof(float)));
  HIP_TRY(hipMalloc(&e->in, e->in_floats * sizeof(float)));
  HIP_TRY(hipMemset(e->a, 0, e->act_floats * sizeof(float)));
  HIP_TRY(hipMemset(e->b, 0, e->act_floats * sizeof(float)));
  HIP_TRY(hipMemset(e->in, 0, e->in_floats * sizeof(float)));
  return 0;
}

void FreeExec(ExecBufs *e) {
  hipFree(e->a);
  hipFree(e->b);
  hipFree(e->in);
  for (int i = 0; i < 2; ++i) hipFree(e->h[i]);
  hipFree(e->xin);
  hipFree(e->range);
  hipFree(e->row_done);
  if (e->h_range) hipHostFree(e->h_range);
  *e = ExecBufs();
}

// Where the result of RunLayers ended up.
struct ExecResult {
  const float *data = nullptr;   // frame-major rows
  int64_t ld = 0;
  int dim = 0;
};

// ---- f16x3 / f16 operand exponents and range words
inline const int32_t *ExpBase(const pk_mi355_am *am) { return reinterpret_cast<const int32_t *>(am->d_blob + am->exp_off); }
inline const int32_t *ExpW(const pk_mi355_am *am, int l) { return ExpBase(am) + l; }
inline const int32_t *ExpX(const pk_mi355_am *am, int l) { return ExpBase(am) + am->lin.size() + l; }
inline const int32_t *ExpZero(const pk_mi355_am *am) { return ExpBase(am) + 2 * am->lin.size(); }
inline uint32_t *RangeOf(const ExecBufs &e, int l) { return e.range ? e.range + (size_t)l * kRangeSlots : nullptr; }

constexpr int kMaxXExp = 30;                     // |operand exponent| (|w_exp| <= 60: 2^(e_out - e_in - e_w) stays a normal float)
constexpr float kRangeSaturated = 65504.0f;      // the split clamps here (gemm_f16.hip: Split)
constexpr float kRangeTooSmall = 0.03125f;       // 2^-5: below this every lo half of the operand is an fp16 subnormal
                                                 // (|lo| <= 2^-12 |x|), and the mode degrades towards plain fp16

int BeginRange(const ExecBufs &e, hipStream_t s) {
  if (e.range) HIP_TRY(hipMemsetAsync(e.range, 0, sizeof(uint32_t) * e.range_words, s));
  return 0;
}
int CollectRange(const ExecBufs &e, hipStream_t s) {
  if (e.range) HIP_TRY(hipMemcpyAsync(e.h_range, e.range, sizeof(uint32_t) * e.range_words, hipMemcpyDeviceToHost, s));
  return 0;
}
// max |hi| of operand l over the call whose range words have been collected (and its stream synchronised)
float RangeMax(const ExecBufs &e, int l) {
  uint32_t m = 0;
  for (int i = 0; i < kRangeSlots; ++i) m = std::max(m, e.h_range[(size_t)l * kRangeSlots + i]);
  float f;
  memcpy(&f, &m, sizeof(f));
  return f;
}
// The loud part of the f16 modes' contract: an operand that hit the clamp, or one so small that its lo halves
// carry nothing, fails the call instead of returning numbers outside the advertised accuracy.
int EvalRange(const pk_mi355_am *am, const ExecBufs *const *bufs, int nbufs) {
  const int nlin = (int)am->lin.size();
  for (int l = 0; l < nlin; ++l) {
    float m = 0.0f;
    for (int i = 0; i < nbufs; ++i)
      if (bufs[i]->h_range) m = std::max(m, RangeMax(*bufs[i], l));
    const int xe = am->exps_stale ? 0 : am->h_exps[nlin + l];
    const char *mode = am->precision == PK_MI355_PRECISION_F16 ? "f16" : "f16x3";
    if (m >= kRangeSaturated)
      return Fail(PK_MI355_E_RANGE, "%s range: the input of affine layer %d saturated (|x| * 2^%d reached 65504, the fp16 split "
                  "clamps there): results withheld; lower the operand's exponent (pk_mi355_am_calibrate)", mode, l, xe);
    if (m > 0.0f && m < kRangeTooSmall)
      return Fail(PK_MI355_E_RANGE, "%s range: the input of affine layer %d is too small for the fp16 split (max |x| * 2^%d = %g "
                  "< 2^-5: every low half is subnormal): results withheld; raise the operand's exponent "
                  "(pk_mi355_am_calibrate)", mode, l, xe, (double)m);
  }
  return 0;
}

// fp32 mode, stable softmax: the log-likelihood tail takes the wave-per-row arithmetic of pk_tail_wave.h everywhere --
// INSIDE the last affine layer's launch where that is a big-tile one (gemm.hip, TAIL variant: +2-3 % on the step,
// profiles/r04_fused_tail32.txt), as TailWaveKernel elsewhere -- so that a model's log-likelihoods do not depend on
// the size of the launch that made them.  PK_MI355_FUSED_TAIL32=0 brings TailKernel back (A/B runs, the equality test).
bool WaveTail32() {
  const char *c = getenv("PK_MI355_FUSED_TAIL32");
  return !(c && atoi(c) == 0);
}

// Run the layer stack on `rows` frames (rows_pad = multiple of 128, <= rows_cap).
// Input: either the spliced view of Yt (splice_dim > 0: q0 points at Yt + first
// column, ldq = ldy) or the plain feature-major panel in e.in (ld = rows_cap).
// tail: -1 none (probabilities / raw outputs stay in a buffer, see *res),
//       otherwise the log-likelihood tail is written to tail_out[row * tail_ld].
int RunLayers(const pk_mi355_am *am, const ExecBufs &e, const float *q0, int64_t ldq,
              int splice_dim, int rows, bool want_tail, float scale, float *tail_out,
              int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res) {
  const int rows_pad = (int)RoundUp(rows, kTile);
  if (rows_pad > e.rows_cap) return Fail(PK_MI355_E_INVALID, "chunk larger than workspace");
  const float *blob = am->d_blob;
  const int nl = (int)am->layers.size();

  int last_linear = -1;
  for (int i = 0; i < nl; ++i)
    if (am->layers[i].type == PK_NNET_LINEAR_LAYER) last_linear = i;

  // current activation: pointer, layout (feature-major panel / frame-major rows), ld, dim
  const float *cur = splice_dim > 0 ? q0 : e.in;
  int64_t cur_ld = splice_dim > 0 ? ldq : e.rows_cap;
  bool cur_rows = false;           // false: feature-major [dim][ld]; true: frame-major [rows][ld]
  bool cur_splice = splice_dim > 0;
  int cur_dim = am->input_dim;
  float *bufs[2] = {e.a, e.b};
  int next_buf = 0;
  int li_lin = 0;
  bool tail_done = false;

  auto to_rows = [&]() {
    float *dst = bufs[next_buf];
    const int64_t ld = RoundUp(cur_dim, kTile);
    Scoped t(timer, PK_MI355_K_OTHER, stream);
    LaunchTransposeToRows(cur, cur_ld, cur_dim, rows_pad, dst, ld, stream);
    cur = dst; cur_ld = ld; cur_rows = true; next_buf ^= 1;
  };

  for (int i = 0; i < nl; ++i) {
    const HostLayer &L = am->layers[i];
    switch (L.type) {
      case PK_NNET_LINEAR_LAYER: {
        const DevLinear &D = am->lin[li_lin++];
        if (cur_rows) return Fail(PK_MI355_E_INVALID, "linear layer after a softmax is not supported");
        const bool fuse_relu = (i + 1 < nl && am->layers[i + 1].type == PK_NNET_RELU_LAYER);
        const bool rows_out = (i == last_linear) && !cur_splice;
        float *dst = bufs[next_buf];
        GemmArgs g;
        g.K = D.Kpad;
        g.relu = fuse_relu ? 1 : 0;
        g.bias = blob + D.b_off;
        g.out = dst;
        if (!rows_out) {           // out[feature][frame]
          g.P = blob + D.wt_off; g.ldp = D.Npad;
          g.Q = cur; g.ldq = cur_ld;
          g.splice_dim = cur_splice ? splice_dim : 0;
          g.bias_on_j = 0;
          g.ldo = e.rows_cap;
          g.tiles_i = D.Npad / kTile; g.tiles_j = rows_pad / kTile;
        } else {                   // out[frame][feature]
          g.P = cur; g.ldp = cur_ld;
          g.Q = blob + D.wt_off; g.ldq = D.Npad;
          g.splice_dim = 0;
          g.bias_on_j = 1;
          g.ldo = D.Npad;
          g.tiles_i = rows_pad / kTile; g.tiles_j = D.Npad / kTile;
        }
        bool fused_tail = false;
        if (rows_out && want_tail && !fuse_relu && !am->softmax_reference && i + 2 == nl &&
            am->layers[i + 1].type == PK_NNET_SOFTMAX_LAYER && WaveTail32()) {
          g.tail_out = tail_out; g.tail_ld = tail_ld;
          g.tail_log_prior = blob + am->logprior_off; g.tail_scale = scale;
          g.tail_n = D.N; g.tail_rows = rows; g.row_done = e.row_done;
          fused_tail = GemmFusesTail(g);
          if (!fused_tail) g.tail_out = nullptr;
        }
        {
          Scoped t(timer, PK_MI355_K_GEMM, stream);
          LaunchGemm(g, stream);
        }
        cur = dst; cur_ld = g.ldo; cur_rows = rows_out; cur_splice = false; cur_dim = D.N;
        next_buf ^= 1;
        if (fuse_relu) ++i;
        if (fused_tail) { tail_done = true; ++i; }
        break;
      }
      case PK_NNET_RELU_LAYER: {
        if (cur_splice || cur == e.in) {   // never modify inputs in place: copy first
          float *dst = bufs[next_buf];
          if (cur_splice) return Fail(PK_MI355_E_INVALID, "network must start with a linear layer when splicing");
          HIP_TRY(hipMemcpyAsync(dst, cur, sizeof(float) * (size_t)RoundUp(cur_dim, kBK) * e.rows_cap,
                                 hipMemcpyDeviceToDevice, stream));
          cur = dst; next_buf ^= 1;
        }
        Scoped t(timer, PK_MI355_K_OTHER, stream);
        const int64_t n = cur_rows ? (int64_t)rows_pad * cur_ld : (int64_t)cur_dim * cur_ld;
        LaunchRelu(const_cast<float *>(cur), n, stream);
        break;
      }
      case PK_NNET_NORMALIZE_LAYER: {
        if (cur_splice) return Fail(PK_MI355_E_INVALID, "network must start with a linear layer when splicing");
        if (cur == e.in) {
          float *dst = bufs[next_buf];
          HIP_TRY(hipMemcpyAsync(dst, cur, sizeof(float) * (size_t)RoundUp(cur_dim, kBK) * e.rows_cap,
                                 hipMemcpyDeviceToDevice, stream));
          cur = dst; next_buf ^= 1;
        }
        Scoped t(timer, PK_MI355_K_OTHER, stream);
        if (cur_rows) LaunchNormalize(const_cast<float *>(cur), rows_pad, cur_dim, cur_ld, 1, stream);
        else LaunchNormalize(const_cast<float *>(cur), rows_pad, cur_dim, 1, cur_ld, stream);
        break;
      }
      case PK_NNET_SOFTMAX_LAYER: {
        if (cur_splice) return Fail(PK_MI355_E_INVALID, "network must start with a linear layer when splicing");
        if (!cur_rows) to_rows();
        const bool final_layer = (i == nl - 1);
        Scoped t(timer, PK_MI355_K_TAIL, stream);
        if (final_layer && want_tail) {
          if (!(WaveTail32() && !am->softmax_reference &&
                LaunchTailWave(cur, cur_ld, rows, cur_dim, blob + am->logprior_off, scale, tail_out, tail_ld, stream)))
            LaunchTail(kTailSoftmaxLoglik, am->softmax_reference, cur, cur_ld, rows, cur_dim,
                       blob + am->logprior_off, scale, tail_out, tail_ld, stream);
          tail_done = true;
        } else {
          float *dst = bufs[next_buf];
          const int64_t ld = RoundUp(cur_dim, kTile);
          LaunchTail(kTailSoftmaxProb, am->softmax_reference, cur, cur_ld, rows_pad, cur_dim, nullptr, 1.0f, dst, ld, stream);
          cur = dst; cur_ld = ld; next_buf ^= 1;
        }
        break;
      }
      default:
        return Fail(PK_MI355_E_INVALID, "unknown layer type %d", L.type);
    }
  }
  if (!cur_rows && !tail_done) {
    if (cur_splice) return Fail(PK_MI355_E_INVALID, "empty network");
    to_rows();
  }
  if (want_tail && !tail_done) {
    Scoped t(timer, PK_MI355_K_TAIL, stream);
    LaunchTail(kTailLoglik, am->softmax_reference, cur, cur_ld, rows, cur_dim, blob + am->logprior_off,
               scale, tail_out, tail_ld, stream);
  }
  if (res) { res->data = cur; res->ld = cur_ld; res->dim = cur_dim; }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return Fail(PK_MI355_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(le));
  return 0;
}

// f16x3 mode: the same layer walk on (hi, lo) fp16 pairs, everything frame-major.
// Input: interleaved rows x [rows][ldx halves] (ldx = 2 feat_dim for the spliced view of
// the CMVN output, row r = frames r .. r+L+R, am.cc:65-88; or the padded plain input).  Finalize has
// already checked the layer pattern: (Linear [ReLU])+ [Softmax].
int RunLayersF16(const pk_mi355_am *am, const ExecBufs &e, const _Float16 *x, int64_t ldx, int rows,
                 bool want_tail, float scale, float *tail_out,
                 int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res) {
  const int rows_pad = (int)RoundUp(rows, kTileF16);
  if (rows_pad > e.rows_cap) return Fail(PK_MI355_E_INVALID, "chunk larger than workspace");
  const float *blob = am->d_blob;
  const int nl = (int)am->layers.size();
  const int nlin = (int)am->lin.size();
  int li = 0, buf = 0;
  int64_t out_ld = 0;
  for (int i = 0; i < nl; ++i) {
    if (am->layers[i].type != PK_NNET_LINEAR_LAYER) continue;
    const DevLinear &D = am->lin[li];
    const bool last = (li == nlin - 1);
    GemmF16Args g;
    g.X = x; g.ldx = ldx;
    g.W = reinterpret_cast<const _Float16 *>(blob + D.wt_off);
    g.ldw = 2 * D.Kpad;
    g.K = D.Kpad;
    g.bias = blob + D.b_off;
    g.relu = (i + 1 < nl && am->layers[i + 1].type == PK_NNET_RELU_LAYER) ? 1 : 0;
    // a NormalizeLayer behind this layer (nnet.cc:62-75; never behind the last one, Finalize checked): the GEMM
    // leaves fp32 rows in `a` and NormalizeSplitKernel turns them into the next layer's (hi, lo) operand
    const int after = i + 1 + g.relu;
    const bool norm = after < nl && am->layers[after].type == PK_NNET_NORMALIZE_LAYER;
    g.out_f32 = (last || norm) ? e.a : nullptr;
    g.out = (last || norm) ? nullptr : e.h[buf];
    g.ldo = (last || norm) ? D.Npad : 2 * D.Npad;
    g.tiles_m = rows_pad / kTileF16;
    g.tiles_n = D.Npad / kTileF16;
    g.terms = am->precision == PK_MI355_PRECISION_F16 ? 1 : 3;
    g.e_w = ExpW(am, li);
    g.e_in = ExpX(am, li);
    g.e_out = (last || norm) ? ExpZero(am) : ExpX(am, li + 1);
    g.range = (last || norm) ? nullptr : RangeOf(e, li + 1);
    {
      Scoped t(timer, PK_MI355_K_GEMM, stream);
      LaunchGemmF16(g, stream);
    }
    if (norm) {
      Scoped t(timer, PK_MI355_K_OTHER, stream);
      LaunchNormalizeSplitF16(e.a, D.Npad, rows_pad, D.N, D.Npad, e.h[buf], 2 * D.Npad, ExpX(am, li + 1), RangeOf(e, li + 1), stream);
    }
    x = e.h[buf]; ldx = 2 * D.Npad;
    out_ld = D.Npad;
    buf ^= 1;
    ++li;
  }
  const int dim = am->output_dim;
  const bool softmax_last = am->layers.back().type == PK_NNET_SOFTMAX_LAYER;
  const float *cur = e.a;
  if (want_tail) {
    Scoped t(timer, PK_MI355_K_TAIL, stream);
    LaunchTail(softmax_last ? kTailSoftmaxLoglik : kTailLoglik, am->softmax_reference, cur, out_ld, rows, dim,
               blob + am->logprior_off, scale, tail_out, tail_ld, stream);
  } else if (softmax_last) {
    Scoped t(timer, PK_MI355_K_TAIL, stream);
    LaunchTail(kTailSoftmaxProb, am->softmax_reference, cur, out_ld, rows_pad, dim, nullptr, 1.0f, e.b, out_ld, stream);
    cur = e.b;
  }
  if (res) { res->data = cur; res->ld = out_ld; res->dim = dim; }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return Fail(PK_MI355_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(le));
  return 0;
}

}  // namespace

// Single-utterance workspace used by pk_decodable_init / nnet_propagate.
struct Workspace {
  ExecBufs exec;
  float *d_feats = nullptr;  int64_t feats_cap = 0;     // frame-major host upload
  float *d_yt = nullptr;     int64_t yt_ld = 0;          // [feat_dim][yt_ld]
  _Float16 *d_y2 = nullptr;                             // f16x3: interleaved rows [yt_ld][2 feat_dim]
  float *d_out = nullptr;    int64_t out_cap = 0;        // [rows][num_pdfs]
  hipStream_t stream = nullptr;
};

namespace {

constexpr int64_t kSingleChunk = 4096;   // frames per pass of the single-utterance path

int EnsureWorkspace(pk_mi355_am *am, int64_t frames, int width) {
  if (!am->ws) {
    am->ws = new Workspace();
    HIP_TRY(hipStreamCreate(&am->ws->stream));
    int rc = AllocExec(am, kSingleChunk, &am->ws->exec);
    if (rc) return rc;
  }
  Workspace *w = am->ws;
  const int64_t need_feats = frames * width;
  if (need_feats > w->feats_cap) {
    hipFree(w->d_feats);
    w->feats_cap = need_feats;
    HIP_TRY(hipMalloc(&w->d_feats, sizeof(float) * w->feats_cap));
  }
  const int64_t pad = am->left + am->right;
  const int64_t need_ld = RoundUp(frames + pad, kSingleChunk) + 256;
  if (am->feat_dim > 0 && need_ld > w->yt_ld) {
    hipFree(w->d_yt);
    w->yt_ld = need_ld;
    HIP_TRY(hipMalloc(&w->d_yt, sizeof(float) * w->yt_ld * am->feat_dim));
    HIP_TRY(hipMemset(w->d_yt, 0, sizeof(float) * w->yt_ld * am->feat_dim));
    if (IsF16(am->precision)) {
      hipFree(w->d_y2);
      HIP_TRY(hipMalloc(&w->d_y2, sizeof(_Float16) * 2 * w->yt_ld * am->feat_dim));
    }
  }
  const int64_t need_out = frames * std::max(am->output_dim, 1);
  if (need_out > w->out_cap) {
    hipFree(w->d_out);
    w->out_cap = need_out;
    HIP_TRY(hipMalloc(&w->d_out, sizeof(float) * w->out_cap));
  }
  return 0;
}

void FreeWorkspace(Workspace *w) {
  if (!w) return;
  FreeExec(&w->exec);
  hipFree(w->d_feats);
  hipFree(w->d_yt);
  hipFree(w->d_y2);
  hipFree(w->d_out);
  if (w->stream) hipStreamDestroy(w->stream);
  delete w;
}

int ResizeHostMatrix(pk_matrix_t *m, int nrow, int ncol) {
  const size_t total = (size_t)nrow * ncol;
  float *p = nullptr;
  if (total > 0) {
    p = static_cast<float *>(realloc(m->data, sizeof(float) * total));   // matrix.cc:84-97
    if (!p) return Fail(PK_MI355_E_INVALID, "out of host memory");
  } else {
    free(m->data);
  }
  m->data = p;
  m->nrow = nrow;
  m->ncol = ncol;
  return 0;
}

}  // namespace

// ================================================================== C ABI

// Page-locked host arenas of pk_mi355_batch_fetch_all.  A decodable handed out by fetch_all is a
// VIEW into its batch's arena; pk_decodable_destroy must not free() such a pointer, and the
// reference's caller destroys its decodable unconditionally, whenever it likes
// (pocketkaldi.cc:247) -- also after the batch is gone.  So the arena is shared property: the
// page-locked memory is released when the batch has been destroyed AND the views of its LAST
// fetch_all have been destroyed, whichever comes last.  (A caller that never destroys its views
// keeps the arena until the process ends.)
//
// Every fetch_all makes one ViewGen -- the generation its views belong to -- and a view's `am` field
// points at it (tagged, bit 0; opaque to every caller; pk_decodable_destroy and pk_decodable_loglikelihood resolve it).
// ViewGen records are 64-byte aligned and RECYCLED, never returned to the heap: bits 1-5 of a handle carry the
// record's serial, which moves on every time the record is retired, and retired records are reused oldest first --
// a bitwise copy of a view that has already been destroyed therefore names a (record, serial) pair that is not
// alive and is ignored, also after the record has been given to a later fetch_all (the guard used
// to key on the record's address alone).
// A view therefore always decrements the count of ITS OWN generation: a stale view of an earlier
// fetch_all, or of another batch whose arena once occupied the same addresses, can never drive the
// current generation's count to zero under views that are still outstanding (the
// counts used to be found by address range).  Whether a decodable is a view is not guessed from its
// address either, so pk_decodable_destroy free()s exactly the matrices malloc() made.
namespace {
std::mutex g_arena_mu;
struct ArenaRec;
struct alignas(64) ViewGen {
  pk_mi355_am_t *am;     // what Untag() resolves a view's handle to
  ArenaRec *arena;       // valid while `current`
  int live;              // views of this generation not yet destroyed
  bool current;          // the batch's latest fetch_all
  unsigned serial;       // 0..31, part of the handle
};
struct ArenaRec {
  void *mem;
  bool batch_alive;
  ViewGen *cur;
};
// live generations: a handle that is not in here (a view destroyed twice through a bitwise copy)
// is ignored instead of dereferenced
std::unordered_set<const ViewGen *> g_gens;
std::vector<ViewGen *> g_retired;     // FIFO of records waiting for reuse (head index below)
size_t g_retired_head = 0;

inline bool IsView(const pk_mi355_am_t *am) { return (reinterpret_cast<uintptr_t>(am) & 1u) != 0; }
inline ViewGen *GenOf(const pk_mi355_am_t *am) { return reinterpret_cast<ViewGen *>(reinterpret_cast<uintptr_t>(am) & ~uintptr_t(63)); }
inline unsigned SerialOf(const pk_mi355_am_t *am) { return (unsigned)((reinterpret_cast<uintptr_t>(am) >> 1) & 31u); }
inline pk_mi355_am_t *TagView(ViewGen *g) { return reinterpret_cast<pk_mi355_am_t *>(reinterpret_cast<uintptr_t>(g) | (uintptr_t(g->serial) << 1) | 1u); }
// under g_arena_mu
ViewGen *NewGenRecord(pk_mi355_am_t *am, ArenaRec *a, int views) {
  ViewGen *v;
  if (g_retired.size() - g_retired_head >= 64) {       // reuse only once 64 later records have been retired after it
    v = g_retired[g_retired_head++];
    if (g_retired_head > 4096) { g_retired.erase(g_retired.begin(), g_retired.begin() + g_retired_head); g_retired_head = 0; }
  } else {
    v = new ViewGen();
    v->serial = 0;
  }
  v->am = am; v->arena = a; v->live = views; v->current = true;
  g_gens.insert(v);
  return v;
}
void RetireGenRecord(ViewGen *v) {
  g_gens.erase(v);
  v->serial = (v->serial + 1) & 31u;
  g_retired.push_back(v);
}
// the model behind a decodable's handle (lock-free: a live view keeps its generation alive)
inline pk_mi355_am_t *Untag(pk_mi355_am_t *am) { return IsView(am) ? GenOf(am)->am : am; }

ArenaRec *RegisterArena(void *p) { return new ArenaRec{p, true, nullptr}; }
// a new fetch_all: its generation replaces the previous one, whose outstanding views are void by
// contract (they still own their generation record, nothing else)
pk_mi355_am_t *NewViewGen(ArenaRec *a, pk_mi355_am_t *am, int views) {
  std::lock_guard<std::mutex> g(g_arena_mu);
  if (a->cur) {
    a->cur->current = false;
    if (a->cur->live <= 0) RetireGenRecord(a->cur);
  }
  a->cur = NewGenRecord(am, a, views);
  return TagView(a->cur);
}
// The batch is going away: release the arena now, or leave that to the last view of its last fetch_all.
void RetireArena(ArenaRec *a) {
  void *release = nullptr;
  {
    std::lock_guard<std::mutex> g(g_arena_mu);
    a->batch_alive = false;
    if (!a->cur || a->cur->live <= 0) {
      if (a->cur) RetireGenRecord(a->cur);
      release = a->mem;
      delete a;
    }
  }
  if (release) hipHostFree(release);
}
// pk_decodable_destroy on a view: one view fewer in ITS generation; the last view of the current
// generation of a batch that is gone releases the arena.
void ReleaseArenaView(pk_mi355_am_t *handle) {
  void *release = nullptr;
  {
    std::lock_guard<std::mutex> g(g_arena_mu);
    ViewGen *v = GenOf(handle);
    if (!g_gens.count(v) || v->serial != SerialOf(handle)) return;   // generation already gone: a copy destroyed twice
    if (--v->live > 0) return;
    if (v->current) {
      ArenaRec *a = v->arena;
      if (a->batch_alive) return;                 // the batch releases it (RetireArena) or re-uses it
      release = a->mem;
      delete a;
    }
    RetireGenRecord(v);
  }
  if (release) hipHostFree(release);
}
// One device-to-host result stream per device (pk_mi355_batch_fetch_all); lives for the process.
hipStream_t ResultStream(int device) {
  static hipStream_t streams[64] = {};
  if (device < 0 || device >= 64) { Fail(PK_MI355_E_INVALID, "device index %d", device); return nullptr; }
  std::lock_guard<std::mutex> g(g_arena_mu);
  if (!streams[device]) {
    hipError_t e = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
    if (e != hipSuccess) { Fail(PK_MI355_E_DEVICE, "result stream: %s", hipGetErrorString(e)); return nullptr; }
  }
  return streams[device];
}
}  // namespace

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

int pk_mi355_am_read(pk_mi355_am_t *am, const char *nnet_path, const char *prior_path,
                     const char *tid2pdf_path, int left_context, int right_context,
                     int num_pdfs) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "model already finalized");
  FileBuf f;
  int rc = f.Open(nnet_path);
  if (rc) return rc;
  int32_t sec, num_layers;
  if (!f.Tag("NNT0") || !f.I32(&sec) || !f.I32(&num_layers) || sec != 4)
    return Fail(PK_MI355_E_IO, "NNT0 section expected in %s", nnet_path);
  for (int l = 0; l < num_layers; ++l) {
    int32_t type;
    if (!f.Tag("LAY0") || !f.I32(&sec) || !f.I32(&type))
      return Fail(PK_MI355_E_IO, "LAY0 section expected in %s", nnet_path);
    if (sec != 4)    // nnet.cc:94-101
      return Fail(PK_MI355_E_IO, "read_layer: section_size == 4 expected, but %d found (%s)", sec, nnet_path);
    if (type == PK_NNET_LINEAR_LAYER) {
      int32_t rows, cols;
      if (!f.Tag("MAT0") || !f.I32(&sec) || !f.I32(&rows) || !f.I32(&cols) || rows <= 0 || cols <= 0)
        return Fail(PK_MI355_E_IO, "MAT0 section expected in %s", nnet_path);
      std::vector<float> W((size_t)rows * cols), row, bias;
      for (int r = 0; r < rows; ++r) {
        if ((rc = f.Vec(&row))) return rc;
        if ((int)row.size() != cols)
          return Fail(PK_MI355_E_IO, "Matrix::Read: row dim %d expected, but %d found: %s", cols, (int)row.size(), nnet_path);
        memcpy(&W[(size_t)r * cols], row.data(), sizeof(float) * cols);
      }
      if ((rc = f.Vec(&bias))) return rc;
      if ((int)bias.size() != rows) return Fail(PK_MI355_E_IO, "bias dimension mismatch in %s", nnet_path);
      if ((rc = pk_mi355_am_add_linear(am, cols, rows, W.data(), bias.data()))) return rc;
    } else {
      if ((rc = pk_mi355_am_add_layer(am, type)))
        return Fail(PK_MI355_E_IO, "read_layer: unexpected layer type: %d (%s)", type, nnet_path);
    }
  }
  std::vector<float> prior;
  FileBuf pf;
  if ((rc = pf.Open(prior_path)) || (rc = pf.Vec(&prior))) return rc;
  std::vector<int32_t> tid;
  if (tid2pdf_path) {
    FileBuf tf;
    if ((rc = tf.Open(tid2pdf_path)) || (rc = tf.Vec(&tid))) return rc;
  }
  if ((int)prior.size() != num_pdfs)
    return Fail(PK_MI355_E_INVALID, "prior has %d entries, num_pdfs = %d", (int)prior.size(), num_pdfs);
  return pk_mi355_am_finalize(am, prior.data(), num_pdfs, left_context, right_context,
                              tid.empty() ? nullptr : tid.data(), (int)tid.size());
}

}  // extern "C"

// ---- pk_load's share of this path (pocketkaldi.cc:72-144): the key = value model file.
namespace {

std::string TrimWs(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

// configuration.cc:16-55: '#' comments and blank lines skipped, exactly one '=' per line,
// keys lower-cased, empty values rejected.
struct ConfigFile {
  std::string filename;
  std::vector<std::pair<std::string, std::string>> table;

  int Read(const char *path) {
    filename = path;
    FILE *f = fopen(path, "r");
    if (!f) return Fail(PK_MI355_E_IO, "cannot open %s", path);
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    size_t pos = 0;
    while (pos < text.size()) {
      size_t e = text.find('\n', pos);
      if (e == std::string::npos) e = text.size();
      const std::string line = TrimWs(text.substr(pos, e - pos));
      pos = e + 1;
      if (line.empty() || line[0] == '#') continue;
      const size_t eq = line.find('=');
      if (eq == std::string::npos || line.find('=', eq + 1) != std::string::npos)
        return Fail(PK_MI355_E_IO, "Unexpected line in %s: %s", path, line.c_str());
      std::string key = TrimWs(line.substr(0, eq));
      const std::string value = TrimWs(line.substr(eq + 1));
      for (auto &c : key) c = (char)tolower((unsigned char)c);
      if (value.empty()) return Fail(PK_MI355_E_IO, "Value cound not be empty: %s: %s", path, line.c_str());
      bool found = false;
      for (auto &kv : table)
        if (kv.first == key) { kv.second = value; found = true; }
      if (!found) table.emplace_back(key, value);
    }
    return 0;
  }
  const std::string *Find(const char *key) const {
    for (const auto &kv : table)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  // configuration.cc:57-72: relative paths are relative to the directory of the file
  int Path(const char *key, std::string *out) const {
    const std::string *v = Find(key);
    if (!v) return Fail(PK_MI355_E_IO, "Unable to find key '%s' in %s", key, filename.c_str());
    const size_t slash = filename.rfind('/');
    *out = ((*v)[0] == '/' || slash == std::string::npos) ? *v : filename.substr(0, slash + 1) + *v;
    return 0;
  }
  int Integer(const char *key, int *out) const {
    const std::string *v = Find(key);
    if (!v) return Fail(PK_MI355_E_IO, "Unable to find key '%s' in %s", key, filename.c_str());
    char *end = nullptr;
    const long x = strtol(v->c_str(), &end, 10);
    if (end == v->c_str()) return Fail(PK_MI355_E_IO, "key '%s' in %s is not an integer: %s", key, filename.c_str(), v->c_str());
    *out = (int)x;
    return 0;
  }
};

}  // namespace

extern "C" {

int pk_mi355_load(const char *config_path, int precision, pk_mi355_am_t **am_out, float *cmvn_stats41) {
  if (!config_path || !am_out || !cmvn_stats41) return Fail(PK_MI355_E_INVALID, "null argument");
  *am_out = nullptr;
  ConfigFile conf;
  int rc = conf.Read(config_path);
  if (rc) return rc;
  // CMVN global statistics, pocketkaldi.cc:101-116: VEC0 of 40 sums + the frame count
  std::string cmvn_path, nnet, prior, tid2pdf;
  if ((rc = conf.Path("cmvn_stats", &cmvn_path))) return rc;
  FileBuf cf;
  std::vector<float> stats;
  if ((rc = cf.Open(cmvn_path.c_str())) || (rc = cf.Vec(&stats))) return rc;
  if ((int)stats.size() != kNumBins + 1)
    return Fail(PK_MI355_E_IO, "cmvn_stats in %s has %d entries, %d expected", cmvn_path.c_str(), (int)stats.size(), kNumBins + 1);
  // AcousticModel::Read, am.cc:22-62 (a missing left_context is not an error there either)
  int left = 0, right = 0, num_pdfs = 0;
  if ((rc = conf.Path("nnet", &nnet)) || (rc = conf.Path("prior", &prior))) return rc;
  if (conf.Find("left_context") && (rc = conf.Integer("left_context", &left))) return rc;
  if ((rc = conf.Integer("right_context", &right)) || (rc = conf.Integer("num_pdfs", &num_pdfs)) ||
      (rc = conf.Path("tid2pdf", &tid2pdf)))
    return rc;
  if (left < 0 || right < 0) return Fail(PK_MI355_E_INVALID, "negative context in %s", config_path);
  pk_mi355_am_t *am = pk_mi355_am_create();
  if (!am) return PK_MI355_E_DEVICE;
  if ((rc = pk_mi355_am_set_precision(am, precision)) ||
      (rc = pk_mi355_am_read(am, nnet.c_str(), prior.c_str(), tid2pdf.c_str(), left, right, num_pdfs))) {
    pk_mi355_am_destroy(am);
    return rc;
  }
  memcpy(cmvn_stats41, stats.data(), sizeof(float) * (kNumBins + 1));
  *am_out = am;
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

// ---- the one collective of the path: weight-blob broadcast over the caller's RCCL communicator.
// RCCL's C API, bound at run time (rccl.h: ncclBroadcast, ncclGetErrorString; ncclUint8 = 1).
namespace {
typedef int (*NcclBroadcastFn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*NcclErrorStringFn)(int);
std::mutex g_rccl_mu;
NcclBroadcastFn g_nccl_broadcast = nullptr;
NcclErrorStringFn g_nccl_error_string = nullptr;

int BindRccl() {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_nccl_broadcast) return 0;
  // The communicator belongs to ONE copy of RCCL: the one the caller created it with, which is
  // therefore already loaded.  Bind to that copy and never load a second one behind the caller's
  // back (a foreign ncclComm_t handed to another copy is undefined behaviour):
  //   1. $PK_MI355_RCCL_LIB, when set, names the library explicitly (loaded if need be);
  //   2. a copy visible in the global scope (a host linked with -lrccl);
  //   3. a copy loaded RTLD_LOCAL (Python / torch): found by soname with RTLD_NOLOAD.
  void *h = nullptr;
  void *sym = nullptr;
  const char *path = getenv("PK_MI355_RCCL_LIB");
  if (path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return Fail(PK_MI355_E_DEVICE, "PK_MI355_RCCL_LIB=%s cannot be opened: %s", path, dlerror());
    sym = dlsym(h, "ncclBroadcast");
  } else {
    sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
    if (!sym) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (!h)
        return Fail(PK_MI355_E_DEVICE, "no RCCL is loaded in this process (the communicator's library must be; "
                                       "or name it in PK_MI355_RCCL_LIB)");
      sym = dlsym(h, "ncclBroadcast");
    }
  }
  if (!sym) return Fail(PK_MI355_E_DEVICE, "ncclBroadcast not found in RCCL");
  void *es = h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
  g_nccl_error_string = reinterpret_cast<NcclErrorStringFn>(es);
  g_nccl_broadcast = reinterpret_cast<NcclBroadcastFn>(sym);
  return 0;
}
}  // namespace

namespace {
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
}  // namespace

int pk_mi355_am_broadcast_from(pk_mi355_am_t *am, pk_mi355_am_t *src, void *rccl_comm, int root, void *stream) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  if (src && (!src->finalized || src->blob_floats != am->blob_floats || src->device != am->device))
    return Fail(PK_MI355_E_INVALID, "source model does not have the destination's blob layout / device");
  if (!rccl_comm) return Fail(PK_MI355_E_INVALID, "null RCCL communicator");
  if (root < 0) return Fail(PK_MI355_E_INVALID, "bad root rank %d", root);
  int rc = UseDevice(am->device);
  if (rc || (rc = BindRccl())) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipStream_t own = nullptr;
  if (!s) {
    HIP_TRY(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    s = own;
  }
  // receive buffer = this rank's blob (same size on every rank: the layout depends on the layer
  // structure only); send buffer = the same blob (in place) or, on the root, `src`'s
  const int nr = g_nccl_broadcast(src ? src->d_blob : am->d_blob, am->d_blob, am->blob_floats * sizeof(float),
                                  /*ncclUint8*/ 1, root, rccl_comm, s);
  if (nr != 0) {
    if (own) hipStreamDestroy(own);
    return Fail(PK_MI355_E_DEVICE, "ncclBroadcast failed: %s", g_nccl_error_string ? g_nccl_error_string(nr) : "?");
  }
  am->exps_stale = IsF16(am->precision);     // the blob's exponent words now are the root's
  if (own) {
    hipError_t e = hipStreamSynchronize(own);
    hipStreamDestroy(own);
    if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "broadcast stream: %s", hipGetErrorString(e));
    return RefreshExps(am);
  }
  return 0;
}

int pk_mi355_am_broadcast(pk_mi355_am_t *am, void *rccl_comm, int root, void *stream) {
  return pk_mi355_am_broadcast_from(am, nullptr, rccl_comm, root, stream);
}

int pk_mi355_nnet_propagate(pk_mi355_am_t *am, const pk_matrix_t *in, pk_matrix_t *out) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  if (!in || !out || in->nrow != am->input_dim)
    return Fail(PK_MI355_E_INVALID, "input has %d rows, the network expects %d", in ? in->nrow : -1, am->input_dim);
  int rc = UseDevice(am->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(am->mu);
  const int T = in->ncol, D = in->nrow;
  if ((rc = ResizeHostMatrix(out, am->output_dim, T))) return rc;
  if (T == 0) return 0;
  if ((rc = EnsureWorkspace(am, T, std::max(D, am->output_dim)))) return rc;
  Workspace *w = am->ws;
  HIP_TRY(hipMemcpyAsync(w->d_feats, in->data, sizeof(float) * (size_t)T * D, hipMemcpyHostToDevice, w->stream));
  for (int64_t r0 = 0; r0 < T; r0 += kSingleChunk) {
    const int rows = (int)std::min<int64_t>(kSingleChunk, T - r0);
    ExecResult res;
    if (IsF16(am->precision)) {
      const int kp = (int)RoundUp(D, kBKF16);
      if ((rc = BeginRange(w->exec, w->stream))) return rc;
      LaunchSplitF16(w->d_feats + r0 * D, D, 1, rows, D, kp, w->exec.xin, 2 * kp, ExpX(am, 0), RangeOf(w->exec, 0), w->stream);
      rc = RunLayersF16(am, w->exec, w->exec.xin, 2 * kp, rows, false, 1.0f, nullptr, 0, w->stream, nullptr, &res);
      if (!rc) rc = CollectRange(w->exec, w->stream);
    } else {
      LaunchTransposeToCols(w->d_feats + r0 * D, D, rows, D, w->exec.in, w->exec.rows_cap, w->stream);
      rc = RunLayers(am, w->exec, nullptr, 0, 0, rows, false, 1.0f, nullptr, 0, w->stream, nullptr, &res);
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(out->data + r0 * am->output_dim, sizeof(float) * am->output_dim, res.data,
                             sizeof(float) * res.ld, sizeof(float) * am->output_dim, rows,
                             hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (IsF16(am->precision)) {
      const ExecBufs *eb = &w->exec;
      if ((rc = EvalRange(am, &eb, 1))) { ResizeHostMatrix(out, 0, 0); return rc; }
    }
  }
  return 0;
}

}  // extern "C"

namespace {
// The device half of pk_decodable_init (decodable.cc:8-17 -> am.cc:90-115), queued on the model's
// single-utterance stream (am->mu held, workspace sized): features up, edge-padded transpose (am.cc:73-75),
// the layer stack chunk by chunk; with want_tail the log-likelihoods land in ws->d_out.  In the f16 modes the
// range words of the call are zeroed first and collected into the page-locked mirror last.
int ScoreSingleQueue(pk_mi355_am *am, const pk_matrix_t *feats, bool want_tail, float prob_scale) {
  Workspace *w = am->ws;
  const int T = feats->ncol, D = feats->nrow, N = am->num_pdfs;
  const bool f16 = IsF16(am->precision);
  int rc;
  HIP_TRY(hipMemcpyAsync(w->d_feats, feats->data, sizeof(float) * (size_t)T * D, hipMemcpyHostToDevice, w->stream));
  LaunchPadTranspose(w->d_feats, T, D, am->left, am->right, w->d_yt, w->yt_ld, 0, w->stream);
  if (f16) {
    if ((rc = BeginRange(w->exec, w->stream))) return rc;
    LaunchSplitF16(w->d_yt, 1, w->yt_ld, (int)w->yt_ld, D, D, w->d_y2, 2 * D, ExpX(am, 0), RangeOf(w->exec, 0), w->stream);
  }
  for (int64_t r0 = 0; r0 < T; r0 += kSingleChunk) {
    const int rows = (int)std::min<int64_t>(kSingleChunk, T - r0);
    rc = f16 ? RunLayersF16(am, w->exec, w->d_y2 + r0 * 2 * D, 2 * D, rows, want_tail, prob_scale,
                            w->d_out + r0 * N, N, w->stream, nullptr, nullptr)
             : RunLayers(am, w->exec, w->d_yt + r0, w->yt_ld, D, rows, want_tail, prob_scale,
                         w->d_out + r0 * N, N, w->stream, nullptr, nullptr);
    if (rc) return rc;
  }
  if (f16 && (rc = CollectRange(w->exec, w->stream))) return rc;
  return 0;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ decodable

void pk_decodable_init(pk_decodable_t *self, pk_mi355_am_t *am, float prob_scale,
                       const pk_matrix_t *feats) {
  self->log_prob.ncol = 0;
  self->log_prob.nrow = 0;
  self->log_prob.data = nullptr;
  self->am = am;
  if (!am || !am->finalized) { Fail(PK_MI355_E_STATE, "model not finalized"); return; }
  if (!feats || feats->nrow != am->feat_dim) {
    Fail(PK_MI355_E_INVALID, "features have %d rows, the model expects %d", feats ? feats->nrow : -1, am->feat_dim);
    return;
  }
  if (UseDevice(am->device)) return;
  const int T = feats->ncol, D = feats->nrow, N = am->num_pdfs;
  if (T <= 0) return;
  const bool f16 = IsF16(am->precision);
  if (f16 && D % 8 != 0) {
    // the interleaved (hi, lo) row of a frame is made of whole 8-k chunks: the spliced view
    // (row stride 2 D halves, gemm_f16.hip) exists only for such D
    Fail(PK_MI355_E_INVALID, "f16x3 / f16 precision needs a feature dimension that is a multiple of 8 (got %d)", D);
    return;
  }
  std::lock_guard<std::mutex> lock(am->mu);
  if (EnsureWorkspace(am, T, D)) return;
  Workspace *w = am->ws;
  auto dev_fail = [&](hipError_t e) { Fail(PK_MI355_E_DEVICE, "HIP failure in pk_decodable_init: %s", hipGetErrorString(e)); };
  // One copy behind the last kernel.  The 12 MB of a 10 s utterance cross the link in 0.22 ms (54 GB/s, into pageable
  // malloc() memory as fast as into page-locked memory) -- more than the network takes (0.17 ms); sending the
  // log-likelihoods in row blocks under the last layer's remaining blocks was built three ways and gains nothing
  // (tools/experiments/init_pipeline, profiles/r04_decodable_init_pipeline.txt).
  const size_t bytes = sizeof(float) * (size_t)T * N;
  float *host = static_cast<float *>(malloc(bytes));                              // util.cc:58-68 pk_alloc
  if (!host) { Fail(PK_MI355_E_INVALID, "out of host memory"); return; }
  int rc = ScoreSingleQueue(am, feats, true, prob_scale);
  hipError_t e = hipSuccess;
  if (!rc) e = hipMemcpyAsync(host, w->d_out, bytes, hipMemcpyDeviceToHost, w->stream);
  const hipError_t se = hipStreamSynchronize(w->stream);      // (drained whatever happened: `host` may be freed next)
  if (e == hipSuccess) e = se;
  if (!rc && e != hipSuccess) { dev_fail(e); rc = PK_MI355_E_DEVICE; }
  if (!rc && f16) {                 // an operand left the fp16 split's range: fail loudly, deliver nothing
    const ExecBufs *eb = &w->exec;
    rc = EvalRange(am, &eb, 1);
  }
  if (rc) { free(host); return; }
  self->log_prob.ncol = T;
  self->log_prob.nrow = N;
  self->log_prob.data = host;
}

void pk_decodable_destroy(pk_decodable_t *self) {
  // matrix.cc:123-128 frees; a decodable handed out by pk_mi355_batch_fetch_all is a view of
  // the batch's page-locked arena (tagged handle) and owns nothing.
  if (IsView(self->am)) ReleaseArenaView(self->am);
  else free(self->log_prob.data);
  self->log_prob.data = nullptr;
  self->log_prob.nrow = 0;
  self->log_prob.ncol = 0;
  self->am = nullptr;
}

float pk_decodable_loglikelihood(pk_decodable_t *self, int frame, int trans_id) {
  const int pdf = pk_mi355_am_transition_to_pdf(Untag(self->am), trans_id);
  return self->log_prob.data[(size_t)frame * self->log_prob.nrow + pdf];
}

bool pk_decodable_islastframe(pk_decodable_t *self, int frame) {
  return frame == self->log_prob.ncol - 1;
}

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

}  // extern "C"

namespace {
// One calibration decision from the range words of one pass (h_range of `e`, stream synchronised).  Operands are
// settled front to back: the first operand not yet settled gets the CANONICAL exponent for its measured maximum --
// the one that puts it into [2^3, 2^4): 4 096 x headroom below the clamp, 2^8 above the too-small threshold, and every
// element within 2^-6 of the maximum with a normal lo half (smaller ones err by <= 2^-25 absolute, 2^-28 of the
// maximum: below the dropped lo x lo term).  Higher placements buy no accuracy and can cost clock: normal lo halves toggle
// more bits in a power-limited mode; measured, the placement is free either way (profiles/r04_f16_calibration_target_ab.txt).  The exponent does not depend on
// where it started from, so a calibration depends on the network and the data only (a saturated
// operand, true maximum unknown, first comes down by 2^12 and is looked at again).  Later operands were computed
// from it, so the caller reruns before looking further.  An operand already settled in this calibration is left
// alone while it stays inside [2^1, 2^7) (the maxima move in the last bits when an earlier exponent changes).
// Returns 1 if an exponent changed, 0 if every operand is settled.
int CalibrateStep(pk_mi355_am *am, const ExecBufs &e, std::vector<char> *settled) {
  const int nlin = (int)am->lin.size();
  // PK_MI355_CALIB_TARGET_LOG2: measurement switch for the placement (default 3: the maximum lands in [2^3, 2^4))
  static const int target = [] { const char *t = getenv("PK_MI355_CALIB_TARGET_LOG2"); const int v = t ? atoi(t) : 3; return v >= 0 && v <= 12 ? v : 3; }();
  for (int l = 0; l < nlin; ++l) {
    const float m = RangeMax(e, l);
    int32_t &xe = am->h_exps[nlin + l];
    int want = xe;
    if (m >= kRangeSaturated) want = xe - 12;
    else if (m > 0.0f && (!(*settled)[l] || m < ldexpf(1.0f, target - 2) || m >= ldexpf(1.0f, target + 4))) { want = xe + target - ilogbf(m); (*settled)[l] = 1; }
    else (*settled)[l] = 1;                        // an all-zero operand carries nothing to place
    want = std::min(kMaxXExp, std::max(-kMaxXExp, want));
    if (want != xe) { xe = want; return 1; }
  }
  return 0;
}
}  // namespace

extern "C" {

int pk_mi355_am_calibrate(pk_mi355_am_t *am, const pk_matrix_t *feats) {
  if (!am || !am->finalized) return Fail(PK_MI355_E_STATE, "model not finalized");
  if (!IsF16(am->precision)) return 0;                       // nothing to calibrate: F32 carries every finite value
  if (!feats || feats->nrow != am->feat_dim || feats->ncol <= 0 || !feats->data)
    return Fail(PK_MI355_E_INVALID, "calibration features have %d rows, the model expects %d", feats ? feats->nrow : -1, am->feat_dim);
  if (feats->nrow % 8 != 0) return Fail(PK_MI355_E_INVALID, "f16x3 / f16 precision needs a feature dimension that is a multiple of 8 (got %d)", feats->nrow);
  int rc = UseDevice(am->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(am->mu);
  if ((rc = EnsureWorkspace(am, feats->ncol, feats->nrow))) return rc;
  if (am->exps_stale && (rc = RefreshExps(am))) return rc;
  Workspace *w = am->ws;
  const int max_passes = 6 * (int)am->lin.size() + 8;
  std::vector<char> settled(am->lin.size(), 0);
  for (int pass = 0; pass < max_passes; ++pass) {
    if ((rc = ScoreSingleQueue(am, feats, false, 1.0f))) return rc;
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (!CalibrateStep(am, w->exec, &settled)) {
      const ExecBufs *eb = &w->exec;               // settled: what this last pass wrote must be in range (an operand
      return EvalRange(am, &eb, 1);                // pinned at the exponent limit is reported, not accepted)
    }
    if ((rc = UploadExps(am))) return rc;
  }
  return Fail(PK_MI355_E_RANGE, "calibration did not settle in %d passes", max_passes);
}

}  // extern "C"

// ================================================================== front-end + batch

struct pk_mi355_batch {
  pk_mi355_am *am = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  Timer timer;
  FrontendTables *d_tables = nullptr;
  float *d_global = nullptr;
  CmvnTables *d_cmvn_tab = nullptr;
  int max_utts = 0;
  int64_t max_samples = 0, max_frames = 0, max_cols = 0;
  int64_t chunk = 262144;  // frames per pass through the layer stack (PK_MI355_CHUNK overrides).  Without the fused tail the
                           // size hardly matters (256 x 10 s: 65536 -> 25.92, 131072 -> 25.79, 262144 -> 25.78 ms per step);
                           // with it every last-layer launch ends in ~0.25 ms of tail phases on an emptying chip, so fewer,
                           // larger passes win: 65536 -> 25.45, 131072 -> 25.0, 262144 -> 24.75 (profiles/r04_fused_tail32.txt)
  // PCM
  float *d_wave = nullptr;          // owned float buffer
  int16_t *d_wave_i16 = nullptr;    // owned int16 buffer
  const float *wave_f32 = nullptr;  // what the kernels read (owned or external)
  const int16_t *wave_i16 = nullptr;
  // per-utterance placement
  int num_utts = 0;
  int max_T = 0;
  int64_t total_frames = 0, total_cols = 0;
  std::vector<int64_t> h_wave_off, h_raw_base, h_pad_base;
  std::vector<int32_t> h_T;
  int64_t *d_wave_off = nullptr, *d_raw_base = nullptr, *d_pad_base = nullptr;
  int32_t *d_T = nullptr;
  // stages
  float *d_raw_alloc = nullptr;
  float *d_raw = nullptr;   // [max_frames][40], kCmvnRawLead floats into d_raw_alloc
  float *d_yt = nullptr;    // [feat_dim][ldy]
  _Float16 *d_y2 = nullptr;   // f16x3: interleaved (hi, lo) rows [ldy][2 feat_dim]
  int64_t ldy = 0;
  float *d_ll = nullptr;    // [max_cols][num_pdfs]
  float *h_ll = nullptr;    // page-locked mirror of d_ll (pk_mi355_batch_fetch_all), made on first use
  ArenaRec *arena = nullptr;  // its shared-ownership record
  hipEvent_t ev_scored = nullptr, ev_fetched = nullptr;
  // Optional second lane for the layer stack (PK_MI355_LANES=2): odd chunks run on their own
  // stream and buffers, so the HBM-bound tail of one chunk overlaps the MFMA-bound GEMMs of the
  // next (+1.7 % f32, +2.8 % f16x3 on 256 x 10 s).  Off by default: two GEMMs then share the
  // chip, every launch takes twice as long, and per-launch durations stop being a roofline
  // measurement.
  int lanes = 1;
  hipStream_t stream2 = nullptr;
  ExecBufs exec2;
  hipEvent_t ev_front = nullptr, ev_lane2 = nullptr;
  ExecBufs exec;
  bool scored = false;
  // f16 modes: the range words of the last score call wait in the page-locked mirrors until the stream has
  // been synchronised; the verdict is then sticky until the next score (every delivering call repeats it)
  bool range_pending = false;
  int range_status = 0;
  char range_msg[512] = "";
};

namespace {

int SetLayout(pk_mi355_batch *b, const int *num_samples, int num_utts) {
  if (num_utts < 0 || num_utts > b->max_utts) return Fail(PK_MI355_E_INVALID, "too many utterances (%d > %d)", num_utts, b->max_utts);
  const int pad = b->am->left + b->am->right;
  int64_t woff = 0, raw = 0, col = 0;
  b->h_wave_off.resize(num_utts); b->h_raw_base.resize(num_utts);
  b->h_pad_base.resize(num_utts); b->h_T.resize(num_utts);
  b->max_T = 0;
  for (int u = 0; u < num_utts; ++u) {
    if (num_samples[u] < 0) return Fail(PK_MI355_E_INVALID, "negative sample count");
    const int T = pk_mi355_num_frames(num_samples[u]);
    b->h_wave_off[u] = woff; b->h_raw_base[u] = raw; b->h_pad_base[u] = col; b->h_T[u] = T;
    woff += num_samples[u];
    raw += T;
    col += T > 0 ? T + pad : 0;
    b->max_T = std::max(b->max_T, T);
  }
  if (woff > b->max_samples) return Fail(PK_MI355_E_INVALID, "too many samples (%lld > %lld)", (long long)woff, (long long)b->max_samples);
  if (raw > b->max_frames || RoundUp(col, kTile) > b->max_cols) return Fail(PK_MI355_E_INVALID, "frame capacity exceeded");
  b->num_utts = num_utts;
  b->total_frames = raw;
  b->total_cols = col;
  b->scored = false;
  if (num_utts == 0) return 0;
  HIP_TRY(hipMemcpyAsync(b->d_wave_off, b->h_wave_off.data(), sizeof(int64_t) * num_utts, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(b->d_raw_base, b->h_raw_base.data(), sizeof(int64_t) * num_utts, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(b->d_pad_base, b->h_pad_base.data(), sizeof(int64_t) * num_utts, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(b->d_T, b->h_T.data(), sizeof(int32_t) * num_utts, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));   // the host vectors may be reused right away
  return 0;
}

// After the batch's stream has been synchronised: evaluate (once) the range words of the last score call.
int BatchRangeStatus(pk_mi355_batch *b) {
  if (b->range_pending) {
    const ExecBufs *eb[2] = {&b->exec, &b->exec2};
    b->range_status = EvalRange(b->am, eb, b->lanes == 2 ? 2 : 1);
    if (b->range_status) snprintf(b->range_msg, sizeof(b->range_msg), "%s", g_err);
    b->range_pending = false;
  }
  if (b->range_status) return Fail(b->range_status, "%s", b->range_msg);
  return 0;
}

int64_t TotalSamples(const int *num_samples, int n) {
  int64_t s = 0;
  for (int i = 0; i < n; ++i) s += num_samples[i];
  return s;
}

}  // namespace

extern "C" {

int pk_mi355_num_frames(int num_samples) {      // fbank.cc:35-42
  if (num_samples < kFrameLength) return 0;
  return 1 + (num_samples - kFrameLength) / kFrameShift;
}

pk_mi355_batch_t *pk_mi355_batch_create(pk_mi355_am_t *am, const float *global_stats41,
                                        int max_utts, int64_t max_total_samples) {
  if (!am || !am->finalized) { Fail(PK_MI355_E_STATE, "model not finalized"); return nullptr; }
  if (am->feat_dim != kNumBins) { Fail(PK_MI355_E_INVALID, "the front-end produces %d-dim features, the model expects %d", kNumBins, am->feat_dim); return nullptr; }
  if (!global_stats41 || max_utts <= 0 || max_total_samples <= 0) { Fail(PK_MI355_E_INVALID, "bad batch capacity"); return nullptr; }
  if (UseDevice(am->device)) return nullptr;
  pk_mi355_batch *b = new pk_mi355_batch();
  b->am = am;
  b->device = am->device;
  b->max_utts = max_utts;
  b->max_samples = max_total_samples;
  if (const char *c = getenv("PK_MI355_CHUNK")) {
    long v = atol(c);
    if (v >= kTile) b->chunk = RoundUp(v, kTileF16);
  }
  const int pad = am->left + am->right;
  b->max_frames = max_total_samples / kFrameShift + max_utts;
  b->max_cols = RoundUp(b->max_frames + (int64_t)max_utts * pad, kTileF16);
  b->chunk = std::min<int64_t>(b->chunk, b->max_cols);
  b->ldy = RoundUp(b->max_cols, b->chunk) + 256;
  FrontendTables host;
  bool ok = BuildFrontendTables(&host) == 0;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && ok) { ok = false; Fail(PK_MI355_E_DEVICE, "batch_create: %s", hipGetErrorString(e)); } };
  if (!ok) Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  chk(hipStreamCreate(&b->stream));
  chk(hipMalloc(&b->d_tables, sizeof(FrontendTables)));
  if (ok) chk(hipMemcpy(b->d_tables, &host, sizeof(FrontendTables), hipMemcpyHostToDevice));
  chk(hipMalloc(&b->d_global, sizeof(float) * (kNumBins + 1)));
  if (ok) chk(hipMemcpy(b->d_global, global_stats41, sizeof(float) * (kNumBins + 1), hipMemcpyHostToDevice));
  CmvnTables ctab;
  BuildCmvnTables(global_stats41[kNumBins], &ctab);
  chk(hipMalloc(&b->d_cmvn_tab, sizeof(CmvnTables)));
  if (ok) chk(hipMemcpy(b->d_cmvn_tab, &ctab, sizeof(CmvnTables), hipMemcpyHostToDevice));
  chk(hipMalloc(&b->d_wave_off, sizeof(int64_t) * max_utts));
  chk(hipMalloc(&b->d_raw_base, sizeof(int64_t) * max_utts));
  chk(hipMalloc(&b->d_pad_base, sizeof(int64_t) * max_utts));
  chk(hipMalloc(&b->d_T, sizeof(int32_t) * max_utts));
  const size_t raw_floats = (size_t)b->max_frames * kNumBins + kCmvnRawLead + kCmvnRawSlack;
  chk(hipMalloc(&b->d_raw_alloc, sizeof(float) * raw_floats));
  if (ok) chk(hipMemset(b->d_raw_alloc, 0, sizeof(float) * raw_floats));
  if (ok) b->d_raw = b->d_raw_alloc + kCmvnRawLead;
  chk(hipMalloc(&b->d_yt, sizeof(float) * b->ldy * kNumBins));
  if (ok) chk(hipMemset(b->d_yt, 0, sizeof(float) * b->ldy * kNumBins));
  if (IsF16(am->precision)) {
    chk(hipMalloc(&b->d_y2, sizeof(_Float16) * 2 * b->ldy * kNumBins));
  }
  chk(hipMalloc(&b->d_ll, sizeof(float) * b->max_cols * am->num_pdfs));
  if (ok && AllocExec(am, b->chunk, &b->exec)) ok = false;
  if (const char *c = getenv("PK_MI355_LANES")) b->lanes = atoi(c) >= 2 ? 2 : 1;
  if (b->max_cols <= b->chunk) b->lanes = 1;           // a single chunk has nothing to overlap with
  if (b->lanes == 2) {
    chk(hipStreamCreate(&b->stream2));
    chk(hipEventCreateWithFlags(&b->ev_front, hipEventDisableTiming));
    chk(hipEventCreateWithFlags(&b->ev_lane2, hipEventDisableTiming));
    if (ok && AllocExec(am, b->chunk, &b->exec2)) ok = false;
  }
  if (!ok) { pk_mi355_batch_destroy(b); return nullptr; }
  return b;
}

void pk_mi355_batch_destroy(pk_mi355_batch_t *b) {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->stream) hipStreamSynchronize(b->stream);
  if (b->stream2) hipStreamSynchronize(b->stream2);
  FreeExec(&b->exec);
  FreeExec(&b->exec2);
  if (b->ev_front) hipEventDestroy(b->ev_front);
  if (b->ev_lane2) hipEventDestroy(b->ev_lane2);
  if (b->stream2) hipStreamDestroy(b->stream2);
  hipFree(b->d_tables); hipFree(b->d_global); hipFree(b->d_cmvn_tab);
  hipFree(b->d_wave); hipFree(b->d_wave_i16);
  hipFree(b->d_wave_off); hipFree(b->d_raw_base); hipFree(b->d_pad_base); hipFree(b->d_T);
  hipFree(b->d_raw_alloc); hipFree(b->d_yt); hipFree(b->d_y2); hipFree(b->d_ll);
  if (b->arena) RetireArena(b->arena);   // released now, or by the last outstanding view of the last fetch_all
  if (b->ev_scored) hipEventDestroy(b->ev_scored);
  if (b->ev_fetched) hipEventDestroy(b->ev_fetched);
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
}

int pk_mi355_batch_set_waves(pk_mi355_batch_t *b, const pk_vector_t *waves, int num_utts) {
  if (!b || !waves) return Fail(PK_MI355_E_INVALID, "null argument");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  std::vector<int> ns(num_utts);
  for (int u = 0; u < num_utts; ++u) ns[u] = waves[u].dim;
  if (TotalSamples(ns.data(), num_utts) > b->max_samples) return Fail(PK_MI355_E_INVALID, "too many samples");
  if (!b->d_wave) HIP_TRY(hipMalloc(&b->d_wave, sizeof(float) * b->max_samples));
  int64_t off = 0;
  for (int u = 0; u < num_utts; ++u) {
    if (ns[u] > 0)
      HIP_TRY(hipMemcpyAsync(b->d_wave + off, waves[u].data, sizeof(float) * ns[u], hipMemcpyHostToDevice, b->stream));
    off += ns[u];
  }
  b->wave_f32 = b->d_wave;
  b->wave_i16 = nullptr;
  return SetLayout(b, ns.data(), num_utts);
}

int pk_mi355_batch_set_waves_i16(pk_mi355_batch_t *b, const int16_t *samples, const int *num_samples,
                                 int num_utts) {
  if (!b || !samples || !num_samples) return Fail(PK_MI355_E_INVALID, "null argument");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  const int64_t total = TotalSamples(num_samples, num_utts);
  if (total > b->max_samples) return Fail(PK_MI355_E_INVALID, "too many samples");
  if (!b->d_wave_i16) HIP_TRY(hipMalloc(&b->d_wave_i16, sizeof(int16_t) * b->max_samples));
  if (total > 0)
    HIP_TRY(hipMemcpyAsync(b->d_wave_i16, samples, sizeof(int16_t) * total, hipMemcpyHostToDevice, b->stream));
  b->wave_i16 = b->d_wave_i16;
  b->wave_f32 = nullptr;
  return SetLayout(b, num_samples, num_utts);
}

int pk_mi355_batch_set_waves_device(pk_mi355_batch_t *b, const float *d_samples, const int *num_samples,
                                    int num_utts) {
  if (!b || !d_samples || !num_samples) return Fail(PK_MI355_E_INVALID, "null argument");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  b->wave_f32 = d_samples;
  b->wave_i16 = nullptr;
  return SetLayout(b, num_samples, num_utts);
}

int pk_mi355_batch_score(pk_mi355_batch_t *b, float prob_scale, int sync) {
  if (!b) return Fail(PK_MI355_E_INVALID, "null batch");
  if (!b->wave_f32 && !b->wave_i16) return Fail(PK_MI355_E_STATE, "no waves set");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  pk_mi355_am *am = b->am;
  Timer *tm = b->timer.enabled ? &b->timer : nullptr;
  if (tm) tm->Reset();
  if (b->num_utts == 0 || b->total_frames == 0) { b->scored = true; return 0; }
  UttLayout lay{b->d_wave_off, b->d_T, b->d_raw_base, b->d_pad_base};
  {
    Scoped t(tm, PK_MI355_K_FBANK, b->stream);
    LaunchFbank(b->wave_f32, b->wave_i16, lay, b->num_utts, b->max_T, b->d_tables, b->d_raw, b->stream);
  }
  {
    Scoped t(tm, PK_MI355_K_CMVN, b->stream);
    LaunchCmvn(b->d_raw, lay, b->num_utts, b->d_global, b->d_cmvn_tab, am->left, am->right, b->d_yt, b->ldy, b->stream);
  }
  // Rows of the spliced operand = padded columns; row r of utterance u (r in
  // [pad_base, pad_base + T)) is its frame r - pad_base.  The few rows that
  // straddle two utterances are computed and ignored.
  const int N = am->num_pdfs;
  const bool f16 = IsF16(am->precision);
  if (f16) {
    if ((rc = BeginRange(b->exec, b->stream))) return rc;
    Scoped t(tm, PK_MI355_K_OTHER, b->stream);
    LaunchSplitF16(b->d_yt, 1, b->ldy, (int)b->ldy, kNumBins, kNumBins, b->d_y2, 2 * kNumBins, ExpX(am, 0), RangeOf(b->exec, 0), b->stream);
  }
  const bool two = b->lanes == 2 && b->total_cols > b->chunk;
  if (two) {                                   // lane 2 starts when the features are ready
    HIP_TRY(hipEventRecord(b->ev_front, b->stream));
    HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_front, 0));
    if (f16 && (rc = BeginRange(b->exec2, b->stream2))) return rc;
  }
  int lane = 0;
  for (int64_t c0 = 0; c0 < b->total_cols; c0 += b->chunk, lane ^= 1) {
    const int rows = (int)std::min<int64_t>(b->chunk, b->total_cols - c0);
    hipStream_t s = (two && lane) ? b->stream2 : b->stream;
    const ExecBufs &e = (two && lane) ? b->exec2 : b->exec;
    rc = f16 ? RunLayersF16(am, e, b->d_y2 + c0 * 2 * kNumBins, 2 * kNumBins, rows, true,
                            prob_scale, b->d_ll + c0 * N, N, s, tm, nullptr)
             : RunLayers(am, e, b->d_yt + c0, b->ldy, kNumBins, rows, true, prob_scale,
                         b->d_ll + c0 * N, N, s, tm, nullptr);
    if (rc) return rc;
  }
  if (two) {                                   // everything is ordered on b->stream again
    if (f16 && (rc = CollectRange(b->exec2, b->stream2))) return rc;
    HIP_TRY(hipEventRecord(b->ev_lane2, b->stream2));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_lane2, 0));
  }
  if (f16) {
    if ((rc = CollectRange(b->exec, b->stream))) return rc;
    b->range_pending = true;
    b->range_status = 0;
  }
  b->scored = true;
  if (sync) return pk_mi355_batch_synchronize(b);
  return 0;
}

int pk_mi355_batch_synchronize(pk_mi355_batch_t *b) {
  if (!b) return Fail(PK_MI355_E_INVALID, "null batch");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return BatchRangeStatus(b);
}

int pk_mi355_batch_calibrate(pk_mi355_batch_t *b) {
  if (!b) return Fail(PK_MI355_E_INVALID, "null batch");
  pk_mi355_am *am = b->am;
  if (!IsF16(am->precision)) return 0;
  if (!b->wave_f32 && !b->wave_i16) return Fail(PK_MI355_E_STATE, "no waves set");
  if (b->total_frames == 0) return Fail(PK_MI355_E_INVALID, "calibration needs at least one frame");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(am->mu);
  if (am->exps_stale && (rc = RefreshExps(am))) return rc;
  const int max_passes = 6 * (int)am->lin.size() + 8;
  std::vector<char> settled(am->lin.size(), 0);
  for (int pass = 0; pass < max_passes; ++pass) {
    rc = pk_mi355_batch_score(b, 1.0f, 0);
    b->scored = false;                           // calibration passes are not results
    b->range_pending = false;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->lanes == 2)
      for (int i = 0; i < b->exec.range_words; ++i) b->exec.h_range[i] = std::max(b->exec.h_range[i], b->exec2.h_range[i]);
    if (!CalibrateStep(am, b->exec, &settled)) {
      const ExecBufs *eb = &b->exec;
      return EvalRange(am, &eb, 1);
    }
    if ((rc = UploadExps(am))) return rc;
  }
  return Fail(PK_MI355_E_RANGE, "calibration did not settle in %d passes", max_passes);
}

int pk_mi355_batch_num_utts(const pk_mi355_batch_t *b) { return b ? b->num_utts : 0; }
int pk_mi355_batch_num_frames(const pk_mi355_batch_t *b, int utt) {
  return (b && utt >= 0 && utt < b->num_utts) ? b->h_T[utt] : 0;
}
int64_t pk_mi355_batch_total_frames(const pk_mi355_batch_t *b) { return b ? b->total_frames : 0; }

const float *pk_mi355_batch_loglik_device(const pk_mi355_batch_t *b, int utt) {
  if (!b || utt < 0 || utt >= b->num_utts) return nullptr;
  return b->d_ll + b->h_pad_base[utt] * b->am->num_pdfs;
}

int pk_mi355_batch_fetch(pk_mi355_batch_t *b, int utt, pk_decodable_t *out) {
  if (!b || !out || utt < 0 || utt >= b->num_utts) return Fail(PK_MI355_E_INVALID, "bad utterance index");
  if (!b->scored) return Fail(PK_MI355_E_STATE, "batch not scored");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  const int T = b->h_T[utt], N = b->am->num_pdfs;
  out->am = b->am;
  out->log_prob.ncol = 0; out->log_prob.nrow = 0; out->log_prob.data = nullptr;
  if (T == 0) return 0;
  float *host = static_cast<float *>(malloc(sizeof(float) * (size_t)T * N));
  if (!host) return Fail(PK_MI355_E_INVALID, "out of host memory");
  hipError_t e = hipMemcpyAsync(host, pk_mi355_batch_loglik_device(b, utt), sizeof(float) * (size_t)T * N,
                                hipMemcpyDeviceToHost, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e != hipSuccess) { free(host); return Fail(PK_MI355_E_DEVICE, "fetch: %s", hipGetErrorString(e)); }
  if ((rc = BatchRangeStatus(b))) { free(host); return rc; }      // f16 modes: out-of-range results are withheld
  out->log_prob.ncol = T; out->log_prob.nrow = N; out->log_prob.data = host;
  return 0;
}

int pk_mi355_batch_fetch_all(pk_mi355_batch_t *b, pk_decodable_t *out, int num_out, int sync) {
  if (!b || !out || num_out != b->num_utts) return Fail(PK_MI355_E_INVALID, "fetch_all: expected %d decodables", b ? b->num_utts : 0);
  if (!b->scored) return Fail(PK_MI355_E_STATE, "batch not scored");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  const int N = b->am->num_pdfs;
  if (!b->h_ll) {
    const size_t bytes = sizeof(float) * (size_t)b->max_cols * N;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_ll), bytes, hipHostMallocDefault));
    b->arena = RegisterArena(b->h_ll);
  }
  // One transfer of the used part of d_ll (the 10 rows between two utterances travel too:
  // 1 % at T = 998, and one large copy runs at the link rate).
  // The transfer is queued on ONE result stream per device, ordered after this batch's scoring
  // and before anything later on the batch's stream.  Results of several batches in flight
  // then leave the device first come, first served at the full link rate -- on their own
  // streams the copies would share the link, finish together, and the batches would fall into
  // step (all scoring, then all copying) instead of overlapping.
  if (b->total_cols > 0) {
    hipStream_t rs = ResultStream(b->device);
    if (!rs) return PK_MI355_E_DEVICE;
    if (!b->ev_scored) {
      HIP_TRY(hipEventCreateWithFlags(&b->ev_scored, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&b->ev_fetched, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(b->ev_scored, b->stream));
    HIP_TRY(hipStreamWaitEvent(rs, b->ev_scored, 0));
    HIP_TRY(hipMemcpyAsync(b->h_ll, b->d_ll, sizeof(float) * (size_t)b->total_cols * N, hipMemcpyDeviceToHost, rs));
    HIP_TRY(hipEventRecord(b->ev_fetched, rs));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_fetched, 0));
  }
  pk_mi355_am_t *gen = NewViewGen(b->arena, b->am, num_out);   // the views of an earlier fetch_all are void by contract
  for (int u = 0; u < num_out; ++u) {
    const int T = b->h_T[u];
    out[u].am = gen;
    out[u].log_prob.ncol = T;
    out[u].log_prob.nrow = T > 0 ? N : 0;
    out[u].log_prob.data = T > 0 ? b->h_ll + (size_t)b->h_pad_base[u] * N : nullptr;
  }
  if (sync && (rc = pk_mi355_batch_synchronize(b))) {   // f16 modes: the range verdict of the score call comes with it
    for (int u = 0; u < num_out; ++u) {                 // nothing is delivered: hand back empty decodables, drop the views
      ReleaseArenaView(gen);
      out[u].am = b->am;
      out[u].log_prob.ncol = 0; out[u].log_prob.nrow = 0; out[u].log_prob.data = nullptr;
    }
    return rc;
  }
  return 0;
}

int pk_mi355_test_logf(const float *x, int n, float *out) {
  if (!x || !out || n < 0) return Fail(PK_MI355_E_INVALID, "bad argument");
  int rc = UseDevice(g_device);
  if (rc || n == 0) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  FrontendTables *d_tab = nullptr;
  float *d_x = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc(&d_tab, sizeof(host));
  if (e == hipSuccess) e = hipMalloc(&d_x, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&d_y, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMemcpy(d_tab, &host, sizeof(host), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x, x, sizeof(float) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    LaunchLogfTest(d_x, n, d_tab, d_y, nullptr);
    e = hipMemcpy(out, d_y, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
  }
  hipFree(d_tab); hipFree(d_x); hipFree(d_y);
  if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "test_logf: %s", hipGetErrorString(e));
  return 0;
}

int pk_mi355_test_srfft512(const float *frames, int num_frames, float *spectra) {
  if (!frames || !spectra || num_frames < 0) return Fail(PK_MI355_E_INVALID, "bad argument");
  int rc = UseDevice(g_device);
  if (rc || num_frames == 0) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  const size_t bytes = sizeof(float) * (size_t)num_frames * kFftSize;
  FrontendTables *d_tab = nullptr;
  float *d_x = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc(&d_tab, sizeof(host));
  if (e == hipSuccess) e = hipMalloc(&d_x, bytes);
  if (e == hipSuccess) e = hipMalloc(&d_y, bytes);
  if (e == hipSuccess) e = hipMemcpy(d_tab, &host, sizeof(host), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x, frames, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    LaunchSrfft512Test(d_x, num_frames, d_tab, d_y, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(spectra, d_y, bytes, hipMemcpyDeviceToHost);
  hipFree(d_tab); hipFree(d_x); hipFree(d_y);
  if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "test_srfft512: %s", hipGetErrorString(e));
  return 0;
}

int pk_mi355_batch_fetch_fbank(pk_mi355_batch_t *b, int utt, float *out) {
  if (!b || !out || utt < 0 || utt >= b->num_utts) return Fail(PK_MI355_E_INVALID, "bad utterance index");
  if (!b->scored) return Fail(PK_MI355_E_STATE, "batch not scored");
  const int T = b->h_T[utt];
  if (T == 0) return 0;
  HIP_TRY(hipMemcpyAsync(out, b->d_raw + b->h_raw_base[utt] * kNumBins, sizeof(float) * (size_t)T * kNumBins,
                         hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int pk_mi355_batch_fetch_cmvn(pk_mi355_batch_t *b, int utt, float *out) {
  if (!b || !out || utt < 0 || utt >= b->num_utts) return Fail(PK_MI355_E_INVALID, "bad utterance index");
  if (!b->scored) return Fail(PK_MI355_E_STATE, "batch not scored");
  const int T = b->h_T[utt];
  if (T == 0) return 0;
  std::vector<float> tmp((size_t)kNumBins * T);
  HIP_TRY(hipMemcpy2DAsync(tmp.data(), sizeof(float) * T, b->d_yt + b->h_pad_base[utt] + b->am->left,
                           sizeof(float) * b->ldy, sizeof(float) * T, kNumBins, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < kNumBins; ++d) out[(size_t)t * kNumBins + d] = tmp[(size_t)d * T + t];
  return 0;
}

int pk_mi355_batch_gather_loglik(pk_mi355_batch_t *b, int utt, const int32_t *d_frames,
                                 const int32_t *d_trans_ids, int n, float *d_out) {
  if (!b || utt < 0 || utt >= b->num_utts || !d_frames || !d_trans_ids || !d_out)
    return Fail(PK_MI355_E_INVALID, "bad gather arguments");
  if (!b->scored) return Fail(PK_MI355_E_STATE, "batch not scored");
  int rc = UseDevice(b->device);
  if (rc) return rc;
  LaunchGather(pk_mi355_batch_loglik_device(b, utt), b->am->num_pdfs, b->am->d_tid2pdf,
               (int)b->am->tid2pdf.size(), d_frames, d_trans_ids, n, d_out, b->stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "gather: %s", hipGetErrorString(e));
  return 0;
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

void *pk_mi355_batch_stream(pk_mi355_batch_t *b) { return b ? (void *)b->stream : nullptr; }

int pk_mi355_batch_enable_timing(pk_mi355_batch_t *b, int enable) {
  if (!b) return Fail(PK_MI355_E_INVALID, "null batch");
  b->timer.enabled = enable != 0;
  if (!enable) b->timer.Reset();
  return 0;
}

int pk_mi355_batch_get_timing(pk_mi355_batch_t *b, float ms[PK_MI355_K_COUNT], int launches[PK_MI355_K_COUNT]) {
  if (!b) return Fail(PK_MI355_E_INVALID, "null batch");
  for (int k = 0; k < PK_MI355_K_COUNT; ++k) { ms[k] = 0; launches[k] = 0; }
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (auto &r : b->timer.recs) {
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.kind] += t;
    launches[r.kind] += 1;
  }
  return 0;
}

// ------------------------------------------------------------------ single-utterance front-end

int pk_mi355_fbank_compute(const pk_vector_t *wave, pk_matrix_t *out) {
  if (!wave || !out) return Fail(PK_MI355_E_INVALID, "null argument");
  int rc = UseDevice(g_device);
  if (rc) return rc;
  const int T = pk_mi355_num_frames(wave->dim);
  if (T == 0) return ResizeHostMatrix(out, 0, 0);          // fbank.cc:272-273
  if ((rc = ResizeHostMatrix(out, kNumBins, T))) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  FrontendTables *d_tab = nullptr;
  float *d_wave = nullptr, *d_raw = nullptr;
  int64_t *d_i64 = nullptr;
  int32_t *d_T = nullptr;
  int64_t zeros[2] = {0, 0};
  int32_t hT = T;
  int ret = 0;
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
  step(hipMalloc(&d_tab, sizeof(FrontendTables)));
  step(hipMalloc(&d_wave, sizeof(float) * wave->dim));
  step(hipMalloc(&d_raw, sizeof(float) * (size_t)T * kNumBins));
  step(hipMalloc(&d_i64, sizeof(int64_t) * 2));
  step(hipMalloc(&d_T, sizeof(int32_t)));
  if (e == hipSuccess) {
    step(hipMemcpy(d_tab, &host, sizeof(FrontendTables), hipMemcpyHostToDevice));
    step(hipMemcpy(d_wave, wave->data, sizeof(float) * wave->dim, hipMemcpyHostToDevice));
    step(hipMemcpy(d_i64, zeros, sizeof(zeros), hipMemcpyHostToDevice));
    step(hipMemcpy(d_T, &hT, sizeof(hT), hipMemcpyHostToDevice));
  }
  if (e == hipSuccess) {
    UttLayout lay{d_i64, d_T, d_i64 + 1, d_i64 + 1};
    LaunchFbank(d_wave, nullptr, lay, 1, T, d_tab, d_raw, nullptr);
    step(hipGetLastError());
    step(hipMemcpy(out->data, d_raw, sizeof(float) * (size_t)T * kNumBins, hipMemcpyDeviceToHost));
  }
  if (e != hipSuccess) ret = Fail(PK_MI355_E_DEVICE, "fbank_compute: %s", hipGetErrorString(e));
  hipFree(d_tab); hipFree(d_wave); hipFree(d_raw); hipFree(d_i64); hipFree(d_T);
  return ret;
}

int pk_mi355_cmvn_apply(const pk_vector_t *global_stats, const pk_matrix_t *raw, pk_matrix_t *out) {
  if (!global_stats || !raw || !out) return Fail(PK_MI355_E_INVALID, "null argument");
  if (global_stats->dim != kNumBins + 1 || (raw->ncol > 0 && raw->nrow != kNumBins))
    return Fail(PK_MI355_E_INVALID, "cmvn expects 41 global stats and 40-dim features");
  int rc = UseDevice(g_device);
  if (rc) return rc;
  const int T = raw->ncol;
  if ((rc = ResizeHostMatrix(out, T > 0 ? kNumBins : 0, T))) return rc;
  if (T == 0) return 0;
  float *d_raw = nullptr, *d_g = nullptr, *d_yt = nullptr;
  CmvnTables *d_ctab = nullptr;
  CmvnTables ctab;
  BuildCmvnTables(global_stats->data[kNumBins], &ctab);
  int64_t *d_i64 = nullptr;
  int32_t *d_T = nullptr;
  int64_t zeros[2] = {0, 0};
  int32_t hT = T;
  const int64_t ld = RoundUp(T, 64);
  std::vector<float> tmp((size_t)kNumBins * T);
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
  float *d_raw_alloc = nullptr;
  const size_t raw_floats = (size_t)T * kNumBins + kCmvnRawLead + kCmvnRawSlack;
  step(hipMalloc(&d_raw_alloc, sizeof(float) * raw_floats));
  if (e == hipSuccess) step(hipMemset(d_raw_alloc, 0, sizeof(float) * raw_floats));
  d_raw = d_raw_alloc ? d_raw_alloc + kCmvnRawLead : nullptr;
  step(hipMalloc(&d_g, sizeof(float) * (kNumBins + 1)));
  step(hipMalloc(&d_ctab, sizeof(CmvnTables)));
  step(hipMalloc(&d_yt, sizeof(float) * ld * kNumBins));
  step(hipMalloc(&d_i64, sizeof(int64_t) * 2));
  step(hipMalloc(&d_T, sizeof(int32_t)));
  if (e == hipSuccess) {
    step(hipMemcpy(d_raw, raw->data, sizeof(float) * (size_t)T * kNumBins, hipMemcpyHostToDevice));
    step(hipMemcpy(d_g, global_stats->data, sizeof(float) * (kNumBins + 1), hipMemcpyHostToDevice));
    step(hipMemcpy(d_ctab, &ctab, sizeof(CmvnTables), hipMemcpyHostToDevice));
    step(hipMemcpy(d_i64, zeros, sizeof(zeros), hipMemcpyHostToDevice));
    step(hipMemcpy(d_T, &hT, sizeof(hT), hipMemcpyHostToDevice));
  }
  if (e == hipSuccess) {
    UttLayout lay{d_i64, d_T, d_i64, d_i64 + 1};
    LaunchCmvn(d_raw, lay, 1, d_g, d_ctab, 0, 0, d_yt, ld, nullptr);
    step(hipGetLastError());
    step(hipMemcpy2D(tmp.data(), sizeof(float) * T, d_yt, sizeof(float) * ld, sizeof(float) * T, kNumBins,
                     hipMemcpyDeviceToHost));
  }
  int ret = 0;
  if (e != hipSuccess) ret = Fail(PK_MI355_E_DEVICE, "cmvn_apply: %s", hipGetErrorString(e));
  else
    for (int t = 0; t < T; ++t)
      for (int d = 0; d < kNumBins; ++d) out->data[(size_t)t * kNumBins + d] = tmp[(size_t)d * T + t];
  hipFree(d_raw_alloc); hipFree(d_g); hipFree(d_ctab); hipFree(d_yt); hipFree(d_i64); hipFree(d_T);
  return ret;
}

// ------------------------------------------------------------------ pk_process, acoustic half

// pcm_reader.cc:45-220: strict 44-byte-header RIFF/WAVE PCM, mono, 16 kHz, 8/16/32-bit,
// sample values kept unscaled as float.
int pk_mi355_16kpcm_read(const char *filename, pk_vector_t *pcm_data) {
  if (!filename || !pcm_data) return Fail(PK_MI355_E_INVALID, "null argument");
  FileBuf f;
  int rc = f.Open(filename);
  if (rc) return rc;
  const unsigned char *b = f.d.data();
  const long size = (long)f.d.size();
  auto i32 = [&](long off) { int32_t v; memcpy(&v, b + off, 4); return v; };
  auto i16 = [&](long off) { int16_t v; memcpy(&v, b + off, 2); return (int)v; };
  if (size < 44) return Fail(PK_MI355_E_IO, "file too short for a WAVE header: %s", filename);
  if (memcmp(b, "RIFF", 4)) return Fail(PK_MI355_E_IO, "chunk_name == 'RIFF' expected: %s", filename);
  if (i32(4) != size - 8) return Fail(PK_MI355_E_IO, "chunk_size == %ld expected, but %d found: %s", size - 8, i32(4), filename);
  if (memcmp(b + 8, "WAVE", 4)) return Fail(PK_MI355_E_IO, "Format == 'WAVE' expected: %s", filename);
  if (memcmp(b + 12, "fmt ", 4)) return Fail(PK_MI355_E_IO, "subchunk1 == 'fmt ' expected: %s", filename);
  if (i32(16) != 16) return Fail(PK_MI355_E_IO, "subchunk1_size == 16 expected, but %d found: %s", i32(16), filename);
  if (i16(20) != 1) return Fail(PK_MI355_E_IO, "audio_format == 1 (PCM) expected, but %d found: %s", i16(20), filename);
  if (i16(22) != 1) return Fail(PK_MI355_E_IO, "num_channels == 1 (mono) expected, but %d found: %s", i16(22), filename);
  const int rate = i32(24);
  if (rate != kSampleRate) return Fail(PK_MI355_E_IO, "sample_rate == 16000 expected, but %d found: %s", rate, filename);
  const int byte_rate = i32(28), align = i16(32), bits = i16(34);
  if (bits != 8 && bits != 16 && bits != 32)
    return Fail(PK_MI355_E_IO, "bits_per_sample == 8, 16 or 32 expected, but %d found: %s", bits, filename);
  if (byte_rate != rate * bits / 8) return Fail(PK_MI355_E_IO, "bytes_rate == %d expected, but %d found: %s", rate * bits / 8, byte_rate, filename);
  if (align != bits / 8) return Fail(PK_MI355_E_IO, "block_align == %d expected, but %d found: %s", bits / 8, align, filename);
  if (memcmp(b + 36, "data", 4)) return Fail(PK_MI355_E_IO, "subchunk2 == 'data' expected: %s", filename);
  if (i32(40) != size - 44) return Fail(PK_MI355_E_IO, "subchunk2_size == %ld expected, but %d found: %s", size - 44, i32(40), filename);
  const int n = (int)((size - 44) / (bits / 8));
  float *s = static_cast<float *>(realloc(pcm_data->data, sizeof(float) * (n > 0 ? n : 1)));
  if (!s) return Fail(PK_MI355_E_INVALID, "out of host memory");
  const unsigned char *p = b + 44;
  for (int i = 0; i < n; ++i) {
    if (bits == 8) { s[i] = (float)(int8_t)p[0]; p += 1; }
    else if (bits == 16) { int16_t v; memcpy(&v, p, 2); s[i] = (float)v; p += 2; }
    else { int32_t v; memcpy(&v, p, 4); s[i] = (float)v; p += 4; }
  }
  pcm_data->data = s;
  pcm_data->dim = n;
  return 0;
}

// The three acoustic stages of pk_process (pocketkaldi.cc:186-218) fused on the device:
// wave -> fbank -> CMVN -> nnet -> decodable.  With verbose != 0 the reference's stage
// lines go to stderr ("Fbank: ..ms", "CMVN: ..ms", "NNET: ..ms"), timed with HIP events.
int pk_mi355_process_acoustic(pk_mi355_am_t *am, const pk_vector_t *cmvn_global_stats,
                              const pk_vector_t *raw_wave, float prob_scale, pk_decodable_t *out,
                              int verbose) {
  if (!am || !cmvn_global_stats || !raw_wave || !out) return Fail(PK_MI355_E_INVALID, "null argument");
  if (cmvn_global_stats->dim != kNumBins + 1) return Fail(PK_MI355_E_INVALID, "cmvn_global_stats must have 41 entries");
  out->am = am;
  out->log_prob.ncol = 0; out->log_prob.nrow = 0; out->log_prob.data = nullptr;
  if (raw_wave->dim == 0) return 0;                           // pocketkaldi.cc:180-184
  std::lock_guard<std::mutex> lock(am->mu);
  const int64_t need = std::max<int64_t>(raw_wave->dim, 16000);
  if (!am->proc || need > am->proc_cap ||
      memcmp(am->proc_stats, cmvn_global_stats->data, sizeof(am->proc_stats)) != 0) {
    if (am->proc) pk_mi355_batch_destroy(am->proc);
    am->proc_cap = std::max<int64_t>(need, 2 * am->proc_cap);
    am->proc = pk_mi355_batch_create(am, cmvn_global_stats->data, 1, am->proc_cap);
    if (!am->proc) { am->proc_cap = 0; return PK_MI355_E_DEVICE; }
    memcpy(am->proc_stats, cmvn_global_stats->data, sizeof(am->proc_stats));
  }
  pk_mi355_batch *b = am->proc;
  int rc = pk_mi355_batch_set_waves(b, raw_wave, 1);
  if (rc) return rc;
  pk_mi355_batch_enable_timing(b, verbose);
  if ((rc = pk_mi355_batch_score(b, prob_scale, 1))) return rc;
  if (verbose) {
    float ms[PK_MI355_K_COUNT];
    int launches[PK_MI355_K_COUNT];
    if ((rc = pk_mi355_batch_get_timing(b, ms, launches))) return rc;
    fprintf(stderr, "Fbank: %lfms\n", (double)ms[PK_MI355_K_FBANK]);
    fprintf(stderr, "CMVN: %lfms\n", (double)ms[PK_MI355_K_CMVN]);
    fprintf(stderr, "NNET: %lfms\n", (double)(ms[PK_MI355_K_GEMM] + ms[PK_MI355_K_TAIL] + ms[PK_MI355_K_OTHER]));
  }
  return pk_mi355_batch_fetch(b, 0, out);
}

}  // extern "C"
