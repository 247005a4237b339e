// pk_host.h -- what the host-side translation units of libpk_mi355.so share (internal):
//   capi_model.hip       errors / device selection, model construction, operand exponents
//   capi_exec.hip        the layer executor, the single-utterance workspace, pk_decodable_*
//   capi_batch.hip       the batched device-resident scorer, result arenas and views
//   capi_io.hip          model / config / WAV files, the single-utterance front-end entries, test hooks
//   capi_collective.hip  the one collective: weight-blob broadcast over the caller's RCCL communicator
// Nothing here is part of the ABI (include/pk_mi355.h is); the library exports the C entries only
// (libpk_mi355.map).
#ifndef PK_HOST_H_
#define PK_HOST_H_

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/pk_mi355.h"
#include "pk_kernels.h"
#include "pk_tables.h"

namespace pkhost {

using namespace pkmi;

// ------------------------------------------------------------------ errors, device selection (capi_model.hip)

// formats the thread's error text (pk_mi355_last_error) and returns `code`
int Fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *LastError();
// Like hipSetDevice, the selected device is a per-thread setting (a worker thread that never
// called pk_mi355_set_device creates its objects on device 0).
int CurrentDevice();
int UseDevice(int device);

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return ::pkhost::Fail(PK_MI355_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                       \
  } while (0)

inline int64_t RoundUp(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// the two precisions that run on the fp16 matrix cores share layouts and code paths
inline bool IsF16(int precision) { return precision == PK_MI355_PRECISION_F16X3 || precision == PK_MI355_PRECISION_F16; }

// ------------------------------------------------------------------ environment switches
// Every PK_MI355_* switch is an A/B or test switch, never needed in production.  They are read when an
// OBJECT is made (a model, a batch scorer) and kept in it: no launch path calls getenv.
struct ModelKnobs {
  bool wave_tail32 = true;       // PK_MI355_FUSED_TAIL32 (0: the workgroup-per-row TailKernel instead of the wave tail)
  int fused_tail_min_tiles = 384;  // PK_MI355_FUSED_TAIL_MIN_TILES (tests raise it to force the stand-alone wave tail)
  int tail_walk = 0;             // PK_MI355_TAIL_WALK: super-tile columns of the fused-tail launch's walk (0: the whole row of tiles)
  int l1_ring = 2;               // PK_MI355_L1_RING: LDS slabs of the spliced first layer's big-tile launch (2: four workgroups per CU; 3: three)
  bool tail_strip = true;        // PK_MI355_TAIL_STRIP (0: the last column tile of the fused-tail launch stays a full 128-wide tile)
};
ModelKnobs ReadModelKnobs();

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

struct Workspace;   // the single-utterance workspace (capi_exec.hip)

}  // namespace pkhost

struct pk_mi355_am {
  int device = 0;
  std::vector<pkhost::HostLayer> layers;
  bool finalized = false;
  int precision = PK_MI355_PRECISION_F32;
  bool softmax_reference = false;  // PK_MI355_SOFTMAX_REFERENCE: the reference's softmax operations one by one
  int left = 0, right = 0, num_pdfs = 0;
  int input_dim = 0, output_dim = 0, feat_dim = 0;
  int max_dim_pad = 0;             // widest activation, rounded to the tile
  std::vector<int32_t> tid2pdf;
  int32_t *d_tid2pdf = nullptr;    // device copy for the on-GPU gather
  std::vector<pkhost::DevLinear> lin;      // one per linear layer, in order
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
  pkhost::ModelKnobs knobs;        // the environment switches as they stood when the model was created
  // The reference's pk_decodable_init / AcousticModel::Compute allocate per call and are re-entrant
  // for a shared model (nnet.cc:149-163 is const); here the single-utterance entry points share
  // one device workspace per model, so they serialise on this mutex instead.
  std::mutex mu;
  pkhost::Workspace *ws = nullptr; // single-utterance workspace of pk_decodable_init (under mu)
  struct pk_mi355_batch *proc = nullptr;   // cached 1-utterance scorer of pk_mi355_process_acoustic (under mu)
  int64_t proc_cap = 0;
  float proc_stats[41] = {0};
};

namespace pkhost {

// host mirror of the exponent words <- device (after a broadcast, or a write through the blob's device pointer); and back
int RefreshExps(pk_mi355_am *am);
int UploadExps(pk_mi355_am *am);

// ------------------------------------------------------------------ layer executor (capi_exec.hip)

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
  unsigned *row_done = nullptr;   // fused tail (gemm.hip, TAIL variant): one arrival counter per 128-row tile,
  size_t row_done_bytes = 0;      // zeroed in front of every fused-tail launch
};

int AllocExec(const pk_mi355_am *am, int64_t rows_cap, ExecBufs *e);
void FreeExec(ExecBufs *e);

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

int BeginRange(const ExecBufs &e, hipStream_t s);
int CollectRange(const ExecBufs &e, hipStream_t s);
void ClearHostRange(const ExecBufs &e);          // a lane that took no part in a call must not contribute stale maxima
float RangeMax(const ExecBufs &e, int l);
int EvalRange(const pk_mi355_am *am, const ExecBufs *const *bufs, int nbufs);
int CalibrateStep(pk_mi355_am *am, const ExecBufs &e, std::vector<char> *settled);

// splice_zero (spliced input only): zero floats in the same allocation as q0 (128 + the largest column shift of them) --
// the tail of feature row 0 of every Yt this library allocates is never written.  splice_shift: GemmArgs::splice_shift
int RunLayers(const pk_mi355_am *am, const ExecBufs &e, const float *q0, int64_t ldq,
              int splice_dim, int rows, bool want_tail, float scale, float *tail_out,
              int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res, const float *splice_zero = nullptr,
              const int32_t *splice_shift = nullptr);
int RunLayersF16(const pk_mi355_am *am, const ExecBufs &e, const _Float16 *x, int64_t ldx, int rows,
                 bool want_tail, float scale, float *tail_out,
                 int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res, const int32_t *row_shift4 = nullptr);

void FreeWorkspace(Workspace *w);
int ResizeHostMatrix(pk_matrix_t *m, int nrow, int ncol);

// ------------------------------------------------------------------ result arenas and views (capi_batch.hip)

struct ArenaRec;
struct alignas(64) ViewGen {
  pk_mi355_am_t *am;     // what Untag() resolves a view's handle to
  ArenaRec *arena;       // valid while `current`
  int live;              // views of this generation not yet destroyed
  bool current;          // the batch's latest fetch_all
  bool withheld;         // f16 modes: the views were handed out (sync == 0) before the score call's range verdict, and the
                         // verdict was PK_MI355_E_RANGE: pk_decodable_loglikelihood on them returns NaN
  unsigned serial;       // 0..31, part of the handle
};
inline bool IsView(const pk_mi355_am_t *am) { return (reinterpret_cast<uintptr_t>(am) & 1u) != 0; }
inline ViewGen *GenOf(const pk_mi355_am_t *am) { return reinterpret_cast<ViewGen *>(reinterpret_cast<uintptr_t>(am) & ~uintptr_t(63)); }
// the model behind a decodable's handle (lock-free: a live view keeps its generation alive)
inline pk_mi355_am_t *Untag(pk_mi355_am_t *am) { return IsView(am) ? GenOf(am)->am : am; }
void ReleaseArenaView(pk_mi355_am_t *handle);

}  // namespace pkhost

#endif  // PK_HOST_H_
