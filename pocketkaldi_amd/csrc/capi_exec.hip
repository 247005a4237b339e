// capi_exec.hip -- the layer executor (fp32 and split-fp16 walks over the layer stack), the range words of the
// f16 modes, the single-utterance workspace, and the reference-compatible pk_decodable_* functions
// (decodable.h:20-41).  Host C++ over the HIP runtime; no CPU compute path.
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

namespace pkhost {

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
    e->row_done_bytes = nd;
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

int BeginRange(const ExecBufs &e, hipStream_t s) {
  if (e.range) HIP_TRY(hipMemsetAsync(e.range, 0, sizeof(uint32_t) * e.range_words, s));
  return 0;
}
int CollectRange(const ExecBufs &e, hipStream_t s) {
  if (e.range) HIP_TRY(hipMemcpyAsync(e.h_range, e.range, sizeof(uint32_t) * e.range_words, hipMemcpyDeviceToHost, s));
  return 0;
}
void ClearHostRange(const ExecBufs &e) {
  if (e.h_range) memset(e.h_range, 0, sizeof(uint32_t) * e.range_words);
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

// Input: either the spliced view of Yt (splice_dim > 0: q0 points at Yt + first
// column, ldq = ldy) or the plain feature-major panel in e.in (ld = rows_cap).
// tail: -1 none (probabilities / raw outputs stay in a buffer, see *res),
//       otherwise the log-likelihood tail is written to tail_out[row * tail_ld].
int RunLayers(const pk_mi355_am *am, const ExecBufs &e, const float *q0, int64_t ldq,
              int splice_dim, int rows, bool want_tail, float scale, float *tail_out,
              int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res, const float *splice_zero,
              const int32_t *splice_shift) {
  const int rows_pad = (int)RoundUp(rows, kTile);
  if (splice_dim > 0 && !splice_zero) return Fail(PK_MI355_E_INVALID, "spliced input without a zero source");
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
          if (cur_splice) {
            g.splice_ctx = am->left + am->right + 1;
            g.splice_zero = splice_zero;
            g.splice_shift = splice_shift;
            g.ring = am->knobs.l1_ring;
          }
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
            am->layers[i + 1].type == PK_NNET_SOFTMAX_LAYER && am->knobs.wave_tail32) {
          g.tail_out = tail_out; g.tail_ld = tail_ld;
          g.tail_log_prior = blob + am->logprior_off; g.tail_scale = scale;
          g.tail_n = D.N; g.tail_rows = rows; g.row_done = e.row_done;
          g.tail_walk = am->knobs.tail_walk;
          fused_tail = GemmFusesTail(g, am->knobs.fused_tail_min_tiles);
          if (!fused_tail) g.tail_out = nullptr;
          // the arrival counters start from zero whatever an earlier launch left behind (one that was cut short
          // must not make this launch's row tiles look complete early): 4 bytes per 128-row tile
          else HIP_TRY(hipMemsetAsync(e.row_done, 0, sizeof(unsigned) * (size_t)g.tiles_i, stream));
        }
        {
          Scoped t(timer, PK_MI355_K_GEMM, stream);
          // The ragged end of a wide last layer (3 000 pdfs: the 24th column tile holds 56 columns, 8 000: the 63rd holds
          // 64): when at most 64 columns of the last 128-wide tile are real, they go through a launch of 64 x 64 tiles of
          // their own, ahead of the big-tile launch, which then walks one column tile fewer -- half a column tile of
          // matrix work per row tile less.  Same per-element arithmetic (the k order does not depend on the tile).
          // (only while the big-tile launch stays one: with a column tile fewer it must still be a fused-tail launch)
          GemmArgs fewer = g;
          fewer.tiles_j -= 1;
          if (fused_tail && am->knobs.tail_strip && g.tiles_j > 8 && D.N <= (g.tiles_j - 1) * kTile + kTile / 2 &&
              GemmFusesTail(fewer, am->knobs.fused_tail_min_tiles)) {
            GemmArgs s = g;
            const int j0 = (g.tiles_j - 1) * kTile;
            s.Q = g.Q + j0; s.bias = g.bias + j0; s.out = g.out + j0;
            s.half_j = 1;
            s.tail_out = nullptr;
            LaunchGemm(s, stream);
            g.tiles_j -= 1;
          }
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
          if (!(am->knobs.wave_tail32 && !am->softmax_reference &&
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
                 int64_t tail_ld, hipStream_t stream, Timer *timer, ExecResult *res, const int32_t *row_shift4) {
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
    g.row_shift4 = li == 0 ? row_shift4 : nullptr;       // (the spliced view of the features only)
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

// Single-utterance workspace used by pk_decodable_init / nnet_propagate.
struct Workspace {
  ExecBufs exec;
  float *d_feats = nullptr;  int64_t feats_cap = 0;     // frame-major host upload
  float *d_yt = nullptr;     int64_t yt_ld = 0;          // [feat_dim][yt_ld]
  _Float16 *d_y2 = nullptr;                             // f16x3: interleaved rows [yt_ld][2 feat_dim]
  float *d_out = nullptr;    int64_t out_cap = 0;        // [rows][num_pdfs]
  hipStream_t stream = nullptr;
};

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

}  // namespace pkhost

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
                         w->d_out + r0 * N, N, w->stream, nullptr, nullptr, w->d_yt + (w->yt_ld - 256));
    if (rc) return rc;
  }
  if (f16 && (rc = CollectRange(w->exec, w->stream))) return rc;
  return 0;
}
}  // namespace

extern "C" {

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
  if (IsView(self->am) && GenOf(self->am)->withheld) return NAN;     // results the range check withheld (see fetch_all, sync == 0)
  const int pdf = pk_mi355_am_transition_to_pdf(Untag(self->am), trans_id);
  return self->log_prob.data[(size_t)frame * self->log_prob.nrow + pdf];
}

bool pk_decodable_islastframe(pk_decodable_t *self, int frame) {
  return frame == self->log_prob.ncol - 1;
}

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
