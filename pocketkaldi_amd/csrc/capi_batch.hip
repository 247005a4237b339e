// capi_batch.hip -- the batched, device-resident scorer (PCM in HBM -> fbank -> CMVN -> layer stack ->
// log-likelihoods in HBM), the page-locked result arenas and the decodable views handed out over them,
// and the acoustic half of pk_process (pocketkaldi.cc:186-218).  Host C++ over the HIP runtime.
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
namespace pkhost {
struct ArenaRec {
  void *mem;
  bool batch_alive;
  ViewGen *cur;
};
}  // namespace pkhost

namespace {
std::mutex g_arena_mu;
// live generations: a handle that is not in here (a view destroyed twice through a bitwise copy)
// is ignored instead of dereferenced
std::unordered_set<const ViewGen *> g_gens;
std::vector<ViewGen *> g_retired;     // FIFO of records waiting for reuse (head index below)
size_t g_retired_head = 0;

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
  v->am = am; v->arena = a; v->live = views; v->current = true; v->withheld = false;
  g_gens.insert(v);
  return v;
}
void RetireGenRecord(ViewGen *v) {
  g_gens.erase(v);
  v->serial = (v->serial + 1) & 31u;
  g_retired.push_back(v);
}
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
}  // namespace

namespace pkhost {
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
}  // namespace pkhost

namespace {
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
  int64_t dirty_cols = 0;           // columns of d_yt that may hold an earlier layout's features (SetLayout zeroes what a smaller one leaves behind)
  std::vector<int64_t> h_wave_off, h_raw_base, h_pad_base;
  // Rows of the layer stack and of d_ll: COMPACT -- utterance after utterance, each padded to a multiple of four rows, no
  // rows for the L + R context pads that separate two utterances in Yt (1 % of the matrix work at 10 s); the first
  // layer finds row j's features at column (f16 modes: row of the split copy) j + shift4[j / 4].
  std::vector<int64_t> h_out_base;  // first row of utterance u
  int64_t total_rows = 0;
  bool compact = false;
  std::vector<int32_t> h_shift4;
  int32_t *d_shift4 = nullptr;
  int64_t zero_span = 256;          // never-written columns at the end of feature row 0 of d_yt (the splice's zero source)
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
  // Columns behind this layout's last one that an earlier, larger layout filled: the padded rows of the last tile
  // are computed from them, and in the f16 modes whatever they produce counts towards the range verdict (a loud
  // batch followed by a smaller healthy one must not fail on the loud one's leftovers).  Zero again.
  if (col < b->dirty_cols) {
    HIP_TRY(hipMemset2DAsync(b->d_yt + col, sizeof(float) * b->ldy, 0, sizeof(float) * (size_t)(b->dirty_cols - col), kNumBins, b->stream));
    b->dirty_cols = col;
  }
  b->dirty_cols = std::max(b->dirty_cols, col);
  b->num_utts = num_utts;
  b->total_frames = raw;
  b->total_cols = col;
  b->scored = false;
  b->h_out_base.resize(num_utts);
  int64_t row = 0;
  for (int u = 0; u < num_utts; ++u) {
    b->h_out_base[u] = b->compact ? row : b->h_pad_base[u];
    if (b->h_T[u] > 0) row += RoundUp(b->h_T[u], 4);
  }
  b->total_rows = b->compact ? row : col;
  if (num_utts == 0) return 0;
  if (b->compact) {
    // one entry per group of four rows, for every row a tile of the layer stack can touch (the last tile's padding rows
    // and the rows past an utterance's last frame read real memory and are ignored)
    const int64_t groups = RoundUp(b->total_rows, kTileF16) / 4 + kTileF16 / 4;
    b->h_shift4.assign(groups, 0);
    int32_t shift = 0;
    int64_t g = 0;
    for (int u = 0; u < num_utts; ++u) {
      if (b->h_T[u] <= 0) continue;
      shift = (int32_t)(b->h_pad_base[u] - b->h_out_base[u]);
      for (const int64_t end = (b->h_out_base[u] + RoundUp(b->h_T[u], 4)) / 4; g < end; ++g) b->h_shift4[g] = shift;
    }
    for (; g < groups; ++g) b->h_shift4[g] = shift;
    if (shift + kTile > b->zero_span) return Fail(PK_MI355_E_INVALID, "internal: column shift %d exceeds the zero span", shift);
    HIP_TRY(hipMemcpyAsync(b->d_shift4, b->h_shift4.data(), sizeof(int32_t) * groups, hipMemcpyHostToDevice, b->stream));
  }
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
    if (b->range_status) {
      snprintf(b->range_msg, sizeof(b->range_msg), "%s", LastError());
      // views handed out by fetch_all(sync = 0) before this verdict hold withheld results: make them say so
      std::lock_guard<std::mutex> g(g_arena_mu);
      if (b->arena && b->arena->cur) b->arena->cur->withheld = true;
    }
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
  b->compact = true;
  if (const char *c = getenv("PK_MI355_COMPACT_ROWS")) b->compact = atoi(c) != 0;    // (A/B switch: 0 = row = column of Yt)
  // (the padding rows of the spliced operand read zeros from the end of feature row 0, at their column shift: at most
  // `pad` columns per utterance in front of them)
  b->zero_span = RoundUp((int64_t)max_utts * pad + 2 * kTile, 256);
  b->ldy = RoundUp(b->max_cols, b->chunk) + 256 + b->zero_span;
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
  if (b->compact) chk(hipMalloc(&b->d_shift4, sizeof(int32_t) * (size_t)(b->max_cols / 4 + kTile)));
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
  hipFree(b->d_wave_off); hipFree(b->d_raw_base); hipFree(b->d_pad_base); hipFree(b->d_T); hipFree(b->d_shift4);
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
  // the range verdict belongs to ONE score call: an empty batch, or a call that fails half-way, must not report
  // (or deliver under) the verdict of the call before it
  b->range_pending = false;
  b->range_status = 0;
  b->range_msg[0] = 0;
  b->scored = false;
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
  // Rows of the layer stack: compact (row out_base[u] + t is frame t of utterance u; the first layer finds its
  // features shift4[row / 4] columns -- f16 modes: rows of the split copy -- further on).
  const int N = am->num_pdfs;
  const bool f16 = IsF16(am->precision);
  if (f16) {
    if ((rc = BeginRange(b->exec, b->stream))) return rc;
    Scoped t(tm, PK_MI355_K_OTHER, b->stream);
    // only the rows the layer stack reads (the last chunk's padded rows and their right context): columns further
    // out hold zeros or an earlier call's split, and their maxima are not this call's
    // (compact rows: the last tile's padding rows sit at most one tile + the context behind the last column)
    const int split_rows = (int)std::min<int64_t>(b->ldy, RoundUp(b->total_cols, kTileF16) + 2 * kTileF16);
    LaunchSplitF16(b->d_yt, 1, b->ldy, split_rows, kNumBins, kNumBins, b->d_y2, 2 * kNumBins, ExpX(am, 0), RangeOf(b->exec, 0), b->stream);
  }
  const bool two = b->lanes == 2 && b->total_rows > b->chunk;
  if (f16 && b->lanes == 2 && !two) ClearHostRange(b->exec2);     // lane 2 takes no part in this call: no stale maxima
  if (two) {                                   // lane 2 starts when the features are ready
    HIP_TRY(hipEventRecord(b->ev_front, b->stream));
    HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_front, 0));
    if (f16 && (rc = BeginRange(b->exec2, b->stream2))) return rc;
  }
  int lane = 0;
  for (int64_t c0 = 0; c0 < b->total_rows; c0 += b->chunk, lane ^= 1) {
    const int rows = (int)std::min<int64_t>(b->chunk, b->total_rows - c0);
    hipStream_t s = (two && lane) ? b->stream2 : b->stream;
    const ExecBufs &e = (two && lane) ? b->exec2 : b->exec;
    rc = f16 ? RunLayersF16(am, e, b->d_y2 + c0 * 2 * kNumBins, 2 * kNumBins, rows, true,
                            prob_scale, b->d_ll + c0 * N, N, s, tm, nullptr, b->compact ? b->d_shift4 + c0 / 4 : nullptr)
             : RunLayers(am, e, b->d_yt + c0, b->ldy, kNumBins, rows, true, prob_scale,
                         b->d_ll + c0 * N, N, s, tm, nullptr, b->d_yt + (b->ldy - b->zero_span),
                         b->compact ? b->d_shift4 + c0 / 4 : nullptr);
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
  return b->d_ll + b->h_out_base[utt] * b->am->num_pdfs;
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
  if (b->total_rows > 0) {
    hipStream_t rs = ResultStream(b->device);
    if (!rs) return PK_MI355_E_DEVICE;
    if (!b->ev_scored) {
      HIP_TRY(hipEventCreateWithFlags(&b->ev_scored, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&b->ev_fetched, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(b->ev_scored, b->stream));
    HIP_TRY(hipStreamWaitEvent(rs, b->ev_scored, 0));
    HIP_TRY(hipMemcpyAsync(b->h_ll, b->d_ll, sizeof(float) * (size_t)b->total_rows * N, hipMemcpyDeviceToHost, rs));
    HIP_TRY(hipEventRecord(b->ev_fetched, rs));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_fetched, 0));
  }
  pk_mi355_am_t *gen = NewViewGen(b->arena, b->am, num_out);   // the views of an earlier fetch_all are void by contract
  for (int u = 0; u < num_out; ++u) {
    const int T = b->h_T[u];
    out[u].am = gen;
    out[u].log_prob.ncol = T;
    out[u].log_prob.nrow = T > 0 ? N : 0;
    out[u].log_prob.data = T > 0 ? b->h_ll + (size_t)b->h_out_base[u] * N : nullptr;
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
