// pk_kernels.h -- host-callable launchers of the gfx950 kernels (internal).
#ifndef PK_KERNELS_H_
#define PK_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pk_tables.h"

namespace pkmi {

// ---------------------------------------------------------------- front-end

// Per-utterance placement, all in device memory (arrays of num_utts entries).
struct UttLayout {
  const int64_t *wave_off;   // first PCM sample of the utterance in the wave buffer
  const int32_t *num_frames; // T_u
  const int64_t *raw_base;   // first row of the utterance in the raw fbank matrix [sum T][40]
  const int64_t *pad_base;   // first column of the utterance in the padded feature-major CMVN matrix
};

// fbank.cc:267-292 for a batch: PCM -> raw log-mel features, frame-major [sum T][40].
// wave is float PCM, or int16 PCM when wave_i16 != nullptr.
void LaunchFbank(const float *wave_f32, const int16_t *wave_i16, const UttLayout &utts,
                 int num_utts, int max_frames, const FrontendTables *d_tables, float *raw,
                 hipStream_t stream);

// parity-test hook: the kernel's logf (pk_logf.h) on n device floats
void LaunchLogfTest(const float *x, int n, const FrontendTables *d_tables, float *out, hipStream_t stream);
// parity-test hook: the kernel's 512-point real FFT (srfft.cc:371-461) on n device frames of 512 floats
void LaunchSrfft512Test(const float *frames, int n, const FrontendTables *d_tables, float *out, hipStream_t stream);

// The CMVN loader reads whole 64-frame tiles without clamping: every `raw` buffer handed to
// LaunchCmvn carries kCmvnRawSlack floats of slack behind its last frame and kCmvnRawLead floats
// in front of its first one (the tile in which the 600-frame window starts to slide reaches
// 24 frames before frame 0 of its utterance; the values are never used).
constexpr int kCmvnRawSlack = 64 * kNumBins;
constexpr int kCmvnRawLead = 1024;

// cmvn.cc:103-115 for a batch: raw [sum T][40] -> feature-major, edge-padded
// Yt[40][ldy]: utterance u occupies columns pad_base[u] .. pad_base[u]+T+left+right-1,
// its frame t at column pad_base[u]+left+t, the first/last frame replicated into
// the left/right pad (the clamp of am.cc:73-75 done once, at write time).
void LaunchCmvn(const float *raw, const UttLayout &utts, int num_utts, const float *d_global41,
                const CmvnTables *d_cmvn_tab, int left, int right, float *yt, int64_t ldy,
                hipStream_t stream);

// Already-normalised features [T][dim] (frame-major) -> padded feature-major Yt.
void LaunchPadTranspose(const float *feats, int T, int dim, int left, int right, float *yt,
                        int64_t ldy, int64_t col0, hipStream_t stream);

// ---------------------------------------------------------------- affine GEMM

constexpr int kTile = 128;   // block tile edge (both output dimensions)
constexpr int kBK = 16;      // k-step staged through LDS
constexpr int kChunkK = 512; // gemm.h:50 KC: the reference's k-blocking (sets rounding points)

// C[i][j] = sum_k P[k][i] * Q[k][j]  (+ bias, optional ReLU), fp32 MFMA.
// P: [K][ldp] (i contiguous), Q: [K][ldq] (j contiguous); K multiple of kBK;
// tiles_i x tiles_j tiles of 128x128 are computed, all buffers padded to that.
// splice_dim > 0: Q is the feature-major CMVN matrix and row k of the operand is
// Yt[k % splice_dim] shifted by k / splice_dim columns (am.cc:65-88 without
// materialising the spliced matrix).
struct GemmArgs {
  const float *P;
  int64_t ldp;
  const float *Q;
  int64_t ldq;
  int K;
  int splice_dim;
  int splice_ctx = 0;            // splice: context frames (operand rows k >= splice_ctx * splice_dim are padding)
  const float *splice_zero = nullptr;   // splice: zero floats within 4 GB of Q (the source of the padding rows): 128 + the
                                        // largest entry of splice_shift of them
  // splice, compact rows: output column j of this launch is NOT column j of Q but column j + splice_shift[j / 4] (one entry
  // per group of four columns; nullptr: no shift).  The batch scorer numbers its rows utterance after utterance with no
  // rows for the context pads between two utterances, while Q keeps the pads (the splice needs them).
  const int32_t *splice_shift = nullptr;
  const float *bias;   // indexed by i (bias_on_j = 0) or j (bias_on_j = 1); padded
  int bias_on_j;
  int relu;
  float *out;          // out[i * ldo + j]
  int64_t ldo;
  int tiles_i, tiles_j;
  int ring = 3;                  // spliced first layer, big tiles, K <= 512: LDS slabs in the ring (2 = four workgroups per CU)
  int half_j = 0;                // 1: ONE strip of 64 columns (64 x 64 tiles), whatever tiles_j says: the ragged end of a wide layer
  // Fused log-likelihood tail (tail_out != nullptr; frame-major output of the last affine layer, big tiles only --
  // GemmFusesTail()): the workgroup that completes a 128-row tile of logits turns those rows into log-likelihoods
  // (pk_tail_wave.h), so the logits are never read back by a second launch.
  float *tail_out = nullptr;       // [rows][tail_ld]
  int64_t tail_ld = 0;
  const float *tail_log_prior = nullptr;
  float tail_scale = 0.0f;
  int tail_n = 0;                  // valid columns (num_pdfs)
  int tail_rows = 0;               // valid rows of this launch
  unsigned *row_done = nullptr;    // one arrival counter per 128-row tile; the caller zeroes tiles_i of them in front of the launch
  int tail_walk = 0;               // A/B switch: super-tile columns of the tail variant's walk (0: the whole row of tiles)
  int walk_j = 8;                  // (set by the launcher) columns of a super-tile of the tile walk
#ifdef PK_MI355_DIAG
  int dbg = 0;                     // measurement switches (PK_DEBUG_TAILFLAGS: 16 = hand-off only)
  unsigned long long *dbg_counters = nullptr;
#endif
};
void LaunchGemm(const GemmArgs &a, hipStream_t stream);
// true if LaunchGemm(a) would run the kernel variant that carries the fused tail; min_tiles: the smallest launch
// (in 128 x 128 tiles, at least 384) that does
bool GemmFusesTail(const GemmArgs &a, int min_tiles);

// ---------------------------------------------------------------- affine GEMM, f16x3 mode

constexpr int kTileF16 = 256;   // block tile edge of the split-fp16 kernel
constexpr int kBKF16 = 32;      // its k-step (one v_mfma_f32_16x16x32_f16 = two k16 half-slabs); K is padded to a multiple of this

// D[m][n] = sum_k X[m][k] * W[n][k] + bias[n] (ReLU), X and W carried as fp16 (hi, lo)
// pairs, fp32 accumulation on the fp16 matrix cores (see gemm_f16.hip).
// Rows are "interleaved": logical k lives at (k / 8) * 16 + k % 8 (hi) and 8 halves
// further (lo), so a row of K values takes 2 K halves.  X: [rows][ldx] (ldx in halves;
// 2 * 40 for the spliced layer-1 view), W: [N pad][ldw].
// Output: fp32 rows (out_f32 != nullptr, last layer) or the next layer's interleaved rows.
struct GemmF16Args {
  const _Float16 *X;
  int64_t ldx;
  const _Float16 *W;
  int64_t ldw;
  int K;                 // multiple of kBKF16
  const float *bias;     // [N pad]
  int relu;
  float *out_f32;
  _Float16 *out;
  int64_t ldo;           // floats (out_f32) or halves (out)
  int tiles_m, tiles_n;  // 256 x 256 tiles
  const int32_t *row_shift4 = nullptr;   // compact rows: row r of this launch is row r + row_shift4[r / 4] of X (nullptr: none)
  int terms = 3;         // 3: hi hi + hi lo + lo hi (f16x3); 1: hi hi only (plain fp16 operands)
  int walk_m = 4, walk_n = 4;   // super-tile shape of the 16x16x32 kernel's tile walk (set by LaunchGemmF16)
  // Power-of-two operand scaling (device words, part of the model blob so that the one broadcast carries
  // them): the operands hold X * 2^e_in and W * 2^e_w; the epilogue undoes both and applies the exponent of
  // the NEXT layer's operand, 2^e_out, to what it writes as (hi, lo) halves (e_out points at a zero word for
  // fp32 output).  All three exact in fp32.
  const int32_t *e_w = nullptr, *e_in = nullptr, *e_out = nullptr;
  // Range words of the operand this launch writes (kRangeSlots uint32, or nullptr for fp32 output):
  // slot s receives max |hi| of the halves some wave wrote, as the bit pattern of a non-negative float.
  uint32_t *range = nullptr;
};
constexpr int kRangeSlots = 256;
void LaunchGemmF16(const GemmF16Args &a, hipStream_t stream);

// fp32 -> interleaved (hi, lo) fp16 rows: element (r, c) read at in[r * stride_r + c *
// stride_c]; row r of the output starts at out + r * ld_out (halves); columns
// cols..cols_pad-1 are zero-filled (cols_pad multiple of 8).
// NormalizeLayer (nnet.cc:62-75) between two f16x3 layers: fp32 rows [rows][ld_in] -> normalized interleaved
// (hi, lo) rows [rows][ld_out halves]; columns n..npad-1 zero (npad a multiple of 8, <= ld_in).
// e_x: device word, the operand's exponent (values are split as x * 2^e_x); range: its kRangeSlots range words
// (either may be nullptr: exponent 0 / no range tracking).
void LaunchNormalizeSplitF16(const float *in, int64_t ld_in, int rows, int n, int npad, _Float16 *out, int64_t ld_out,
                             const int32_t *e_x, uint32_t *range, hipStream_t stream);
void LaunchSplitF16(const float *in, int64_t stride_r, int64_t stride_c, int rows, int cols,
                    int cols_pad, _Float16 *out, int64_t ld_out, const int32_t *e_x, uint32_t *range,
                    hipStream_t stream);

// ---------------------------------------------------------------- row-wise / elementwise

// ReLU, nnet.cc:49-60 (x < 0 -> 0; -0.0 and NaN pass through)
void LaunchRelu(float *x, int64_t n, hipStream_t stream);

// Normalize, nnet.cc:62-75, for `frames` vectors of `dim` features; element
// (frame f, feature d) lives at x[f * stride_f + d * stride_d].
void LaunchNormalize(float *x, int frames, int dim, int64_t stride_f, int64_t stride_d,
                     hipStream_t stream);

// Feature-major [dim][ld_in] -> frame-major [frames][ld_out] and back.
void LaunchTransposeToRows(const float *in, int64_t ld_in, int dim, int frames, float *out,
                           int64_t ld_out, hipStream_t stream);
void LaunchTransposeToCols(const float *in, int64_t ld_in, int frames, int dim, float *out,
                           int64_t ld_out, hipStream_t stream);

enum TailMode {
  kTailSoftmaxProb = 0,    // SoftmaxLayer, nnet.cc:38-47: p = exp(x) / sum exp(x)
  kTailSoftmaxLoglik = 1,  // + am.cc:106-112 + decodable.cc:15 fused:
                           //   (max(log p, log 1e-20) - log_prior) * scale
  kTailLoglik = 2          // no softmax layer: (log(max(x, 1e-20)) - log_prior) * scale
};
// in: [rows][ld_in] frame-major; out: [rows][ld_out]; n valid columns.  reference_exact: the
// softmax modes reproduce the reference's operations one by one (libm expf, in-order float
// sum, division, libm logf) instead of the overflow-safe log-softmax.
void LaunchTail(int mode, bool reference_exact, const float *in, int64_t ld_in, int rows, int n,
                const float *log_prior, float scale, float *out, int64_t ld_out, hipStream_t stream);

// The softmax-log-likelihood tail with one wave per row (pk_tail_wave.h: the arithmetic of the fused last-layer
// launches).  false: the row is too wide for it (more than 8 192 columns); the caller uses LaunchTail.
bool LaunchTailWave(const float *in, int64_t ld_in, int rows, int n, const float *log_prior, float scale, float *out,
                    int64_t ld_out, hipStream_t stream);

// Device-side pk_decodable_loglikelihood (decodable.cc:24-31) for n (frame, trans_id) pairs.
void LaunchGather(const float *ll, int64_t ld, const int32_t *tid2pdf, int num_tids,
                  const int32_t *frames, const int32_t *tids, int n, float *out, hipStream_t stream);

}  // namespace pkmi

#endif  // PK_KERNELS_H_
