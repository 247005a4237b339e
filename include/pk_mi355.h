/*
 * pk_mi355.h -- C ABI of libpk_mi355.so: pocketkaldi's acoustic-scoring hot path
 * (fbank -> CMVN -> splice -> nnet -> log-likelihoods) on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary.  Plain C types only; every entry point cites the
 * reference interface (path:line under the pocketkaldi tree) it replaces.
 * INTEGRATION.md shows the reference-side change that binds to it.
 *
 * Error model: the reference's boundary has no status channel (decodable.h:20-41
 * are void/float/bool; misuse is assert()).  Here every pk_mi355_* function that
 * can fail returns 0 on success and a negative code on failure, and
 * pk_mi355_last_error() returns a thread-local message.  The four pk_decodable_*
 * functions keep the reference signatures; a device failure inside
 * pk_decodable_init() leaves log_prob empty (ncol = 0) and sets the error string.
 * No C++ exception crosses this ABI.  There is no CPU fallback: without a
 * usable gfx950 device every compute entry point fails.
 */
#ifndef PK_MI355_H_
#define PK_MI355_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data-layout contract (field order and meaning as in the reference) ----
 * Inside the reference tree (include/reference_binding/decodable.h is the file that takes the
 * place of src/decodable.h there) the reference's own matrix.h / vector.h / nnet.h come first
 * and their declarations are the ones used: the guards below are the reference's include guards
 * (matrix.h:5, vector.h:26) and layer-kind macros (nnet.h:14-17).                              */

/* matrix.h:20-24 -- column-major; memory is [ncol][nrow], i.e. for features and
 * log-likelihoods one frame (column) is contiguous.                              */
#ifndef POCKETKALDI_MATRIX_H_   /* inside the reference tree matrix.h has already declared it */
typedef struct pk_matrix_t {
  int ncol;
  int nrow;
  float *data;
} pk_matrix_t;
#endif

/* vector.h:39-42 */
#ifndef POCKETKALDI_VECTOR_H_   /* likewise vector.h */
typedef struct pk_vector_t {
  int dim;
  float *data;
} pk_vector_t;
#endif

/* Stands where `AcousticModel *` stands in the reference (am.h:23-52): holds the
 * nnet weights (in HBM), log-priors, context and the tid->pdf map (host).        */
#ifdef PK_MI355_AM_T            /* reference_binding/decodable.h: the handle keeps the NAME the
                                   reference's translation units use for this pointer
                                   (pocketkaldi::AcousticModel, decodable.h:17, pocketkaldi.h:38);
                                   across the C ABI it is the same opaque pointer                 */
typedef PK_MI355_AM_T pk_mi355_am_t;
#else
typedef struct pk_mi355_am pk_mi355_am_t;
#endif

/* decodable.h:15-18 -- same size and field offsets on LP64 (16 + 8 bytes).       */
typedef struct pk_decodable_t {
  pk_matrix_t log_prob;
  pk_mi355_am_t *am;
} pk_decodable_t;

/* Layer kinds, nnet.h:13-16 / nnet.h:28-33 */
#ifndef PK_NNET_LINEAR_LAYER    /* nnet.h #defines the same names to the same values */
enum {
  PK_NNET_LINEAR_LAYER = 0,
  PK_NNET_RELU_LAYER = 1,
  PK_NNET_NORMALIZE_LAYER = 2,
  PK_NNET_SOFTMAX_LAYER = 3
};
#endif

/* Error codes */
enum {
  PK_MI355_OK = 0,
  PK_MI355_E_INVALID = -1,   /* bad argument / shape mismatch            */
  PK_MI355_E_DEVICE = -2,    /* HIP runtime failure or no gfx950 device  */
  PK_MI355_E_IO = -3,        /* file missing or corrupted                */
  PK_MI355_E_STATE = -4,     /* call order (e.g. model not finalized)    */
  PK_MI355_E_RANGE = -5      /* f16x3 / f16: an operand left the fp16 split's range (results withheld) */
};

const char *pk_mi355_last_error(void);

/* Select the HIP device used by objects this THREAD creates afterwards (default 0; like
 * hipSetDevice the setting is per host thread).  One process drives one GPU (one rank per GPU
 * in multi-GPU runs).
 *
 * Threads: a model may be shared by host threads.  pk_decodable_init, pk_mi355_nnet_propagate and
 * pk_mi355_process_acoustic use one device workspace per model and serialise on a per-model lock
 * (the reference's versions allocate per call, nnet.cc:149-163); the four pk_decodable_* readers
 * are lock-free.  A pk_mi355_batch_t belongs to one thread at a time; different batches of one
 * model may be driven from different threads.                                                 */
int pk_mi355_set_device(int device);

/* ------------------------------------------------------------------------- */
/* The four functions decoder.cc consumes -- decodable.h:20-41, decodable.cc:8-36 */
/* ------------------------------------------------------------------------- */

/* decodable.cc:8-17.  feats: CMVN'd features, {ncol = T, nrow = feat_dim}, host,
 * borrowed.  Allocates self->log_prob {ncol = T, nrow = num_pdfs} on the host with
 * malloc(), fills it with (log softmax - log prior) * prob_scale.                */
void pk_decodable_init(pk_decodable_t *self, pk_mi355_am_t *am, float prob_scale,
                       const pk_matrix_t *feats);
/* decodable.cc:19-22 */
void pk_decodable_destroy(pk_decodable_t *self);
/* decodable.cc:24-31: log_prob[frame][tid2pdf[trans_id]] -- host lookup          */
float pk_decodable_loglikelihood(pk_decodable_t *self, int frame, int trans_id);
/* decodable.cc:33-36 (frame = -1 on the decoder's first poll -> false)           */
bool pk_decodable_islastframe(pk_decodable_t *self, int frame);

/* ------------------------------------------------------------------------- */
/* Acoustic model -- replaces AcousticModel (am.h:23-52) + Nnet (nnet.h:88-104)   */
/* ------------------------------------------------------------------------- */

pk_mi355_am_t *pk_mi355_am_create(void);
void pk_mi355_am_destroy(pk_mi355_am_t *am);

/* LinearLayer(W, b), nnet.cc:11-20.  W is [out_dim][in_dim] row-major, the order
 * of the model file (convert_am.py:77-83); b is [out_dim].                        */
int pk_mi355_am_add_linear(pk_mi355_am_t *am, int in_dim, int out_dim, const float *W,
                           const float *b);
/* ReLULayer / NormalizeLayer / SoftmaxLayer, nnet.cc:38-75                       */
int pk_mi355_am_add_layer(pk_mi355_am_t *am, int layer_type);

/* Arithmetic of the affine layers (set before finalize / read; default F32).
 *   F32   : fp32 MFMA, accumulation order of the reference's SGEMM -- bit-identical layers.
 *   F16X3 : every fp32 operand carried as an fp16 (hi, lo) pair, three fp16 MFMAs per
 *           product, fp32 accumulation: ~1e-6 relative on log-likelihoods (inside the
 *           1e-4 contract, not bit-exact), several times faster.  Supports
 *           (Linear [ReLU] [Normalize])+ [Softmax] networks.
 *           RANGE (round 4).  fp16 holds 2^-24 .. 65504 and the lo half sits 2^-12 below its value, so the
 *           accuracy above holds only while operands sit well inside that window.  Weights: every affine
 *           layer's W is multiplied by an exact power of two at finalize (max |W| -> [2^13, 2^14)) and the
 *           GEMM's epilogue divides it out again -- automatic, any training scale.  Activations: the operand
 *           of every affine layer carries an exponent too (x * 2^e is what is split; default 0, set by
 *           pk_mi355_am_calibrate / pk_mi355_am_set_input_exponents), and every call CHECKS what it wrote:
 *           an operand that reached the clamp at 65504, or whose largest magnitude stayed below 2^-5
 *           (every lo half subnormal), fails the call with PK_MI355_E_RANGE -- pk_decodable_init leaves
 *           log_prob empty, the batch calls return the code (score with sync, synchronize, fetch, fetch_all)
 *           -- and pk_mi355_last_error() names the layer.  Nothing out of range is returned quietly.
 *           NaN is not carried: where the reference's NormalizeLayer turns an all-zero row into NaN (0 * inf,
 *           nnet.cc:62-75) this mode keeps the row zero.  F32 reproduces the NaN.
 *   F16   : plain fp16 operands (the hi halves only), ONE fp16 MFMA per product, fp32 accumulation --
 *           the throughput ceiling of the fp16 matrix cores at a STATED, looser tolerance: ~1e-3
 *           relative on log-likelihoods, OUTSIDE the 1e-4 contract of the path (SURVEY section 7,
 *           step 8: "its own, separately reported, tolerance").  Same network restrictions as F16X3. */
enum { PK_MI355_PRECISION_F32 = 0, PK_MI355_PRECISION_F16X3 = 1, PK_MI355_PRECISION_F16 = 2 };
int pk_mi355_am_set_precision(pk_mi355_am_t *am, int precision);
int pk_mi355_am_precision(const pk_mi355_am_t *am);

/* F16X3 / F16: the operand exponents (no-ops returning 0 exponents / success in F32).
 * get: w_exp[l] = the power of two the weights of affine layer l were multiplied by at finalize, x_exp[l] = the
 *      exponent of that layer's input operand; returns the number of affine layers (<= capacity) or a negative code.
 * set_input_exponents: count == number of affine layers, each in [-30, 30].
 * calibrate: feats as for pk_decodable_init (CMVN'd features, {ncol = T, nrow = feat_dim}).  Runs the network on
 *      them, reads every operand's largest magnitude back from the device and sets its exponent so that it lands in
 *      [2^3, 2^4) (4 096 x headroom to the clamp; higher placements buy no accuracy), layer by layer, front to back.  The exponents live in the
 *      weight blob: pk_mi355_am_broadcast carries the root's calibration to every rank.  Calibrate while nothing
 *      is being scored with the model.  pk_mi355_batch_calibrate (below) does the same from the batch's waves. */
int pk_mi355_am_get_exponents(pk_mi355_am_t *am, int32_t *w_exp, int32_t *x_exp, int capacity);
int pk_mi355_am_set_input_exponents(pk_mi355_am_t *am, const int32_t *x_exp, int count);
int pk_mi355_am_calibrate(pk_mi355_am_t *am, const pk_matrix_t *feats);

/* Arithmetic of the softmax / log-likelihood tail (any time; default STABLE).
 *   STABLE    : log-softmax with the row maximum subtracted -- finite for any logits, ~1e-6
 *               (measured 1.2e-6) from the reference wherever the reference does not overflow.
 *   REFERENCE : the reference's operations one by one (nnet.cc:38-47 -> vector.cc:265-277,
 *               am.cc:106-112): libm expf, float sum in column order, division, floor,
 *               libm logf, prior, scale.  With F32 precision the log-likelihoods are then the
 *               reference's bit patterns, its overflow for logits above 88.7 included.       */
enum { PK_MI355_SOFTMAX_STABLE = 0, PK_MI355_SOFTMAX_REFERENCE = 1 };
int pk_mi355_am_set_softmax(pk_mi355_am_t *am, int mode);
int pk_mi355_am_softmax(const pk_mi355_am_t *am);

/* AcousticModel::Read tail, am.cc:41-60: prior holds probabilities (the log is
 * taken here); tid2pdf is indexed by transition-id.  tid2pdf may be NULL (then
 * pk_decodable_loglikelihood treats trans_id as the pdf index).  Uploads the
 * packed weight blob to HBM.                                                     */
int pk_mi355_am_finalize(pk_mi355_am_t *am, const float *prior, int num_pdfs,
                         int left_context, int right_context, const int32_t *tid2pdf,
                         int num_tids);

/* Nnet::Read (nnet.cc:132-147) + prior / tid2pdf files (am.cc:28-60): the
 * NNT0/LAY0/MAT0/VEC0 little-endian section files.  tid2pdf_path may be NULL.    */
int pk_mi355_am_read(pk_mi355_am_t *am, const char *nnet_path, const char *prior_path,
                     const char *tid2pdf_path, int left_context, int right_context,
                     int num_pdfs);

/* pk_load's share of this path (pocketkaldi.cc:72-144): read the reference's "key = value" model
 * file (configuration.cc:16-72: '#' comments, keys case-insensitive, relative paths resolved
 * against the file's directory) and load what acoustic scoring needs -- cmvn_stats (VEC0 of 40
 * sums + count, into cmvn_stats41) and the AcousticModel keys nnet, prior, left_context,
 * right_context, num_pdfs, tid2pdf (am.cc:22-62).  The decoder's keys (fst, symbol_table) are
 * not touched.  precision: PK_MI355_PRECISION_*.  On success *am_out is a finalized model.     */
int pk_mi355_load(const char *config_path, int precision, pk_mi355_am_t **am_out, float *cmvn_stats41);

int pk_mi355_am_num_pdfs(const pk_mi355_am_t *am);      /* am.h:38 */
int pk_mi355_am_input_dim(const pk_mi355_am_t *am);     /* spliced width */
int pk_mi355_am_transition_to_pdf(const pk_mi355_am_t *am, int trans_id); /* am.h:30-32 */

/* The packed device weight blob (weights, biases, log-priors), for the one RCCL
 * broadcast of multi-GPU runs: every rank builds the same model structure, rank 0
 * holds the real values, all ranks broadcast [ptr, ptr+bytes) from rank 0.       */
void *pk_mi355_am_blob_device_ptr(pk_mi355_am_t *am);
size_t pk_mi355_am_blob_bytes(const pk_mi355_am_t *am);
/* That broadcast, for a C/C++ host (the multi-GPU form of pk_load, pocketkaldi.cc:72-144): ONE
 * ncclBroadcast of the blob from rank `root` into every rank's blob, in place, over the caller's
 * RCCL communicator.  rccl_comm: the caller's ncclComm_t (one process per GPU, the communicator's
 * device is the model's device).  stream: a hipStream_t to enqueue on (the caller synchronises it
 * before scoring), or NULL: the call then uses a stream of its own and returns when the
 * broadcast has completed.  tid2pdf and the layer structure are host-side and are NOT sent: every
 * rank reads the (small) tid2pdf file itself or builds the same structure, as bench.py does.
 * RCCL is bound at run time to the copy ALREADY LOADED in the process -- the one the communicator
 * was created with (global scope, or RTLD_LOCAL as under Python: found by soname) -- or to
 * $PK_MI355_RCCL_LIB when set; a second copy is never loaded implicitly.  libpk_mi355.so has no
 * link-time dependency on RCCL.  Behaviour at more than one rank has been exercised only through
 * the Python path's gloo rehearsal so far.                                 */
int pk_mi355_am_broadcast(pk_mi355_am_t *am, void *rccl_comm, int root, void *stream);
/* Out-of-place form: every rank still RECEIVES into `am`'s blob; on the root the bytes sent are
 * `src`'s blob (a second model with the same layer structure on the same device, e.g. one just
 * read from new files while `am` keeps serving).  src == NULL is pk_mi355_am_broadcast.  At one
 * rank this is RCCL copying src's blob into am's -- the form the one-GPU tests use to show that the
 * collective really writes the destination model's weights.                                     */
int pk_mi355_am_broadcast_from(pk_mi355_am_t *am, pk_mi355_am_t *src, void *rccl_comm, int root,
                               void *stream);

/* Nnet::Propagate, nnet.cc:149-163: in {ncol = T, nrow = in_dim} host ->
 * out {ncol = T, nrow = out_dim} host (out->data is (re)allocated with malloc). */
int pk_mi355_nnet_propagate(pk_mi355_am_t *am, const pk_matrix_t *in, pk_matrix_t *out);

/* ------------------------------------------------------------------------- */
/* Front-end -- replaces Fbank (fbank.h:47-110) and CMVN (cmvn.h:17-45)           */
/* ------------------------------------------------------------------------- */

/* Fbank::CalcNumFrames, fbank.cc:35-42 */
int pk_mi355_num_frames(int num_samples);

/* Fbank::Compute, fbank.cc:267-292: wave (host, 16 kHz, unscaled sample values)
 * -> out {ncol = T, nrow = 40} host (resized with malloc/realloc).               */
int pk_mi355_fbank_compute(const pk_vector_t *wave, pk_matrix_t *out);

/* CMVN(global_stats, raw).GetFrame(t) for t = 0..T-1, cmvn.cc:103-125:
 * global_stats dim 41 (40 sums + count); raw {ncol = T, nrow = 40} host ->
 * out {ncol = T, nrow = 40} host.                                                */
int pk_mi355_cmvn_apply(const pk_vector_t *global_stats, const pk_matrix_t *raw,
                        pk_matrix_t *out);

/* ------------------------------------------------------------------------- */
/* Batched scorer: many utterances in flight, PCM in -> log-likelihoods out,     */
/* everything device-resident (the stages of pk_process, pocketkaldi.cc:176-218). */
/* ------------------------------------------------------------------------- */

typedef struct pk_mi355_batch pk_mi355_batch_t;

/* global_stats41: the cmvn_stats vector pk_load reads (pocketkaldi.cc:96-112).
 * Capacity: at most max_utts utterances and max_total_samples PCM samples.       */
pk_mi355_batch_t *pk_mi355_batch_create(pk_mi355_am_t *am, const float *global_stats41,
                                        int max_utts, int64_t max_total_samples);
void pk_mi355_batch_destroy(pk_mi355_batch_t *b);

/* Hand over B utterances as host PCM (float sample values as pk_16kpcm_read
 * produces them, pcm_reader.cc:189-211); copied to HBM.                          */
int pk_mi355_batch_set_waves(pk_mi355_batch_t *b, const pk_vector_t *waves, int num_utts);
/* Same, from one concatenated host int16 buffer (the WAV payload itself).        */
int pk_mi355_batch_set_waves_i16(pk_mi355_batch_t *b, const int16_t *samples,
                                 const int *num_samples, int num_utts);
/* Same, PCM already resident in HBM: d_samples is a device pointer to the
 * concatenated float samples of all utterances.                                  */
int pk_mi355_batch_set_waves_device(pk_mi355_batch_t *b, const float *d_samples,
                                    const int *num_samples, int num_utts);

/* Run fbank -> CMVN -> nnet -> log-likelihood tail for the current utterances.
 * Asynchronous on the batch's stream unless sync != 0.  Results stay in HBM.     */
int pk_mi355_batch_score(pk_mi355_batch_t *b, float prob_scale, int sync);
int pk_mi355_batch_synchronize(pk_mi355_batch_t *b);
/* F16X3 / F16: pk_mi355_am_calibrate on the utterances currently set (front-end included).  Overwrites the
 * batch's results; score again afterwards.                                                               */
int pk_mi355_batch_calibrate(pk_mi355_batch_t *b);

int pk_mi355_batch_num_utts(const pk_mi355_batch_t *b);
int pk_mi355_batch_num_frames(const pk_mi355_batch_t *b, int utt);
int64_t pk_mi355_batch_total_frames(const pk_mi355_batch_t *b);

/* Device pointer to utterance utt's [T][num_pdfs] log-likelihoods.               */
const float *pk_mi355_batch_loglik_device(const pk_mi355_batch_t *b, int utt);
/* Copy utterance utt's results into a host decodable (malloc'd log_prob), ready
 * for Decoder::Decode (decoder.cc:39).                                           */
int pk_mi355_batch_fetch(pk_mi355_batch_t *b, int utt, pk_decodable_t *out);
/* All utterances at once: ONE device-to-host transfer into a page-locked arena the batch owns
 * (made on first use), and out[0..num_out) (num_out == pk_mi355_batch_num_utts) filled as
 * decodables whose log_prob VIEWS that arena -- same fields, same [T][num_pdfs] layout, usable
 * by Decoder::Decode like any other (decoder.cc:39); their `am` field is an opaque tagged handle
 * (that is how pk_decodable_destroy tells a view from a malloc'd matrix): copy such a struct whole
 * and hand it to the pk_decodable_* functions only.  pk_decodable_destroy on such a view
 * frees nothing of the caller's and may happen at any time, also after pk_mi355_batch_destroy
 * (pocketkaldi.cc:247 destroys its decodable unconditionally): the arena is released when the
 * batch is gone AND the views of its last fetch_all have been destroyed (each fetch_all is a
 * generation of its own: destroying views of an earlier one never releases memory under the
 * current ones).  Rely on the CONTENTS of the views only until the batch is scored again or
 * destroyed; the views of the LAST fetch_all in fact stay readable until the last of them goes.  With sync == 0 the copy is queued on
 * the device's result stream, ordered after this batch's scoring and before anything queued
 * later on the batch's stream; pk_mi355_batch_synchronize completes it, so it overlaps another
 * batch's scoring.  F16X3 / F16 with sync == 0: the views exist before the score call's range verdict does; if
 * pk_mi355_batch_synchronize then returns PK_MI355_E_RANGE the results are withheld -- pk_decodable_loglikelihood
 * on those views returns NaN from then on (do not read log_prob.data of views whose synchronize failed).      */
int pk_mi355_batch_fetch_all(pk_mi355_batch_t *b, pk_decodable_t *out, int num_out, int sync);
/* Intermediate stages, for parity tests: raw fbank / CMVN'd features of utt,
 * copied to host as [T][40].                                                     */
int pk_mi355_batch_fetch_fbank(pk_mi355_batch_t *b, int utt, float *out);
int pk_mi355_batch_fetch_cmvn(pk_mi355_batch_t *b, int utt, float *out);
/* Parity-test hook: the front-end's logf (fbank.cc:244-245 -> vector.cc:334-339 -> libm logf,
 * restated for the device in csrc/pk_logf.h) on n host floats (positive normal, +inf or NaN).  */
int pk_mi355_test_logf(const float *x, int n, float *out);
/* Parity-test hook: the front-end's 512-point real FFT alone -- pk_srfft_compute (srfft.cc:371-461,
 * forward) as FbankKernel runs it, on num_frames host frames of 512 floats; spectra receives the
 * reference's packed layout [Re0+Im0, Re0-Im0, Re1, Im1, ..., Re255, Im255] per frame.            */
int pk_mi355_test_srfft512(const float *frames, int num_frames, float *spectra);

/* Device-side pk_decodable_loglikelihood (decodable.cc:24-31) for a GPU-resident consumer
 * (decoder.cc:252-279 evaluates one (frame, transition-id) pair per arc): n pairs in
 * device memory -> out[i] = log_prob[frame[i]][tid2pdf[trans_id[i]]] of utterance utt,
 * on the batch's stream, nothing leaves HBM.                                       */
int pk_mi355_batch_gather_loglik(pk_mi355_batch_t *b, int utt, const int32_t *d_frames,
                                 const int32_t *d_trans_ids, int n, float *d_out);

/* Device buffers for callers without a HIP toolchain of their own (the inputs of
 * pk_mi355_batch_set_waves_device / pk_mi355_batch_gather_loglik).  A process must use ONE
 * HIP runtime: allocate through these (or through the runtime this library is bound to).
 * kind: 1 host->device, 2 device->host, 3 device->device; synchronous.              */
void *pk_mi355_device_malloc(size_t bytes);
void pk_mi355_device_free(void *ptr);
int pk_mi355_memcpy(void *dst, const void *src, size_t bytes, int kind);
/* Page-locked host memory for PCM handed to pk_mi355_batch_set_waves_i16 / _set_waves: from
 * such a buffer the upload is a true asynchronous DMA that overlaps another batch's scoring
 * and result transfer (from pageable memory the runtime stages it and the call blocks).     */
void *pk_mi355_host_malloc(size_t bytes);
void pk_mi355_host_free(void *ptr);

/* The HIP stream the batch launches on (hipStream_t as void*), so callers can
 * bracket it with their own events.                                              */
void *pk_mi355_batch_stream(pk_mi355_batch_t *b);

/* Per-kernel timing of the LAST score call measured with hipEvents on the batch's
 * stream (enable before scoring).  Kinds index the arrays below.                 */
enum {
  PK_MI355_K_FBANK = 0,
  PK_MI355_K_CMVN = 1,
  PK_MI355_K_GEMM = 2,       /* all affine-layer launches */
  PK_MI355_K_TAIL = 3,       /* log-softmax / prior / scale */
  PK_MI355_K_OTHER = 4,
  PK_MI355_K_COUNT = 5
};
int pk_mi355_batch_enable_timing(pk_mi355_batch_t *b, int enable);
/* total milliseconds and launch count per kind for the last score call           */
int pk_mi355_batch_get_timing(pk_mi355_batch_t *b, float ms[PK_MI355_K_COUNT],
                              int launches[PK_MI355_K_COUNT]);
/* algorithmic FLOPs of the affine layers per frame (2 * sum K*N)                 */
double pk_mi355_am_flops_per_frame(const pk_mi355_am_t *am);

/* ------------------------------------------------------------------------- */
/* The acoustic half of pk_process (pocketkaldi.cc:176-218) and its input reader   */
/* ------------------------------------------------------------------------- */

/* pk_16kpcm_read, pcm_reader.cc:45-220: strict 44-byte-header RIFF/WAVE PCM, mono,
 * 16 kHz, 8/16/32-bit; samples are stored unscaled as float.  pcm_data->data is
 * (re)allocated with realloc().  Host only, needs no GPU.                          */
int pk_mi355_16kpcm_read(const char *filename, pk_vector_t *pcm_data);

/* Stages 1-3 of pk_process fused on the device: raw_wave -> fbank -> CMVN -> nnet ->
 * decodable (host log_prob, ready for Decoder::Decode).  An empty wave gives an empty
 * decodable (pocketkaldi.cc:180-184).  verbose != 0 prints the reference's per-stage
 * lines ("Fbank: ..ms", "CMVN: ..ms", "NNET: ..ms", pocketkaldi.cc:194,206,218) to stderr. */
int pk_mi355_process_acoustic(pk_mi355_am_t *am, const pk_vector_t *cmvn_global_stats,
                              const pk_vector_t *raw_wave, float prob_scale, pk_decodable_t *out,
                              int verbose);

/* Library / device facts */
int pk_mi355_device_count(void);
const char *pk_mi355_version(void);

#ifdef __cplusplus
}
#endif
#endif  /* PK_MI355_H_ */
