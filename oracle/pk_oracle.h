/*
 * pk_oracle.h -- CPU restatement of pocketkaldi's acoustic-scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pocketkaldi_amd/,
 * include/) may include, link or call this.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py use it, and only as
 * the checker / the reported CPU baseline.
 *
 * Every function cites the reference file:line (under /root/reference/) whose
 * arithmetic it restates.  Build flags are part of the contract: plain
 * x86-64 (no -march), -O2, -ffp-contract=off, so that every float expression
 * rounds exactly where the reference's g++ -O2 build rounds.
 *
 * Parity pin (see DESIGN.md "Oracle"):
 *   - srfft: bit-identical to the reference's own srfft.cc compiled into
 *     oracle/_ref (tests/test_oracle_ref.py) and within 1e-6 of the 128-point
 *     known-answer vector of test/srfft_test.cc.
 *   - sgemm: bit-identical to the reference's GEMM<float>::Gemm (gemm.cc +
 *     gemm_haswell.cc compiled into oracle/_ref).
 *   - fbank / cmvn: pinned by the reference's Kaldi dumps
 *     (test/data/fbank*_en-us-hello.wav.txt) at the precision those dumps have.
 *   - layers: pinned by the known answers of test/nnet_test.cc.
 *   The reference's vector.cc / matrix.cc cannot be compiled in this image
 *   (they include <cblas.h>, which the image lacks), so fbank.cc, cmvn.cc,
 *   nnet.cc, am.cc and decodable.cc cannot be linked into oracle/_ref.
 */
#ifndef PK_ORACLE_H_
#define PK_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of the reference (fbank.h:7-13, cmvn.h:10-11) ---- */
#define PKO_SAMPLE_RATE 16000
#define PKO_FRAME_LENGTH 400
#define PKO_FRAME_SHIFT 160
#define PKO_FFT_SIZE 512
#define PKO_NUM_BINS 40
#define PKO_CMVN_WINDOW 600
#define PKO_CMVN_GLOBAL_FRAMES 200

/* ---- split-radix real FFT (srfft.cc) ---- */
typedef struct pko_fft_block { int off; int logm; } pko_fft_block_t;

typedef struct pko_srfft {
  int n_real;            /* number of real points (512)              */
  int n_cplx;            /* n_real / 2                                */
  int logn;              /* log2(n_cplx)                              */
  int num_blocks;        /* split-radix sub-transforms, largest first */
  pko_fft_block_t *blocks;
  /* per stage logm (index logm, valid for logm >= 4): 6 tables indexed by n */
  float *tw[32];         /* tw[logm] -> 6 * (m/4) floats: cn, spcn, smcn, c3n, spc3n, smc3n */
  int *bitrev;           /* n_cplx entries */
  float *post_re;        /* real post-pass twiddle recurrence, k = 0..n_cplx/2 */
  float *post_im;
} pko_srfft_t;

int  pko_srfft_init(pko_srfft_t *p, int n_real);
void pko_srfft_free(pko_srfft_t *p);
/* in-place forward real FFT, output packed as srfft.cc:444-447 */
void pko_srfft_forward(const pko_srfft_t *p, float *data);

/* ---- fbank (fbank.cc) ---- */
typedef struct pko_fbank {
  pko_srfft_t fft;
  float window[PKO_FRAME_LENGTH];
  int   mel_offset[PKO_NUM_BINS];
  int   mel_len[PKO_NUM_BINS];
  float mel_weight[PKO_NUM_BINS][PKO_FFT_SIZE / 2];
} pko_fbank_t;

int  pko_fbank_init(pko_fbank_t *fb);
void pko_fbank_free(pko_fbank_t *fb);
int  pko_num_frames(int num_samples);
/* out is [T][40] (frame-major, the memory image of pk_matrix_t{nrow=40,ncol=T}) */
void pko_fbank_compute(const pko_fbank_t *fb, const float *wave, int num_samples,
                       float *out);
/* single already-extracted frame (400 raw samples) -> 40 log-mel values;
 * also exposes the intermediate 512-float buffer after the FFT if spec != NULL */
void pko_fbank_frame(const pko_fbank_t *fb, const float *samples400, float *out40,
                     float *spec512);

/* ---- CMVN (cmvn.cc) ---- */
void pko_cmvn(const float *global_stats41, const float *raw, int T, float *out);

/* ---- splice (am.cc:65-88) ---- */
void pko_splice(const float *feats, int T, int dim, int left, int right, float *out);

/* vector.cc:334-339 ApplyLog: libm's logf, element by element (what the reference calls). */
void pko_logf_array(const float *x, int n, float *out);

/* ---- SGEMM with the reference's accumulation order (gemm.cc, gemm_haswell.cc) ----
 * C[m][n] (row-major, ldc) = A[m][k] (row-major, lda) * B[k][n] (row-major, ldb)
 * Per element: k ascending fused-multiply-add chain from 0 inside each chunk of
 * 512 k's; chunks combined by one float add each, in order.                      */
void pko_sgemm(int m, int n, int k, const float *A, int lda, const float *B, int ldb,
               float *C, int ldc);
/* same arithmetic, plain loops, no SIMD (cross-check of the blocked kernel) */
void pko_sgemm_naive(int m, int n, int k, const float *A, int lda, const float *B,
                     int ldb, float *C, int ldc);

/* ---- layers (nnet.cc) ---- */
enum { PKO_LINEAR = 0, PKO_RELU = 1, PKO_NORMALIZE = 2, PKO_SOFTMAX = 3 };

typedef struct pko_layer {
  int type;
  int in_dim, out_dim;   /* linear only */
  float *Wt;             /* [in_dim][out_dim], the transposed copy nnet.cc:11-20 keeps */
  float *b;              /* [out_dim] */
} pko_layer_t;

typedef struct pko_nnet {
  int num_layers;
  pko_layer_t *layers;
} pko_nnet_t;

void pko_nnet_init(pko_nnet_t *nn);
void pko_nnet_free(pko_nnet_t *nn);
/* W is [out_dim][in_dim] as stored in the model file (Kaldi order) */
int  pko_nnet_add_linear(pko_nnet_t *nn, int in_dim, int out_dim, const float *W,
                         const float *b);
int  pko_nnet_add_simple(pko_nnet_t *nn, int type);
/* in [T][in_dim] -> out [T][out_dim]; returns out_dim (or <0) */
int  pko_nnet_propagate(const pko_nnet_t *nn, const float *in, int T, int in_dim,
                        float *out, int out_cap_per_row);
int  pko_nnet_output_dim(const pko_nnet_t *nn, int in_dim);

void pko_relu(float *x, int64_t n);
void pko_normalize_rows(float *x, int T, int dim);
void pko_softmax_rows(float *x, int T, int dim);

/* ---- acoustic model tail + decodable (am.cc:90-115, decodable.cc:8-17) ----
 * feats [T][feat_dim] (CMVN'd) -> loglik [T][num_pdfs], already "- log prior"
 * and "* prob_scale".  prior holds probabilities (log taken here, am.cc:43).   */
int pko_am_compute(const pko_nnet_t *nn, const float *prior, int num_pdfs, int left,
                   int right, const float *feats, int T, int feat_dim,
                   float prob_scale, float *loglik);

/* ---- WAV ingestion (pcm_reader.cc:45-220): returns sample count or <0 ---- */
int pko_wav_read(const char *path, float **samples_out);

/* ---- model-file sections (vector.cc:393-425, matrix.cc:288-319, nnet.cc:80-147) ---- */
int pko_read_vec_f32(const char *path, float **out, int *dim);
int pko_nnet_read(pko_nnet_t *nn, const char *path);

#ifdef __cplusplus
}
#endif
#endif  /* PK_ORACLE_H_ */
