/*
 * pk_oracle.c -- CPU restatement of pocketkaldi's acoustic-scoring hot path.
 * TEST INFRASTRUCTURE ONLY (see pk_oracle.h).  Plain C99, scalar, single thread.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (no -march).  The only fused
 * multiply-adds are the explicit fmaf() calls in pko_sgemm_naive(), which model
 * the reference's AVX2 vfmadd231ps micro-kernel (gemm_haswell.cc:122-282).
 *
 * The code is organised around data flow (iterative FFT passes over a block
 * schedule, fused per-frame loops), not around the reference's call tree; what
 * is kept exactly is WHERE each float expression rounds.
 */
#define _POSIX_C_SOURCE 200809L
#include "pk_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* srfft.cc:37-39 uses the full-precision 2*pi; fbank.cc:18-20 a truncated one. */
#define TWO_PI_FULL 6.283185307179586476925286766559005
#define TWO_PI_TRUNC 6.28318530718
#define SQRT_HALF 0.70710678118654752440

/* ------------------------------------------------------------------------- */
/* split-radix real FFT                                                       */
/* ------------------------------------------------------------------------- */

/* The reference recursion (srfft.cc:95-237) visits a sub-transform of 2^logm
 * points at offset off, then recurses on (off, logm-1), (off+m/2, logm-2) and
 * (off+3m/4, logm-2).  A block only depends on its parent, and parents are
 * always larger, so running all blocks in order of decreasing size is the same
 * dataflow.  We enumerate the blocks once here.                               */
static int enumerate_blocks(int off, int logm, pko_fft_block_t *out, int count) {
  if (logm <= 0) return count;
  if (out) { out[count].off = off; out[count].logm = logm; }
  count++;
  if (logm >= 2) {
    int m = 1 << logm;
    count = enumerate_blocks(off, logm - 1, out, count);
    count = enumerate_blocks(off + m / 2, logm - 2, out, count);
    count = enumerate_blocks(off + 3 * (m / 4), logm - 2, out, count);
  }
  return count;
}

static int cmp_block_desc(const void *a, const void *b) {
  const pko_fft_block_t *x = (const pko_fft_block_t *)a;
  const pko_fft_block_t *y = (const pko_fft_block_t *)b;
  if (x->logm != y->logm) return y->logm - x->logm;
  return x->off - y->off;
}

static int reverse_bits(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

int pko_srfft_init(pko_srfft_t *p, int n_real) {
  memset(p, 0, sizeof(*p));
  int n = n_real / 2;
  if (n <= 1 || (n & (n - 1)) != 0) return -1;   /* srfft.cc:343-347 */
  p->n_real = n_real;
  p->n_cplx = n;
  while ((1 << p->logn) < n) p->logn++;

  p->num_blocks = enumerate_blocks(0, p->logn, NULL, 0);
  p->blocks = (pko_fft_block_t *)malloc(sizeof(pko_fft_block_t) * p->num_blocks);
  enumerate_blocks(0, p->logn, p->blocks, 0);
  qsort(p->blocks, p->num_blocks, sizeof(pko_fft_block_t), cmp_block_desc);

  /* Butterfly coefficient tables, srfft.cc:64-91: the angle is formed in double
   * and rounded to float BEFORE the float cos/sin; the three combinations
   * c, -(s+c), s-c are float expressions.                                     */
  for (int logm = 4; logm <= p->logn; ++logm) {
    int m = 1 << logm, m4 = m / 4, m8 = m / 8;
    float *t = (float *)calloc((size_t)6 * m4, sizeof(float));
    p->tw[logm] = t;
    for (int n1 = 1; n1 < m4; ++n1) {
      if (n1 == m8) continue;
      float ang = n1 * TWO_PI_FULL / m;
      float c = cosf(ang), s = sinf(ang);
      t[0 * m4 + n1] = c;
      t[1 * m4 + n1] = -(s + c);
      t[2 * m4 + n1] = s - c;
      ang = 3 * n1 * TWO_PI_FULL / m;
      c = cosf(ang); s = sinf(ang);
      t[3 * m4 + n1] = c;
      t[4 * m4 + n1] = -(s + c);
      t[5 * m4 + n1] = s - c;
    }
  }

  /* srfft.cc:239-265 (Evans' seed-table unshuffle) realises the plain
   * bit-reversal permutation on logn bits; checked against oracle/_ref.       */
  p->bitrev = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) p->bitrev[i] = reverse_bits(i, p->logn);

  /* Real post-pass twiddles, srfft.cc:384-394: exp(-2*pi*i*k/N) is produced by
   * repeated FLOAT complex multiplication by the k=1 root.  The chain does not
   * depend on the data, so it is tabulated once with the same arithmetic.     */
  int half = n / 2;
  p->post_re = (float *)malloc(sizeof(float) * (half + 1));
  p->post_im = (float *)malloc(sizeof(float) * (half + 1));
  float ang1 = (float)(TWO_PI_FULL / n_real * -1);
  float root_re = cosf(ang1), root_im = sinf(ang1);
  float kre = 1.0f, kim = 0.0f;
  p->post_re[0] = kre; p->post_im[0] = kim;
  for (int k = 1; k <= half; ++k) {
    float t = (kre * root_re) - (kim * root_im);
    kim = kre * root_im + kim * root_re;
    kre = t;
    p->post_re[k] = kre; p->post_im[k] = kim;
  }
  return 0;
}

void pko_srfft_free(pko_srfft_t *p) {
  free(p->blocks);
  for (int i = 0; i < 32; ++i) free(p->tw[i]);
  free(p->bitrev);
  free(p->post_re);
  free(p->post_im);
  memset(p, 0, sizeof(*p));
}

/* One "L-shaped" split-radix butterfly on the four points n, n+m/4, n+m/2,
 * n+3m/4 of a block: srfft.cc:163-173 (step 1), :176-188 (step 2) and
 * :198-222 (steps 3 and 4) applied to the same quad.  The reference runs each
 * step as its own loop over the block; the quads are independent, so doing the
 * three steps quad by quad rounds identically.                                */
static void l_butterfly(float *xr, float *xi, int base, int n, int m,
                        const float *tw) {
  int m2 = m / 2, m4 = m / 4, m8 = m / 8;
  int i0 = base + n, i1 = i0 + m4, i2 = i0 + m2, i3 = i2 + m4;

  /* step 1: radix-2 sums/differences across the two halves */
  float ar = xr[i0] + xr[i2], br = xr[i0] - xr[i2];
  float ai = xi[i0] + xi[i2], bi = xi[i0] - xi[i2];
  float cr = xr[i1] + xr[i3], dr = xr[i1] - xr[i3];
  float ci = xi[i1] + xi[i3], di = xi[i1] - xi[i3];
  xr[i0] = ar; xi[i0] = ai; xr[i1] = cr; xi[i1] = ci;

  /* step 2 on the upper half: (b, d) -> b -/+ j d, with the reference's
   * particular placement of real and imaginary parts                          */
  float r1 = br + di;      /* new xr at i2 */
  float q2 = bi + dr;      /* new xi at i3 */
  float q1 = bi - dr;      /* new xi at i2 */
  float r2 = br - di;      /* new xr at i3 */

  /* steps 3 & 4: twiddles (none for n == 0) */
  if (n != 0) {
    if (n == m8) {
      const float sq = SQRT_HALF;
      float t1 = sq * (r1 + q1);
      q1 = sq * (q1 - r1);
      r1 = t1;
      float t2 = sq * (q2 - r2);
      q2 = -sq * (r2 + q2);
      r2 = t2;
    } else {
      const float cn = tw[0 * m4 + n], spcn = tw[1 * m4 + n], smcn = tw[2 * m4 + n];
      const float c3n = tw[3 * m4 + n], spc3n = tw[4 * m4 + n], smc3n = tw[5 * m4 + n];
      float t2 = cn * (r1 + q1);
      float t1 = spcn * r1 + t2;
      r1 = smcn * q1 + t2;
      q1 = t1;
      t2 = c3n * (r2 + q2);
      t1 = spc3n * r2 + t2;
      r2 = smc3n * q2 + t2;
      q2 = t1;
    }
  }
  xr[i2] = r1; xi[i2] = q1; xr[i3] = r2; xi[i3] = q2;
}

void pko_srfft_forward(const pko_srfft_t *p, float *data) {
  const int n = p->n_cplx, N = p->n_real;
  float xr[n], xi[n];

  /* de-interleave, srfft.cc:296-303 */
  for (int i = 0; i < n; ++i) { xr[i] = data[2 * i]; xi[i] = data[2 * i + 1]; }

  /* complex split-radix DIF, srfft.cc:95-237, as passes over the block list */
  for (int b = 0; b < p->num_blocks; ++b) {
    int off = p->blocks[b].off, logm = p->blocks[b].logm, m = 1 << logm;
    if (logm == 1) {                      /* srfft.cc:140-150 */
      float t = xr[off] + xr[off + 1];
      xr[off + 1] = xr[off] - xr[off + 1];
      xr[off] = t;
      t = xi[off] + xi[off + 1];
      xi[off + 1] = xi[off] - xi[off + 1];
      xi[off] = t;
    } else {
      for (int q = 0; q < m / 4; ++q) l_butterfly(xr, xi, off, q, m, p->tw[logm]);
    }
  }

  /* bit-reversal unshuffle (srfft.cc:239-265, :279-282) and re-interleave
   * (srfft.cc:306-316)                                                         */
  for (int i = 0; i < n; ++i) {
    int j = p->bitrev[i];
    data[2 * i] = xr[j];
    data[2 * i + 1] = xi[j];
  }

  /* real post-pass, srfft.cc:389-436.  0.5 is a double literal in the
   * reference; scaling by 0.5 is exact so float arithmetic gives the same.    */
  const int N2 = n;
  for (int k = 1; 2 * k <= N2; ++k) {
    float kre = p->post_re[k], kim = p->post_im[k];
    float ck_re = 0.5 * (data[2 * k] + data[N - 2 * k]);
    float ck_im = 0.5 * (data[2 * k + 1] - data[N - 2 * k + 1]);
    float dk_re = 0.5 * (data[2 * k + 1] + data[N - 2 * k + 1]);
    float dk_im = -0.5 * (data[2 * k] - data[N - 2 * k]);
    float out_re = ck_re, out_im = ck_im;
    out_re += kre * dk_re - kim * dk_im;
    out_im += kre * dk_im + kim * dk_re;
    data[2 * k] = out_re;
    data[2 * k + 1] = out_im;
    int kd = N2 - k;
    if (kd != k) {
      /* conjugate partner, srfft.cc:417-435: a = (dk_re, -dk_im), b = (-kre, kim) */
      float o_re = ck_re, o_im = -ck_im;
      float nk_re = -kre, ndk_im = -dk_im;
      o_re += nk_re * dk_re - kim * ndk_im;
      o_im += nk_re * ndk_im + kim * dk_re;
      data[2 * kd] = o_re;
      data[2 * kd + 1] = o_im;
    }
  }
  /* srfft.cc:444-447 */
  float zeroth = data[0] + data[1], n2th = data[0] - data[1];
  data[0] = zeroth;
  data[1] = n2th;
}

/* ------------------------------------------------------------------------- */
/* fbank                                                                      */
/* ------------------------------------------------------------------------- */

static float mel_scale(float freq) {            /* fbank.h:30-32 */
  return 1127.0f * logf(1.0f + freq / 700.0f);
}

int pko_fbank_init(pko_fbank_t *fb) {
  memset(fb, 0, sizeof(*fb));
  if (pko_srfft_init(&fb->fft, PKO_FFT_SIZE) != 0) return -1;

  /* Hamming window, fbank.cc:249-256 */
  float a = TWO_PI_TRUNC / (PKO_FRAME_LENGTH - 1);
  for (int i = 0; i < PKO_FRAME_LENGTH; ++i) {
    float i_fl = (float)i;
    fb->window[i] = 0.54 - 0.46 * cosf(a * i_fl);
  }

  /* mel triangles, fbank.cc:103-163 */
  float sample_freq = PKO_SAMPLE_RATE;
  int num_fft_bins = PKO_FFT_SIZE / 2;
  float fft_bin_width = sample_freq / PKO_FFT_SIZE;
  float mel_low = mel_scale(20);
  float mel_high = mel_scale(PKO_SAMPLE_RATE / 2);
  float mel_delta = (mel_high - mel_low) / (PKO_NUM_BINS + 1);
  for (int bin = 0; bin < PKO_NUM_BINS; ++bin) {
    float left = mel_low + bin * mel_delta;
    float center = mel_low + (bin + 1) * mel_delta;
    float right = mel_low + (bin + 2) * mel_delta;
    float w[PKO_FFT_SIZE / 2];
    int first = -1, last = -1;
    for (int i = 0; i < num_fft_bins; ++i) {
      float freq = fft_bin_width * i;
      float mel = mel_scale(freq);
      w[i] = 0.0f;
      if (mel > left && mel < right) {
        if (mel <= center) w[i] = (mel - left) / (center - left);
        else w[i] = (right - mel) / (right - center);
        if (first < 0) first = i;
        last = i;
      }
    }
    if (first < 0 || last <= first) return -2;
    fb->mel_offset[bin] = first;
    fb->mel_len[bin] = last + 1 - first;
    memcpy(fb->mel_weight[bin], w + first, sizeof(float) * fb->mel_len[bin]);
  }
  return 0;
}

void pko_fbank_free(pko_fbank_t *fb) { pko_srfft_free(&fb->fft); }

int pko_num_frames(int num_samples) {           /* fbank.cc:35-42 */
  if (num_samples < PKO_FRAME_LENGTH) return 0;
  return 1 + (num_samples - PKO_FRAME_LENGTH) / PKO_FRAME_SHIFT;
}

void pko_fbank_frame(const pko_fbank_t *fb, const float *samples, float *out40,
                     float *spec512) {
  float x[PKO_FFT_SIZE];
  memcpy(x, samples, sizeof(float) * PKO_FRAME_LENGTH);
  for (int i = PKO_FRAME_LENGTH; i < PKO_FFT_SIZE; ++i) x[i] = 0.0f;   /* fbank.cc:92-96 */

  /* DC removal, fbank.cc:48-55: sequential float sum, float divide */
  float sum = 0;
  for (int i = 0; i < PKO_FRAME_LENGTH; ++i) sum += x[i];
  float mean = sum / PKO_FRAME_LENGTH;
  for (int i = 0; i < PKO_FRAME_LENGTH; ++i) x[i] -= mean;

  /* pre-emphasis, fbank.cc:58-61: 0.97 is a double literal, so each step is
   * evaluated in double and rounded once to float                              */
  for (int i = PKO_FRAME_LENGTH - 1; i > 0; --i) x[i] -= 0.97 * x[i - 1];
  x[0] -= 0.97 * x[0];

  for (int i = 0; i < PKO_FRAME_LENGTH; ++i) x[i] *= fb->window[i];     /* fbank.cc:66-68 */

  pko_srfft_forward(&fb->fft, x);
  if (spec512) memcpy(spec512, x, sizeof(x));

  /* power spectrum, fbank.cc:193-211 */
  const int half = PKO_FFT_SIZE / 2;
  float first_energy = x[0] * x[0];
  float last_energy = x[1] * x[1];
  for (int i = 1; i < half; ++i) {
    float re = x[2 * i], im = x[2 * i + 1];
    x[i] = re * re + im * im;
  }
  x[0] = first_energy;
  x[half] = last_energy;

  /* mel energies (fbank.cc:165-184 -> vector.cc:252-262), floor and log
   * (fbank.cc:244-245 -> vector.cc:322-339)                                    */
  for (int b = 0; b < PKO_NUM_BINS; ++b) {
    const float *w = fb->mel_weight[b];
    const float *ps = x + fb->mel_offset[b];
    float e = 0.0;
    for (int j = 0; j < fb->mel_len[b]; ++j) e += w[j] * ps[j];
    if (e < FLT_EPSILON) e = FLT_EPSILON;
    out40[b] = logf(e);
  }
}

void pko_fbank_compute(const pko_fbank_t *fb, const float *wave, int num_samples,
                       float *out) {
  int T = pko_num_frames(num_samples);
  for (int t = 0; t < T; ++t)                    /* fbank.cc:281-291 */
    pko_fbank_frame(fb, wave + (size_t)t * PKO_FRAME_SHIFT, out + (size_t)t * PKO_NUM_BINS,
                    NULL);
}

/* ------------------------------------------------------------------------- */
/* CMVN                                                                       */
/* ------------------------------------------------------------------------- */

void pko_cmvn(const float *g, const float *raw, int T, float *out) {
  const int D = PKO_NUM_BINS;
  float cached[PKO_NUM_BINS + 1];
  for (int i = 0; i <= D; ++i) cached[i] = 0.0f;
  for (int t = 0; t < T; ++t) {
    const float *x = raw + (size_t)t * D;
    /* cmvn.cc:44-70: the running window sum goes through a double temporary
     * and is stored back to float every frame                                  */
    double acc[PKO_NUM_BINS + 1];
    for (int i = 0; i <= D; ++i) acc[i] = cached[i];
    for (int i = 0; i < D; ++i) acc[i] += x[i];
    acc[D] += 1.0;
    int prev = t - PKO_CMVN_WINDOW;
    if (prev >= 0) {
      const float *xp = raw + (size_t)prev * D;
      for (int i = 0; i < D; ++i) acc[i] += -1.0 * xp[i];
      acc[D] -= 1.0;
    }
    float stats[PKO_NUM_BINS + 1];
    for (int i = 0; i <= D; ++i) { stats[i] = (float)acc[i]; cached[i] = stats[i]; }

    /* cmvn.cc:73-92: smoothing with the global prior, float axpy with a scalar
     * that was computed in double and narrowed                                 */
    double count = stats[D];
    if (count < PKO_CMVN_WINDOW) {
      double from_global = PKO_CMVN_WINDOW - count;
      double global_count = g[D];
      if (from_global > PKO_CMVN_GLOBAL_FRAMES) from_global = PKO_CMVN_GLOBAL_FRAMES;
      float alpha = (float)(from_global / global_count);
      for (int i = 0; i <= D; ++i) stats[i] += alpha * g[i];
    }

    /* cmvn.cc:94-101 */
    double cnt = stats[D];
    float scale = 1 / cnt;
    float ns = -scale;
    float *y = out + (size_t)t * D;
    for (int i = 0; i < D; ++i) { float v = x[i]; v += ns * stats[i]; y[i] = v; }
  }
}

/* ------------------------------------------------------------------------- */
/* splice                                                                     */
/* ------------------------------------------------------------------------- */

void pko_logf_array(const float *x, int n, float *out) {
  for (int i = 0; i < n; ++i) out[i] = logf(x[i]);   /* vector.cc:336-338 */
}

void pko_splice(const float *feats, int T, int dim, int left, int right, float *out) {
  int width = (left + right + 1) * dim;           /* am.cc:65-88 */
  for (int t = 0; t < T; ++t) {
    float *dst = out + (size_t)t * width;
    for (int f = -left; f <= right; ++f) {
      int src = t + f;
      if (src < 0) src = 0;
      if (src >= T) src = T - 1;
      memcpy(dst, feats + (size_t)src * dim, sizeof(float) * dim);
      dst += dim;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* SGEMM, naive form (the blocked AVX2 form lives in pk_oracle_gemm_avx2.c)    */
/* ------------------------------------------------------------------------- */

/* gemm.cc:95-123 walks k in chunks of KC=512; inside a chunk the micro-kernel
 * (gemm_haswell.cc:122-282) keeps one accumulator per output element and issues
 * one vfmadd231ps per k, ascending, starting from zero; the first chunk is
 * stored (beta=0), later chunks are added to C with one float add.            */
void pko_sgemm_naive(int m, int n, int k, const float *A, int lda, const float *B,
                     int ldb, float *C, int ldc) {
  const int KC = 512;
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) {
      float total = 0.0f;
      for (int k0 = 0; k0 < k; k0 += KC) {
        int k1 = k0 + KC < k ? k0 + KC : k;
        float acc = 0.0f;
        for (int kk = k0; kk < k1; ++kk)
          acc = fmaf(A[(size_t)i * lda + kk], B[(size_t)kk * ldb + j], acc);
        total = (k0 == 0) ? acc : total + acc;
      }
      C[(size_t)i * ldc + j] = total;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* layers                                                                     */
/* ------------------------------------------------------------------------- */

void pko_nnet_init(pko_nnet_t *nn) { nn->num_layers = 0; nn->layers = NULL; }

void pko_nnet_free(pko_nnet_t *nn) {
  for (int i = 0; i < nn->num_layers; ++i) { free(nn->layers[i].Wt); free(nn->layers[i].b); }
  free(nn->layers);
  nn->layers = NULL;
  nn->num_layers = 0;
}

static pko_layer_t *push_layer(pko_nnet_t *nn) {
  nn->layers = (pko_layer_t *)realloc(nn->layers, sizeof(pko_layer_t) * (nn->num_layers + 1));
  pko_layer_t *l = &nn->layers[nn->num_layers++];
  memset(l, 0, sizeof(*l));
  return l;
}

int pko_nnet_add_linear(pko_nnet_t *nn, int in_dim, int out_dim, const float *W,
                        const float *b) {
  pko_layer_t *l = push_layer(nn);
  l->type = PKO_LINEAR;
  l->in_dim = in_dim;
  l->out_dim = out_dim;
  l->Wt = (float *)malloc(sizeof(float) * (size_t)in_dim * out_dim);
  l->b = (float *)malloc(sizeof(float) * out_dim);
  for (int o = 0; o < out_dim; ++o)             /* nnet.cc:16-17: keep W transposed */
    for (int i = 0; i < in_dim; ++i) l->Wt[(size_t)i * out_dim + o] = W[(size_t)o * in_dim + i];
  memcpy(l->b, b, sizeof(float) * out_dim);
  return 0;
}

int pko_nnet_add_simple(pko_nnet_t *nn, int type) {
  if (type != PKO_RELU && type != PKO_NORMALIZE && type != PKO_SOFTMAX) return -1;
  push_layer(nn)->type = type;
  return 0;
}

void pko_relu(float *x, int64_t n) {            /* nnet.cc:49-60 */
  for (int64_t i = 0; i < n; ++i) if (x[i] < 0.0f) x[i] = 0.0f;
}

void pko_normalize_rows(float *x, int T, int dim) {   /* nnet.cc:62-75 */
  float D = dim;
  for (int t = 0; t < T; ++t) {
    float *row = x + (size_t)t * dim;
    float ss = 0.0;
    for (int i = 0; i < dim; ++i) ss += row[i] * row[i];
    double squared_sum = ss;
    float scale = (float)sqrt(D / squared_sum);
    for (int i = 0; i < dim; ++i) row[i] *= scale;
  }
}

void pko_softmax_rows(float *x, int T, int dim) {     /* nnet.cc:38-47, vector.cc:265-277 */
  for (int t = 0; t < T; ++t) {
    float *row = x + (size_t)t * dim;
    float sum = 0;
    for (int i = 0; i < dim; ++i) { float e = expf(row[i]); row[i] = e; sum += e; }
    for (int i = 0; i < dim; ++i) row[i] /= sum;
  }
}

int pko_nnet_output_dim(const pko_nnet_t *nn, int in_dim) {
  int d = in_dim;
  for (int i = 0; i < nn->num_layers; ++i)
    if (nn->layers[i].type == PKO_LINEAR) {
      if (nn->layers[i].in_dim != d) return -1;
      d = nn->layers[i].out_dim;
    }
  return d;
}

int pko_nnet_propagate(const pko_nnet_t *nn, const float *in, int T, int in_dim,
                       float *out, int out_cap_per_row) {
  int out_dim = pko_nnet_output_dim(nn, in_dim);
  if (out_dim < 0 || out_dim > out_cap_per_row) return -1;
  int cur_dim = in_dim;
  float *cur = (float *)malloc(sizeof(float) * (size_t)T * in_dim);
  memcpy(cur, in, sizeof(float) * (size_t)T * in_dim);
  for (int li = 0; li < nn->num_layers; ++li) {       /* nnet.cc:149-163 */
    const pko_layer_t *l = &nn->layers[li];
    switch (l->type) {
      case PKO_LINEAR: {                              /* nnet.cc:22-36 */
        float *next = (float *)malloc(sizeof(float) * (size_t)T * l->out_dim);
        pko_sgemm(T, l->out_dim, l->in_dim, cur, l->in_dim, l->Wt, l->out_dim, next,
                  l->out_dim);
        for (int t = 0; t < T; ++t)
          for (int o = 0; o < l->out_dim; ++o) next[(size_t)t * l->out_dim + o] += l->b[o];
        free(cur);
        cur = next;
        cur_dim = l->out_dim;
        break;
      }
      case PKO_RELU: pko_relu(cur, (int64_t)T * cur_dim); break;
      case PKO_NORMALIZE: pko_normalize_rows(cur, T, cur_dim); break;
      case PKO_SOFTMAX: pko_softmax_rows(cur, T, cur_dim); break;
      default: free(cur); return -2;
    }
  }
  memcpy(out, cur, sizeof(float) * (size_t)T * cur_dim);
  free(cur);
  return cur_dim;
}

/* ------------------------------------------------------------------------- */
/* acoustic-model tail + decodable scale                                      */
/* ------------------------------------------------------------------------- */

int pko_am_compute(const pko_nnet_t *nn, const float *prior, int num_pdfs, int left,
                   int right, const float *feats, int T, int feat_dim,
                   float prob_scale, float *loglik) {
  int width = (left + right + 1) * feat_dim;
  float *spliced = (float *)malloc(sizeof(float) * (size_t)T * width);
  pko_splice(feats, T, feat_dim, left, right, spliced);
  int od = pko_nnet_propagate(nn, spliced, T, width, loglik, num_pdfs);
  free(spliced);
  if (od != num_pdfs) return -1;

  float *log_prior = (float *)malloc(sizeof(float) * num_pdfs);
  for (int i = 0; i < num_pdfs; ++i) log_prior[i] = logf(prior[i]);   /* am.cc:41-44 */
  const float floor_val = 1.0e-20;                                    /* am.cc:109 */
  for (int t = 0; t < T; ++t) {                                       /* am.cc:106-112 */
    float *row = loglik + (size_t)t * num_pdfs;
    for (int i = 0; i < num_pdfs; ++i) {
      float v = row[i];
      if (v < floor_val) v = floor_val;
      v = logf(v);
      v += -1.0f * log_prior[i];
      row[i] = v;
    }
  }
  /* decodable.cc:15 -> matrix.cc:99-103 */
  for (size_t i = 0; i < (size_t)T * num_pdfs; ++i) loglik[i] *= prob_scale;
  free(log_prior);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* WAV + model-file readers                                                   */
/* ------------------------------------------------------------------------- */

static int32_t rd_i32(const unsigned char *p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
                   ((uint32_t)p[3] << 24));
}
static int16_t rd_i16(const unsigned char *p) { return (int16_t)(p[0] | (p[1] << 8)); }

static unsigned char *slurp(const char *path, long *size) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  *size = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc(*size > 0 ? *size : 1);
  if (fread(buf, 1, *size, f) != (size_t)*size) { free(buf); fclose(f); return NULL; }
  fclose(f);
  return buf;
}

int pko_wav_read(const char *path, float **samples_out) {   /* pcm_reader.cc:45-220 */
  long size = 0;
  unsigned char *b = slurp(path, &size);
  if (!b) return -1;
  int rc = -2;
  if (size < 44) goto done;
  if (memcmp(b, "RIFF", 4) || rd_i32(b + 4) != size - 8) goto done;
  if (memcmp(b + 8, "WAVE", 4) || memcmp(b + 12, "fmt ", 4)) goto done;
  if (rd_i32(b + 16) != 16 || rd_i16(b + 20) != 1 || rd_i16(b + 22) != 1) goto done;
  int rate = rd_i32(b + 24);
  if (rate != PKO_SAMPLE_RATE) goto done;
  int byte_rate = rd_i32(b + 28), align = rd_i16(b + 32), bits = rd_i16(b + 34);
  if (bits != 8 && bits != 16 && bits != 32) goto done;
  if (byte_rate != rate * bits / 8 || align != bits / 8) goto done;
  if (memcmp(b + 36, "data", 4) || rd_i32(b + 40) != size - 44) goto done;
  {
    int n = (int)(size - 44) / (bits / 8);
    float *s = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    const unsigned char *p = b + 44;
    for (int i = 0; i < n; ++i) {               /* unscaled integer -> float */
      if (bits == 8) { s[i] = (float)(int8_t)p[0]; p += 1; }
      else if (bits == 16) { s[i] = (float)rd_i16(p); p += 2; }
      else { s[i] = (float)rd_i32(p); p += 4; }
    }
    *samples_out = s;
    rc = n;
  }
done:
  free(b);
  return rc;
}

/* "VEC0" i32 size(=4n+4) i32 n, n x f32 -- vector.cc:393-425 */
static long parse_vec(const unsigned char *b, long size, long pos, float **out, int *dim) {
  if (pos + 12 > size || memcmp(b + pos, "VEC0", 4)) return -1;
  int32_t sec = rd_i32(b + pos + 4), n = rd_i32(b + pos + 8);
  if (n < 0 || sec != n * 4 + 4 || pos + 12 + (long)n * 4 > size) return -1;
  float *v = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
  memcpy(v, b + pos + 12, (size_t)n * 4);
  *out = v;
  *dim = n;
  return pos + 12 + (long)n * 4;
}

int pko_read_vec_f32(const char *path, float **out, int *dim) {
  long size = 0;
  unsigned char *b = slurp(path, &size);
  if (!b) return -1;
  long end = parse_vec(b, size, 0, out, dim);
  free(b);
  return end < 0 ? -2 : 0;
}

int pko_nnet_read(pko_nnet_t *nn, const char *path) {    /* nnet.cc:80-147 */
  long size = 0;
  unsigned char *b = slurp(path, &size);
  if (!b) return -1;
  int rc = -2;
  pko_nnet_init(nn);
  if (size < 12 || memcmp(b, "NNT0", 4) || rd_i32(b + 4) != 4) goto done;
  int num_layers = rd_i32(b + 8);
  long pos = 12;
  for (int li = 0; li < num_layers; ++li) {
    if (pos + 12 > size || memcmp(b + pos, "LAY0", 4) || rd_i32(b + pos + 4) != 4) goto done;
    int type = rd_i32(b + pos + 8);
    pos += 12;
    if (type == PKO_LINEAR) {
      /* "MAT0" i32 8, i32 rows, i32 cols, rows x VEC0 -- matrix.cc:288-319 */
      if (pos + 16 > size || memcmp(b + pos, "MAT0", 4)) goto done;
      int rows = rd_i32(b + pos + 8), cols = rd_i32(b + pos + 12);
      pos += 16;
      float *W = (float *)malloc(sizeof(float) * (size_t)rows * cols);
      for (int r = 0; r < rows; ++r) {
        float *row; int d;
        pos = parse_vec(b, size, pos, &row, &d);
        if (pos < 0 || d != cols) { free(W); goto done; }
        memcpy(W + (size_t)r * cols, row, sizeof(float) * cols);
        free(row);
      }
      float *bias; int bd;
      pos = parse_vec(b, size, pos, &bias, &bd);
      if (pos < 0 || bd != rows) { free(W); goto done; }
      pko_nnet_add_linear(nn, cols, rows, W, bias);
      free(W);
      free(bias);
    } else if (pko_nnet_add_simple(nn, type) != 0) {
      goto done;
    }
  }
  rc = 0;
done:
  free(b);
  if (rc != 0) pko_nnet_free(nn);
  return rc;
}
