/*
 * pk_oracle_gemm_avx2.c -- blocked AVX2/FMA form of pko_sgemm().
 * TEST INFRASTRUCTURE ONLY (see pk_oracle.h).
 *
 * Same algorithmic class as the reference's CPU path (gemm.cc:70-125 driver,
 * gemm.cc:186-252 packing, gemm_haswell.cc:72-621 6x16 micro-kernel): packed
 * panels, cache blocking, a 6x16 register tile of twelve 8-float accumulators,
 * one fused multiply-add per (element, k), k ascending, chunks of KC = 512
 * combined through C.  Written with intrinsics, single thread.  It exists so
 * that (i) the oracle finishes parity-size problems in seconds and (ii) the
 * cpu_baseline leg of bench.py times a CPU path of the reference's own class.
 * Its results are bit-identical to pko_sgemm_naive() and to the reference's
 * GEMM<float>::Gemm (tests/test_oracle_ref.py).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -mavx2 -mfma (this file only).
 */
#define _POSIX_C_SOURCE 200809L
#include "pk_oracle.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

enum { MR = 6, NR = 16, KC = 512, MC = 288, NC = 4096 };

/* a: packed MR x kc (k-major, MR contiguous), b: packed kc x NR.  acc from zero. */
static void micro_6x16(int kc, const float *a, const float *b, float *out /* MR*NR */) {
  __m256 c[MR][2];
  for (int i = 0; i < MR; ++i) { c[i][0] = _mm256_setzero_ps(); c[i][1] = _mm256_setzero_ps(); }
  for (int k = 0; k < kc; ++k) {
    __m256 b0 = _mm256_load_ps(b), b1 = _mm256_load_ps(b + 8);
    for (int i = 0; i < MR; ++i) {
      __m256 ai = _mm256_broadcast_ss(a + i);
      c[i][0] = _mm256_fmadd_ps(ai, b0, c[i][0]);
      c[i][1] = _mm256_fmadd_ps(ai, b1, c[i][1]);
    }
    a += MR;
    b += NR;
  }
  for (int i = 0; i < MR; ++i) {
    _mm256_storeu_ps(out + i * NR, c[i][0]);
    _mm256_storeu_ps(out + i * NR + 8, c[i][1]);
  }
}

void pko_sgemm(int m, int n, int k, const float *A, int lda, const float *B, int ldb,
               float *C, int ldc) {
  if (m <= 0 || n <= 0) return;
  if (k <= 0) {
    for (int i = 0; i < m; ++i) memset(C + (size_t)i * ldc, 0, sizeof(float) * n);
    return;
  }
  float *pa = NULL, *pb = NULL;
  if (posix_memalign((void **)&pa, 64, sizeof(float) * (MC + MR) * KC) ||
      posix_memalign((void **)&pb, 64, sizeof(float) * KC * (size_t)(NC + NR)))
    abort();
  float tile[MR * NR];

  for (int jc = 0; jc < n; jc += NC) {
    int nc = n - jc < NC ? n - jc : NC;
    for (int pc = 0; pc < k; pc += KC) {
      int kc = k - pc < KC ? k - pc : KC;
      int first_chunk = (pc == 0);
      /* pack B[pc:pc+kc][jc:jc+nc] into NR-wide panels, zero padded */
      for (int jr = 0; jr < nc; jr += NR) {
        int nr = nc - jr < NR ? nc - jr : NR;
        float *dst = pb + (size_t)(jr / NR) * kc * NR;
        for (int kk = 0; kk < kc; ++kk) {
          const float *src = B + (size_t)(pc + kk) * ldb + jc + jr;
          for (int j = 0; j < nr; ++j) dst[kk * NR + j] = src[j];
          for (int j = nr; j < NR; ++j) dst[kk * NR + j] = 0.0f;
        }
      }
      for (int ic = 0; ic < m; ic += MC) {
        int mc = m - ic < MC ? m - ic : MC;
        /* pack A[ic:ic+mc][pc:pc+kc] into MR-tall panels, zero padded */
        for (int ir = 0; ir < mc; ir += MR) {
          int mr = mc - ir < MR ? mc - ir : MR;
          float *dst = pa + (size_t)(ir / MR) * kc * MR;
          for (int kk = 0; kk < kc; ++kk) {
            for (int i = 0; i < mr; ++i)
              dst[kk * MR + i] = A[(size_t)(ic + ir + i) * lda + pc + kk];
            for (int i = mr; i < MR; ++i) dst[kk * MR + i] = 0.0f;
          }
        }
        for (int jr = 0; jr < nc; jr += NR) {
          int nr = nc - jr < NR ? nc - jr : NR;
          const float *bp = pb + (size_t)(jr / NR) * kc * NR;
          for (int ir = 0; ir < mc; ir += MR) {
            int mr = mc - ir < MR ? mc - ir : MR;
            micro_6x16(kc, pa + (size_t)(ir / MR) * kc * MR, bp, tile);
            float *cdst = C + (size_t)(ic + ir) * ldc + jc + jr;
            if (first_chunk) {
              for (int i = 0; i < mr; ++i)
                for (int j = 0; j < nr; ++j) cdst[(size_t)i * ldc + j] = tile[i * NR + j];
            } else {
              for (int i = 0; i < mr; ++i)
                for (int j = 0; j < nr; ++j) cdst[(size_t)i * ldc + j] += tile[i * NR + j];
            }
          }
        }
      }
    }
  }
  free(pa);
  free(pb);
}
