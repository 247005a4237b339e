// ref_shim.cc -- extern "C" entry points over the parts of the reference that
// build in this image from their own sources (no stand-in headers or libraries):
//   /root/reference/src/srfft.cc        (pk_srfft_init / pk_srfft_compute, already extern "C")
//   /root/reference/src/gemm.cc         (GEMM<float> driver)
//   /root/reference/src/gemm_haswell.cc (AVX2/FMA 6x16 micro-kernel)
// TEST INFRASTRUCTURE ONLY.  The reference sources are compiled where they lie;
// nothing of them is copied into this repository.  Output: oracle/_ref/libpkref.so.
//
// The reference's vector.cc / matrix.cc include <cblas.h>, which this image does
// not have, so fbank.cc, cmvn.cc, nnet.cc, am.cc and decodable.cc cannot be
// linked here; those stages are pinned by the reference's fixtures instead.
#include "gemm.h"
#include "srfft.h"

#include <vector>

extern "C" {

// Same call as MatMat (matrix.cc:418-436): row-major operands, alpha=1, beta=0.
void pkref_sgemm(int m, int n, int k, const float *A, int lda, const float *B, int ldb,
                 float *C, int ldc) {
  pocketkaldi::GEMM<float> sgemm;  // constructed per call, as nnet.cc:28 does
  sgemm.Gemm(m, n, k, 1.0f, A, lda, 1, B, ldb, 1, 0.0f, C, ldc, 1);
}

// In-place forward real FFT of n_real points (fbank.cc:230-236 usage).
void pkref_srfft_forward(float *data, int n_real) {
  pk_srfft_t fft;
  pk_srfft_init(&fft, n_real);
  std::vector<float> buffer(n_real);
  pk_srfft_compute(&fft, data, n_real, true, buffer.data(), n_real);
  pk_srfft_destroy(&fft);
}

}  // extern "C"
