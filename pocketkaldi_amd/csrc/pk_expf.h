// pk_expf.h -- expf as the C library the reference links against computes it.
//
// SoftmaxLayer (nnet.cc:38-47 -> vector.cc:265-277) exponentiates with libm's expf.  glibc's
// expf (2.27 and later: sysdeps/ieee754/flt-32/e_expf.c, the exp2f-table algorithm of the ARM
// optimized-routines) is restated here -- x N/ln2 = k + r, 2^(k/N) from a 32-entry table,
// 2^(r/N) as a cubic in double -- with the fused multiply-adds of the x86-64 FMA build of the
// library (the one that runs on every host with FMA): r = fma(N/ln2, x, -k) and a fused
// polynomial.  tools/expf_check.c compares it with the system expf over every non-NaN float
// (4 278 190 082 values, no mismatch); tests/cpp/libm_restated_test.cc repeats a strided sweep
// in the CPU suite.  Used by the reference-exact softmax tail only; the default tail is the
// overflow-safe log-softmax.
#ifndef PK_EXPF_H_
#define PK_EXPF_H_

#include <stdint.h>

#ifdef __HIPCC__
#define PK_EXPF_FN __host__ __device__ inline
#else
#define PK_EXPF_FN inline
#endif

namespace pkmi {

constexpr int kExpfTableWords = 32;

// glibc's __exp2f_data.tab: asuint64(2^(i/32)) - (i << 47), i = 0..31 (identical to the
// table in this image's libm.so.6).
#define PK_EXPF_TABLE_INIT                                                                  \
  {                                                                                         \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,   \
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,   \
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,   \
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,   \
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,   \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,   \
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,   \
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull   \
  }

// tab = the 32 words above, in whatever memory is close.
PK_EXPF_FN float ExpfRestated(float x, const uint64_t *tab) {
  const uint32_t ix = __builtin_bit_cast(uint32_t, x);
  const uint32_t abstop = (ix >> 20) & 0x7ff;
  if (abstop >= 0x42b) {                                   // |x| >= 88 or NaN (top12(88.0f) = 0x42b)
    if (ix == 0xff800000u) return 0.0f;                    // exp(-inf)
    if (abstop >= 0x7f8) return x + x;                     // +inf, NaN
    if (x > 0x1.62e42ep6f) return __builtin_inff();        // overflow, x > log(2^128)
    if (x < -0x1.9fe368p6f) return 0.0f;                   // underflow, x < log(2^-150)
  }
  const double xd = (double)x;
  const double z = 0x1.71547652b82fep+5 * xd;              // x N / ln2
  double kd = z + 0x1.8p+52;                               // round to integer, keep it in the low bits
  const uint64_t ki = __builtin_bit_cast(uint64_t, kd);
  kd -= 0x1.8p+52;
  const double r = __builtin_fma(0x1.71547652b82fep+5, xd, -kd);
  const uint64_t t = tab[ki & 31] + (ki << 47);
  const double s = __builtin_bit_cast(double, t);
  const double q = __builtin_fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.62e42ff0c52d6p-6, r, 1.0);
  y = __builtin_fma(q, r2, y);
  y = y * s;
  return (float)y;
}

#ifdef __HIPCC__
// The same function for a wavefront: the special cases (|x| >= 88, infinities, NaN) are tested once
// for the whole wave and patched in with selects, so that the common case is straight-line code (as
// separate per-lane branches hipcc moves the arithmetic out of line behind an exec-mask branch per
// element).  Same results as ExpfRestated for every input: the arithmetic below is its main path,
// and the selects are its early returns in the same priority.
__device__ __forceinline__ float ExpfRestatedWave(float x, const uint64_t *tab) {
  const uint32_t ix = __builtin_bit_cast(uint32_t, x);
  const uint32_t abstop = (ix >> 20) & 0x7ff;
  const double xd = (double)x;
  const double z = 0x1.71547652b82fep+5 * xd;
  double kd = z + 0x1.8p+52;
  const uint64_t ki = __builtin_bit_cast(uint64_t, kd);
  kd -= 0x1.8p+52;
  const double r = __builtin_fma(0x1.71547652b82fep+5, xd, -kd);
  const uint64_t t = tab[ki & 31] + (ki << 47);
  const double s = __builtin_bit_cast(double, t);
  const double q = __builtin_fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.62e42ff0c52d6p-6, r, 1.0);
  y = __builtin_fma(q, r2, y);
  y = y * s;
  float out = (float)y;
  if (__builtin_amdgcn_ballot_w64(abstop >= 0x42b) != 0) {      // some lane of the wave is special
    out = x < -0x1.9fe368p6f ? 0.0f : out;                        // underflow
    out = x > 0x1.62e42ep6f ? __builtin_inff() : out;             // overflow
    out = abstop >= 0x7f8 ? x + x : out;                          // +inf, NaN
    out = ix == 0xff800000u ? 0.0f : out;                         // exp(-inf)
  }
  return out;
}
#endif

}  // namespace pkmi

#endif  // PK_EXPF_H_
