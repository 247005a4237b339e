// pk_logf.h -- logf as the C library the reference links against computes it.
//
// The reference takes the log of the mel energies with libm's logf (fbank.cc:244-245 ->
// vector.cc:334-339).  glibc's logf (2.28 and later: sysdeps/ieee754/flt-32/e_logf.c, the
// table-driven algorithm of the ARM optimized-routines) is not correctly rounded, so a
// double-precision log rounded to float differs from it in the last bit now and then.  This is
// a restatement of that algorithm -- x = 2^k z, 16 sub-intervals of [0x1.66p-1, 0x1.66p0),
// log(x) = log1p(z/c - 1) + log(c) + k ln2 with a cubic in double -- so that the GPU features
// are the reference's bit for bit.  tools/logf_check.c compares it with the system logf over
// every positive finite float (2 139 095 039 values, no mismatch, with and without fused
// multiply-adds); tests/cpp/libm_restated_test.cc repeats a strided
// sweep in the CPU suite.
#ifndef PK_LOGF_H_
#define PK_LOGF_H_

#include <stdint.h>

#ifdef __HIPCC__
#define PK_LOGF_FN __host__ __device__ inline
#else
#define PK_LOGF_FN inline
#endif

namespace pkmi {

constexpr int kLogfTableDoubles = 32;   // 16 x (1/c, log c)

// The constants of glibc's __logf_data (logf_data.c): { 1/c_i, log c_i }, i = 0..15.
#define PK_LOGF_TABLE_INIT                                  \
  {                                                         \
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2,            \
    0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,            \
    0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2,             \
    0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,            \
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3,            \
    0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,               \
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4,            \
    0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,            \
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5,            \
    0x1p+0, 0x0p+0,                                         \
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,             \
    0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,              \
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,             \
    0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,              \
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,             \
    0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2              \
  }

// x must be a positive normal float, +inf or NaN (the callers floor at FLT_EPSILON first);
// tab = the 32 doubles above, in whatever memory is close.
PK_LOGF_FN float LogfRestated(float x, const double *tab) {
  uint32_t ix = __builtin_bit_cast(uint32_t, x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix >= 0x7f800000u) return x;                       // log(inf) = inf, NaN stays NaN
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = tab[2 * i], logc = tab[2 * i + 1];
  const double z = (double)__builtin_bit_cast(float, iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return (float)y;
}

#ifdef __HIPCC__
// The same function for a wavefront (see ExpfRestatedWave): x = 1, infinities and NaN are tested once
// per wave and patched in with selects.
__device__ __forceinline__ float LogfRestatedWave(float x, const double *tab) {
  const uint32_t ix = __builtin_bit_cast(uint32_t, x);
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = tab[2 * i], logc = tab[2 * i + 1];
  const double z = (double)__builtin_bit_cast(float, iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  float out = (float)y;
  if (__builtin_amdgcn_ballot_w64(ix == 0x3f800000u || ix >= 0x7f800000u) != 0) {
    out = ix == 0x3f800000u ? 0.0f : out;
    out = ix >= 0x7f800000u ? x : out;
  }
  return out;
}
#endif

}  // namespace pkmi

#endif  // PK_LOGF_H_
