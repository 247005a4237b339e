// CPU check of csrc/pk_logf.h and csrc/pk_expf.h (the logf / expf the kernels use) against the C
// library the reference links: strided sweeps over the floats, dense runs around 1.0 / 0.0, the
// special values.  The exhaustive sweeps are tools/logf_check.c and tools/expf_check.c.
// Built and run by tests/test_cpp_host.py.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../pocketkaldi_amd/csrc/pk_expf.h"
#include "../../pocketkaldi_amd/csrc/pk_logf.h"

static uint32_t Bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float FromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
  static const double ltab[pkmi::kLogfTableDoubles] = PK_LOGF_TABLE_INIT;
  static const uint64_t etab[pkmi::kExpfTableWords] = PK_EXPF_TABLE_INIT;
  long bad = 0, n = 0;
  auto check_log = [&](uint32_t u) {
    const float x = FromBits(u);
    if (Bits(logf(x)) != Bits(pkmi::LogfRestated(x, ltab))) {
      if (bad < 5) fprintf(stderr, "logf x = %a: libm %a, restated %a\n", x, logf(x), pkmi::LogfRestated(x, ltab));
      ++bad;
    }
    ++n;
  };
  auto check_exp = [&](uint32_t u) {
    const float x = FromBits(u);
    if (x != x) return;
    if (Bits(expf(x)) != Bits(pkmi::ExpfRestated(x, etab))) {
      if (bad < 5) fprintf(stderr, "expf x = %a: libm %a, restated %a\n", x, expf(x), pkmi::ExpfRestated(x, etab));
      ++bad;
    }
    ++n;
  };
  for (uint64_t u = 0x00800000u; u < 0x7f800000u; u += 61) check_log((uint32_t)u);
  for (uint32_t u = 0x3f800000u - 300000; u < 0x3f800000u + 300000; ++u) check_log(u);
  check_log(Bits(1.1920928955078125e-07f));
  check_log(Bits(1.0e-20f));
  check_log(0x7f800000u);   // +inf
  for (uint64_t u = 0; u < 0x100000000ull; u += 127) check_exp((uint32_t)u);
  for (uint32_t u = 0x42b00000u; u < 0x42b40000u; ++u) { check_exp(u); check_exp(u | 0x80000000u); }   // |x| in [88, 90): over/underflow edges
  for (uint32_t u = 0x42cf0000u; u < 0x42d10000u; ++u) check_exp(u | 0x80000000u);                        // x around -103.9
  check_exp(0x7f800000u); check_exp(0xff800000u); check_exp(0); check_exp(0x80000000u);
  printf("%ld values, %ld mismatches\n", n, bad);
  return bad ? 1 : 0;
}
