// CPU check of csrc/pk_logf.h (the logf the front-end kernel uses) against the C library the
// reference links: every 61st positive normal float, a dense run around 1.0, and the floor.
// The exhaustive sweep is tools/logf_check.c.  Built and run by tests/test_cpp_host.py.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../pocketkaldi_amd/csrc/pk_logf.h"

static uint32_t Bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float FromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
  static const double tab[pkmi::kLogfTableDoubles] = PK_LOGF_TABLE_INIT;
  long bad = 0, n = 0;
  auto check = [&](uint32_t u) {
    const float x = FromBits(u);
    if (Bits(logf(x)) != Bits(pkmi::LogfRestated(x, tab))) {
      if (bad < 5) fprintf(stderr, "x = %a: libm %a, restated %a\n", x, logf(x), pkmi::LogfRestated(x, tab));
      ++bad;
    }
    ++n;
  };
  for (uint64_t u = 0x00800000u; u < 0x7f800000u; u += 61) check((uint32_t)u);
  for (uint32_t u = 0x3f800000u - 300000; u < 0x3f800000u + 300000; ++u) check(u);
  check(Bits(1.1920928955078125e-07f));
  check(0x7f800000u);   // +inf
  printf("%ld values, %ld mismatches\n", n, bad);
  return bad ? 1 : 0;
}
