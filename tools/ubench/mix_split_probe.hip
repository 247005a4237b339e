// mix_split_probe.hip -- is lo = fp16(x - float(hi)) the same bits when formed by v_fma_mixlo/hi_f16 (one instruction
// per value) as by v_cvt_f32_f16 + v_sub_f32 + v_cvt_pk_f16_f32 (2.5)?  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x2 LoMix(f16x2 hi, float x0, float x1) {
  unsigned h = __builtin_bit_cast(unsigned, hi), l = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x0));
  asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
  return __builtin_bit_cast(f16x2, l);
}
__global__ void k(const float *x, int n, unsigned *ref, unsigned *mix, unsigned *his) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float a = fminf(fmaxf(x[2 * i], -65504.0f), 65504.0f), b = fminf(fmaxf(x[2 * i + 1], -65504.0f), 65504.0f);
  f16x2 hi = f16x2{(_Float16)a, (_Float16)b};
  f16x2 lo = f16x2{(_Float16)(a - (float)hi[0]), (_Float16)(b - (float)hi[1])};
  ref[i] = __builtin_bit_cast(unsigned, lo);
  mix[i] = __builtin_bit_cast(unsigned, LoMix(hi, a, b));
  his[i] = __builtin_bit_cast(unsigned, hi);
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    // magnitudes from 2^-30 to 2^17 (clamped above 65504), both signs, some exact zeros and halves
    const int e = rand() % 48 - 30;
    const float m = 1.0f + (rand() & 0xffffff) / 16777216.0f;
    h[i] = (rand() & 1 ? -1.0f : 1.0f) * ldexpf(m, e);
    if (i % 1000 == 0) h[i] = 0.0f;
    if (i % 1000 == 1) h[i] = ldexpf(1.0f, e);
  }
  float *dx; unsigned *dr, *dm, *dh;
  hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 2); hipMalloc(&dm, n * 2); hipMalloc(&dh, n * 2);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dr, dm, dh);
  std::vector<unsigned> r(n / 2), m(n / 2);
  hipMemcpy(r.data(), dr, n * 2, hipMemcpyDeviceToHost);
  hipMemcpy(m.data(), dm, n * 2, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n / 2; ++i) if (r[i] != m[i]) { if (bad < 5) printf("differs at %d: ref %08x mix %08x  x = %g %g\n", i, r[i], m[i], h[2 * i], h[2 * i + 1]); ++bad; }
  printf("%d pairs, %ld differ\n", n / 2, bad);
  return bad != 0;
}
