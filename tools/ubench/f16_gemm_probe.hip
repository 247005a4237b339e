// Developer probe: where a wave of GemmF16K32Kernel (default) / GemmF16Kernel (PK_MI355_F16_SHAPE=32) spends its k loop.  Builds the product kernel
// with s_memtime stamps around the three waits of a step (PK_F16_STAMPS; the product build has
// none) and runs one hidden layer of the wide model: 65536 rows, K = N = 2048.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../pocketkaldi_amd/csrc f16_gemm_probe.hip -o f16_gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
__device__ long long pk_f16_stamps[512 * 4 * 8];
#define PK_F16_STAMPS 1
#include "gemm_f16.hip"

int main() {
  const int rows = 65536, K = 2048, N = 2048;
  std::vector<_Float16> hx((size_t)rows * 2 * K), hw((size_t)N * 2 * K);
  srand(3);
  for (auto &v : hx) v = (_Float16)((rand() / (float)RAND_MAX) * 2.0f - 1.0f);
  for (auto &v : hw) v = (_Float16)((rand() / (float)RAND_MAX) * 0.1f - 0.05f);
  _Float16 *x, *w, *out; float *bias;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&out, (size_t)rows * 2 * N * 2); hipMalloc(&bias, N * 4);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  pkmi::GemmF16Args a;
  a.X = x; a.ldx = 2 * K; a.W = w; a.ldw = 2 * K; a.K = K; a.bias = bias; a.relu = 1;
  a.out_f32 = nullptr; a.out = out; a.ldo = 2 * N; a.tiles_m = rows / 256; a.tiles_n = N / 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) pkmi::LaunchGemmF16(a, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) pkmi::LaunchGemmF16(a, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("with stamps: %.3f ms per launch = %.1f TFLOP/s algorithmic (x3 = %.1f issued)\n", ms / reps,
         2.0 * rows * K * N / (ms / reps) / 1e9, 6.0 * rows * K * N / (ms / reps) / 1e9);
  std::vector<long long> h(512 * 4 * 8);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(pk_f16_stamps), h.size() * 8);
  const char *forced = getenv("PK_MI355_F16_SHAPE");
  const bool k32 = !(forced && atoi(forced) == 32);
  const int nsteps = k32 ? K / 32 : K / 16;
  double tot = 0, seg = 0, lg = 0, vm = 0, bar = 0, pro = 0, epi = 0; int n = 0;
  for (int b = 0; b < 512; ++b)
    for (int wv = 0; wv < 4; ++wv) {
      const long long *o = &h[(b * 4 + wv) * 8];
      if (o[0] <= 0) continue;
      tot += o[0]; seg += o[1]; lg += o[2]; vm += o[3]; bar += o[4]; pro += o[5]; epi += o[6]; ++n;
    }
  printf("%s\n", k32 ? "GemmF16K32Kernel (16x16x32, k32 steps; ideal 2 x 1536 cycles per step)" : "GemmF16Kernel (32x32x16, k16 steps; ideal 2 x 768 cycles per step)");
  printf("per wave: k loop %.0f cycles = %.1f per step; of it: between waits %.1f %%, "
         "lgkm wait %.1f %%, vmcnt wait %.1f %%, barrier wait %.1f %%;  prologue %.0f cycles, epilogue %.0f cycles (%.1f %% of the tile)\n",
         tot / n, tot / n / nsteps, 100 * seg / tot, 100 * lg / tot, 100 * vm / tot, 100 * bar / tot, pro / n, epi / n,
         100 * (pro + epi) / (tot + pro + epi));
  return 0;
}
