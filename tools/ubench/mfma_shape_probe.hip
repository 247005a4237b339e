// Developer microbenchmark (round 3): which fp16 MFMA shape should carry the f16x3 GEMM?
// Bare loops on RANDOM data, 8 waves per CU (2 per SIMD), 128 accumulator registers per wave (a 128 x 64
// tile), operands in registers.  One "unit" = the MACs of a 128 x 64 x 16 block (what one of the three
// f16x3 terms costs per k16 step).
//   0: v_mfma_f32_32x32x16_f16   8 MFMAs per unit            (the round-2 kernel's instruction)
//   1: v_mfma_f32_16x16x32_f16  16 MFMAs per unit
//   2: v_mfma_f32_16x16x16_f16  32 MFMAs per unit            (CDNA3-era k16 form: full rate or half rate on gfx950?)
//   3: per 16 x 16 tile one 16x16x32 + one 16x16x16 = 3 units of f16x3 work per k16 step:
//      [Xhi|Xlo].[Whi|Whi] (hi hi + lo hi, k32 fully used) + Xhi.Wlo (k16)
//   4: v_mfma_f32_32x32x8_f16   16 MFMAs per unit
// Prints TFLOP/s (HIP events), in-kernel clock (s_memtime / s_memrealtime) and cycles per unit and wave.
// build: hipcc --offload-arch=gfx950 -O3 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void K(const f16x8 *__restrict__ rnd, float *out, long long *stamps, int iters) {
  const int tid = threadIdx.x;
  f16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = rnd[(blockIdx.x * 4096 + tid * 8 + i) & 0xFFFFF];
  for (int i = 0; i < 4; ++i) b[i] = rnd[(blockIdx.x * 4096 + tid * 4 + i + 2048) & 0xFFFFF];
  f16x4 a4[8], b4[4];
  for (int i = 0; i < 8; ++i) a4[i] = f16x4{a[i][0], a[i][1], a[i][2], a[i][3]};
  for (int i = 0; i < 4; ++i) b4[i] = f16x4{b[i][4], b[i][5], b[i][6], b[i][7]};
  f32x16 acc32[4][2];
  f32x4 acc16[8][4];
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) acc32[x][y] = f32x16{0};
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 4; ++y) acc16[x][y] = f32x4{0};
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {            // 2 units
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc32[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[x + 4 * ks], b[y + 2 * ks], acc32[x][y], 0, 0, 0);
    } else if (SHAPE == 1) {     // 2 units
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x], b[y], acc16[x][y], 0, 0, 0);
    } else if (SHAPE == 2) {     // 2 units
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y)
            acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[(x + ks) & 7], b4[y], acc16[x][y], 0, 0, 0);
    } else if (SHAPE == 3) {     // 3 units
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x], b[y], acc16[x][y], 0, 0, 0);
          acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[x], b4[y], acc16[x][y], 0, 0, 0);
        }
    } else {                     // 2 units
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc32[x][y] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4[(x + ks) & 7], b4[(y + ks) & 3], acc32[x][y], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) s += acc32[x][y][r];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 4; ++y) for (int r = 0; r < 4; ++r) s += acc16[x][y][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const f16x8 *rnd, float *out, long long *stamps, const char *name, double units) {
  const int blocks = 256, iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 30; ++w) K<SHAPE><<<blocks, 512>>>(rnd, out, stamps, iters);
  hipDeviceSynchronize();
  const int reps = 40;
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) K<SHAPE><<<blocks, 512>>>(rnd, out, stamps, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)reps * blocks * 8 * iters * units * 2.0 * 128 * 64 * 16;
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  printf("%-58s %7.1f TFLOP/s  %.3f of 2500  clock %.0f MHz  %.1f cycles per unit and wave (ideal 128 with the SIMD alone... 2 waves share: 256)\n",
         name, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, clk[blocks / 2], (double)h[0] / iters / units);
}

int main() {
  const size_t n = 1 << 20;
  std::vector<_Float16> h(n * 8);
  srand(7);
  for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.0f - 1.0f);
  f16x8 *rnd; float *out; long long *stamps;
  hipMalloc(&rnd, n * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 16);
  hipMemcpy(rnd, h.data(), n * 16, hipMemcpyHostToDevice);
  run<0>(rnd, out, stamps, "32x32x16 f16", 2);
  run<1>(rnd, out, stamps, "16x16x32 f16", 2);
  run<2>(rnd, out, stamps, "16x16x16 f16 (CDNA3-era k16 form)", 2);
  run<3>(rnd, out, stamps, "16x16x32 + 16x16x16 per tile (f16x3 on one k16 slab)", 3);
  run<4>(rnd, out, stamps, "32x32x8 f16 (CDNA3-era)", 2);
  run<0>(rnd, out, stamps, "32x32x16 f16 (again, warm)", 2);
  return 0;
}
