// Developer microbenchmark (round 3): does the fp32 MFMA shape change the clock the chip holds, as the fp16 shape does?
// Bare loops on RANDOM data, operands in registers, 64 accumulator registers per wave (a 64 x 64 tile as in GemmKernel),
// WAVES waves per SIMD.  One unit = the MACs of a 64 x 64 x 4 block.
//   0: v_mfma_f32_32x32x2_f32  (GemmKernel's)  4 tiles x 2 k-steps = 8 MFMAs per unit
//   1: v_mfma_f32_16x16x4_f32                 16 tiles x 1 k-step  = 16 MFMAs per unit
// build: hipcc --offload-arch=gfx950 -O3 mfma_f32_shape_probe.hip -o mfma_f32_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void K(const float *__restrict__ rnd, float *out, long long *stamps, int iters) {
  const int tid = threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd[(blockIdx.x * 4096 + tid * 8 + i) & 0xFFFFF]; b[i] = rnd[(blockIdx.x * 4096 + tid * 8 + i + 2048) & 0xFFFFF]; }
  f32x16 acc32[2][2];
  f32x4 acc16[4][4];
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) acc32[x][y] = f32x16{0};
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc16[x][y] = f32x4{0};
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc32[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x + 2 * ks], b[y + 2 * ks], acc32[x][y], 0, 0, 0);
    } else {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y + 4], acc16[x][y], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) s += acc32[x][y][r];
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) for (int r = 0; r < 4; ++r) s += acc16[x][y][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const float *rnd, float *out, long long *stamps, const char *name) {
  const int blocks = 256 * 3, iters = 40000;       // three 4-wave workgroups per CU, like GemmKernel
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 20; ++w) K<SHAPE><<<blocks, 256>>>(rnd, out, stamps, iters);
  (void)hipDeviceSynchronize();
  const int reps = 30;
  (void)hipEventRecord(a);
  for (int r = 0; r < reps; ++r) K<SHAPE><<<blocks, 256>>>(rnd, out, stamps, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double flops = (double)reps * blocks * 4 * iters * 2.0 * 64 * 64 * 4;
  std::vector<long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  printf("%-28s %7.1f TFLOP/s  %.3f of 157.3  clock %.0f MHz  %.1f cycles per unit and wave (3 waves per SIMD share: ideal 3 x 128 = 384... alone 128)\n",
         name, flops / ms / 1e9, flops / ms / 1e9 / 157.3, clk[blocks / 2], (double)h[0] / iters);
}

int main() {
  const size_t n = 1 << 20;
  std::vector<float> h(n);
  srand(7);
  for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.0f - 1.0f;
  float *rnd, *out; long long *stamps;
  (void)hipMalloc(&rnd, n * 4); (void)hipMalloc(&out, 768 * 256 * 4); (void)hipMalloc(&stamps, 768 * 16);
  (void)hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice);
  run<0>(rnd, out, stamps, "32x32x2 f32");
  run<1>(rnd, out, stamps, "16x16x4 f32");
  run<0>(rnd, out, stamps, "32x32x2 f32 (again)");
  run<1>(rnd, out, stamps, "16x16x4 f32 (again)");
  return 0;
}
