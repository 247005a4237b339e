// Developer microbenchmark: what the chip sustains on fp16 MFMA under load ("the envelope" the
// f16x3 GEMM is priced against).  Bare loops on RANDOM data, 8 waves per CU (2 per SIMD) like
// GemmF16Kernel, the same 128 accumulator registers per wave (a 128 x 64 output tile):
//   shape 0: v_mfma_f32_32x32x16_f16 (4 x 2 tiles)     shape 1: v_mfma_f32_16x16x32_f16 (8 x 4 tiles)
//   src   0: operands stay in registers                src   1: every operand re-read from LDS by ds_read_b128
// Reports TFLOP/s (HIP events over ~1 s of back-to-back launches) and the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6).
// build: hipcc --offload-arch=gfx950 -O3 mfma_f16_envelope.hip -o mfma_f16_envelope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int SRC>
__global__ __launch_bounds__(512, 2) void K(const f16x8 *__restrict__ rnd, float *out, long long *stamps, int iters) {
  __shared__ f16x8 lds[4096];            // 64 KiB of random halves
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) lds[i] = rnd[(blockIdx.x * 4096 + i) & 0xFFFFF];
  __syncthreads();
  f16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = lds[(tid * 8 + i) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = lds[(tid * 4 + i + 2048) & 4095];
  f32x16 acc32[4][2];
  f32x4 acc16[8][4];
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) acc32[x][y] = f32x16{0};
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 4; ++y) acc16[x][y] = f32x4{0};
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f16x8 na[8], nb[4];
  for (int it = 0; it < iters; ++it) {
    if (SRC == 1) {                      // 12 ds_read_b128 per k32 of the wave tile, like the GEMM's fragment
      // traffic; software-pipelined: the next iteration's operands are requested before this one's MFMAs
      const int base = ((it + 1) * 64 + lane) & 2047;
#pragma unroll
      for (int i = 0; i < 8; ++i) na[i] = lds[(base + i * 64) & 4095];
#pragma unroll
      for (int i = 0; i < 4; ++i) nb[i] = lds[(base + 2048 + i * 64) & 4095];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (SHAPE == 0) {
      // one k32 of a 128 x 64 tile: 4 x 2 tiles x 2 k16 steps = 16 MFMAs of 32 cycles
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc32[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[x + 4 * ks], b[y + 2 * ks], acc32[x][y], 0, 0, 0);
    } else {
      // the same k32 of the same tile: 8 x 4 tiles = 32 MFMAs of 16 cycles
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc16[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x], b[y], acc16[x][y], 0, 0, 0);
    }
    if (SRC == 1) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = na[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = nb[i];
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) s += acc32[x][y][r];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 4; ++y) for (int r = 0; r < 4; ++r) s += acc16[x][y][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int SRC>
void run(const f16x8 *rnd, float *out, long long *stamps, const char *name) {
  const int blocks = 256, iters = 20000;           // one 8-wave workgroup per CU
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 40; ++w) K<SHAPE, SRC><<<blocks, 512>>>(rnd, out, stamps, iters);   // warm: reach the steady clock
  hipDeviceSynchronize();
  const int reps = 60;
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) K<SHAPE, SRC><<<blocks, 512>>>(rnd, out, stamps, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per iteration and wave: 128 x 64 x 32 MACs
  const double flops = (double)reps * blocks * 8 * iters * 2.0 * 128 * 64 * 32;
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double cyc_per_it = (double)h[0] / iters;
  printf("%-44s %7.1f TFLOP/s  %.2f of 2500  in-kernel clock %.0f MHz (median)  %.1f cycles per k32 (ideal 512)\n",
         name, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, clk[blocks / 2], cyc_per_it);
}

int main() {
  const size_t n = 1 << 20;
  std::vector<_Float16> h(n * 8);
  srand(7);
  for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.0f - 1.0f);
  f16x8 *rnd; float *out; long long *stamps;
  hipMalloc(&rnd, n * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 16);
  hipMemcpy(rnd, h.data(), n * 16, hipMemcpyHostToDevice);
  run<0, 0>(rnd, out, stamps, "32x32x16 f16, operands in registers");
  run<1, 0>(rnd, out, stamps, "16x16x32 f16, operands in registers");
  run<0, 1>(rnd, out, stamps, "32x32x16 f16, operands re-read from LDS");
  run<1, 1>(rnd, out, stamps, "16x16x32 f16, operands re-read from LDS");
  run<0, 0>(rnd, out, stamps, "32x32x16 f16, registers (again, warm)");
  return 0;
}
