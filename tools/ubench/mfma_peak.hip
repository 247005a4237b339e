// Developer microbenchmark: fp32 MFMA issue rate vs waves/SIMD, with/without LDS operand reads
// and with/without a per-16-MFMA workgroup barrier.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef const __attribute__((address_space(1))) void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;
template <int MODE>   // 0: registers only; 1: LDS reads per k-step; 2: + barrier per 32 MFMA; 3: + 4 LDS-DMA pieces per 32 MFMA
__global__ __launch_bounds__(256) void K(float *out, int iters, const float *src) {
  __shared__ float lds[3 * 2 * 16 * 128];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 16 * 128; i += 256) lds[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float p0 = lane * 0.001f, p1 = 0.5f, q0 = 0.25f, q1 = 1.5f;
  const float *ps = lds + (lane & 31) + (lane >> 5) * 128;
  for (int it = 0; it < iters; ++it) {
    float pf[8][2], qf[8][2];
    if (MODE >= 1) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        pf[ks][0] = ps[2 * ks * 128]; pf[ks][1] = ps[2 * ks * 128 + 32];
        qf[ks][0] = ps[2048 + 2 * ks * 128]; qf[ks][1] = ps[2048 + 2 * ks * 128 + 32];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float x0 = MODE ? pf[ks][0] : p0, x1 = MODE ? pf[ks][1] : p1, y0 = MODE ? qf[ks][0] : q0, y1 = MODE ? qf[ks][1] : q1;
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a3, 0, 0, 0);
    }
    if (MODE == 3) {
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const float *g = src + ((size_t)blockIdx.x * 64 + (it & 1023)) * 4096 + wave * 1024 + lane * 4;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((GlobalPtr)(g + p * 256), (LdsPtr)(lds + 4096 + ((it % 2) * 4096) + wave * 1024 + p * 256), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 2) __syncthreads();
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *src;
template <int MODE>
void run(int blocks_per_cu, int iters, float *d) {
  int blocks = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  K<MODE><<<blocks, 256>>>(d, 10, src);
  hipDeviceSynchronize();
  hipEventRecord(a);
  K<MODE><<<blocks, 256>>>(d, iters, src);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flops = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("mode %d  blocks/CU %d (waves/SIMD %d): %.1f TFLOP/s  (%.3f ms)\n", MODE, blocks_per_cu, blocks_per_cu, flops / ms / 1e9, ms);
}

int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  hipMalloc(&src, (size_t)1024 * 64 * 4096 * 4 + (1 << 24)); hipMemset(src, 0, (size_t)1024 * 64 * 4096 * 4);
  for (int bpc = 1; bpc <= 3; ++bpc) { run<0>(bpc, 4000, d); run<1>(bpc, 4000, d); run<2>(bpc, 4000, d); run<3>(bpc, 4000, d); }
  return 0;
}
