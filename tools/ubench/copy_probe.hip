// Developer probe: streaming copy with 64-bit per-lane addresses vs scalar base + 32-bit lane offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows of 3000 floats (750 float4), one workgroup walks rows; 256 threads x 3 float4
__global__ __launch_bounds__(256) void CopyVaddr(const float *__restrict__ in, float *__restrict__ out, int rows, int64_t ld) {
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const f32x4 *x = reinterpret_cast<const f32x4 *>(in + row * ld);
    f32x4 *y = reinterpret_cast<f32x4 *>(out + row * ld);
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { int q = threadIdx.x + c * 256; if (q < 750) v[c] = __builtin_nontemporal_load(&x[q]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { int q = threadIdx.x + c * 256; if (q < 750) __builtin_nontemporal_store(v[c] * 1.0001f, &y[q]); }
  }
}

__global__ __launch_bounds__(256) void CopySaddr(const float *__restrict__ in, float *__restrict__ out, int rows, int64_t ld) {
  unsigned off[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { int q = threadIdx.x + c * 256; off[c] = (q < 750 ? q : 749) * 16; }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const char *xb = reinterpret_cast<const char *>(in + row * ld);
    const char *yb = reinterpret_cast<const char *>(out + row * ld);
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v[c]) : "v"(off[c]), "s"(xb) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 r = v[c] * 1.0001f;
      int q = threadIdx.x + c * 256;
      if (q < 750) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off[c]), "v"(r), "s"(yb) : "memory");
    }
  }
}

int main() {
  const int rows = 258048; const int64_t ld = 3072;
  float *in, *out;
  hipMalloc(&in, rows * ld * 4); hipMalloc(&out, rows * ld * 4);
  hipMemset(in, 0, rows * ld * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int which = 0; which < 2; ++which)
    for (int grid : {2048, 4096, 8192}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (which == 0) hipLaunchKernelGGL(CopyVaddr, dim3(grid), dim3(256), 0, 0, in, out, rows, ld);
        else hipLaunchKernelGGL(CopySaddr, dim3(grid), dim3(256), 0, 0, in, out, rows, ld);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("%s grid %d: %.3f ms, %.2f TB/s (read+write of 3000 floats per row)\n", which ? "saddr" : "vaddr", grid, ms, rows * 3000.0 * 8 / ms / 1e9);
      }
    }
  return 0;
}
