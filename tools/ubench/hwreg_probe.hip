// Developer probe: what do HW_REG_HW_ID / HW_REG_LDS_ALLOC read for co-resident workgroups?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 3) void Probe(unsigned *out, int spin) {
  __shared__ float smem[12288];   // 48 KiB, like the GEMM
  smem[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);
  float x = smem[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if ((threadIdx.x & 63) == 0) {
    int w = threadIdx.x >> 6;
    out[(blockIdx.x * 4 + w) * 3 + 0] = hwid;
    out[(blockIdx.x * 4 + w) * 3 + 1] = lds;
    out[(blockIdx.x * 4 + w) * 3 + 2] = (unsigned)x;
  }
}
int main() {
  const int nb = 768 * 2;
  unsigned *d; hipMalloc(&d, nb * 12 * 4);
  hipLaunchKernelGGL(Probe, dim3(nb), dim3(256), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 12);
  hipMemcpy(h.data(), d, nb * 12 * 4, hipMemcpyDeviceToHost);
  for (int b : {0, 8, 16, 256, 264, 512, 520, 768, 776, 1024, 1280, 1535}) {
    printf("wg %4d:", b);
    for (int w = 0; w < 4; ++w) {
      unsigned id = h[(b * 4 + w) * 3], l = h[(b * 4 + w) * 3 + 1];
      printf("  [wave_id %u simd %u cu %u se %u | lds %08x]", id & 15, (id >> 4) & 3, (id >> 8) & 15, (id >> 13) & 7, l);
    }
    printf("\n");
  }
  // how many distinct lds words, and per (cu-ish) grouping
  return 0;
}
