// Developer probe: what a layer boundary costs INSIDE a persistent kernel on gfx950 -- 256 workgroups
// (one per CU, 512 threads), per round every workgroup stores a 16 KiB tile, signals its row block's
// counter, waits until the 16 producers of its row block have signalled, and reads one 16-byte piece per
// thread from those producers' tiles (checked: a stale read counts as an error).
//   mode 0  plain stores, release fence (buffer_wbl2 sc1) + atomic add; acquire fence (buffer_inv sc1); plain loads
//   mode 1  sc1 (agent-coherent, write-through) stores + s_waitcnt + atomic add; sc1 loads; no cache maintenance
//   mode 2  no synchronisation at all (stores + loads only; errors expected) -- the floor
//   mode 3  XCD-local hand-off: a workgroup reads HW_REG_XCC_ID, takes a ticket on that XCD and joins one of
//           the XCD's two row blocks, so producers and consumers of a row block share ONE L2: plain stores,
//           drained vmcnt(0), a counter, sc1 (L1-bypassing, L2-served) loads
// build: hipcc --offload-arch=gfx950 -O3 sync_probe.hip -o sync_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256, kThreads = 512, kTile = 4096;       // floats per tile (16 KiB)
__global__ __launch_bounds__(kThreads) void Probe(float *buf0, float *buf1, unsigned *cnt, int rounds, int mode,
                                                  long long *ticks, unsigned *errors) {
  int wg = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ int s_slot;
  if (mode == 3) {              // logical id from where the workgroup really runs
    if (tid == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
      const unsigned ticket = __hip_atomic_fetch_add(cnt + rounds * 16 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_slot = (xcc < 8 && ticket < 32) ? (int)(xcc * 32 + ticket) : -1;
    }
    __syncthreads();
    wg = s_slot;
    if (wg < 0) { if (tid == 0) atomicAdd(errors, 1u << 28); return; }
  }
  const int j = wg / 16;
  long long t0 = 0;
  if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float *out = (r & 1) ? buf1 : buf0;
    const float v = (float)(r * 1024 + wg);
    f32x4 val = {v, v, v, v};
    float *dst = out + (size_t)wg * kTile + tid * 8;
    if (mode == 1) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %1, off offset:16 sc1" ::"v"(dst), "v"(val) : "memory");
    } else {
      *reinterpret_cast<f32x4 *>(dst) = val;
      *reinterpret_cast<f32x4 *>(dst + 4) = val;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mode != 2) {
      if (tid == 0) {
        if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(cnt + r * 16 + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt + r * 16 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u && ++spins < (1 << 16))
          __builtin_amdgcn_s_sleep(1);
        if (spins >= (1 << 16)) bad += 1u << 20;
      }
      __syncthreads();
      if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // one 16-byte piece per thread from producer (tid % 16) of this row block
    const int prod = j * 16 + (tid & 15);
    const float *src = out + (size_t)prod * kTile + (tid >> 4) * 8;
    f32x4 got;
    if (mode == 1 || mode == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
    else got = *reinterpret_cast<const f32x4 *>(src);
    if (got[0] != (float)(r * 1024 + prod)) ++bad;
  }
  if (bad) atomicAdd(errors, bad);
  if (tid == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}
int main() {
  const int rounds = 200;
  float *b0, *b1; unsigned *cnt, *err; long long *ticks;
  hipMalloc(&b0, sizeof(float) * kWG * kTile); hipMalloc(&b1, sizeof(float) * kWG * kTile);
  hipMalloc(&cnt, 4 * (rounds * 16 + 16)); hipMalloc(&err, 4); hipMalloc(&ticks, 8 * kWG);
  const char *names[] = {"fences (wbl2 / inv)", "sc1 stores + sc1 loads", "no synchronisation (floor)", "XCD-local: plain stores, sc1 loads"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cnt, 0, 4 * (rounds * 16 + 16)); hipMemset(err, 0, 4);
      hipMemset(b0, 0, sizeof(float) * kWG * kTile); hipMemset(b1, 0, sizeof(float) * kWG * kTile);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(Probe, dim3(kWG), dim3(kThreads), 0, 0, b0, b1, cnt, rounds, mode, ticks, err);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      std::vector<long long> ht(kWG); hipMemcpy(ht.data(), ticks, 8 * kWG, hipMemcpyDeviceToHost);
      long long mx = 0; for (auto t : ht) mx = t > mx ? t : mx;
      if (rep) printf("mode %d %-32s %7.3f us/round (event), %7.3f us/round (100 MHz ticks, slowest workgroup), stale/timeout count %u\n",
                      mode, names[mode], ms * 1e3 / rounds, mx * 0.01 / rounds, herr);
    }
  }
  return 0;
}
