// d2h_probe.hip -- what the 12 MB device-to-host transfer of pk_decodable_init costs, by destination kind and split.
// build: hipcc --offload-arch=gfx950 -O2 d2h_probe.hip -o d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void Spin(float *p, int us) {             // a stand-in for "compute": `us` microseconds on one wave (100 MHz counter)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (p && us < 0) p[threadIdx.x] = 1.0f;
}

int main() {
  const size_t bytes = 998ull * 3000 * 4;
  float *d = nullptr;
  hipMalloc(&d, bytes);
  hipMemset(d, 1, bytes);
  hipStream_t s[4];
  for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  float *pinned = nullptr;
  hipHostMalloc(reinterpret_cast<void **>(&pinned), bytes, hipHostMallocDefault);
  float *reg = static_cast<float *>(malloc(bytes));
  memset(reg, 0, bytes);
  double t0 = now_us();
  hipError_t re = hipHostRegister(reg, bytes, hipHostRegisterDefault);
  printf("hipHostRegister(12 MB malloc): %.1f us (%s)\n", now_us() - t0, hipGetErrorString(re));
  float *pageable = static_cast<float *>(malloc(bytes));
  memset(pageable, 0, bytes);

  auto run = [&](const char *name, float *dst, int pieces, int nstreams) {
    double best = 1e30, sum = 0;
    const int reps = 30;
    for (int r = 0; r < reps + 3; ++r) {
      hipDeviceSynchronize();
      const double a = now_us();
      const size_t chunk = bytes / pieces;
      for (int p = 0; p < pieces; ++p)
        hipMemcpyAsync(reinterpret_cast<char *>(dst) + p * chunk, reinterpret_cast<char *>(d) + p * chunk, chunk,
                       hipMemcpyDeviceToHost, s[p % nstreams]);
      for (int i = 0; i < nstreams; ++i) hipStreamSynchronize(s[i]);
      const double t = now_us() - a;
      if (r >= 3) { sum += t; if (t < best) best = t; }
    }
    printf("%-46s pieces %d streams %d: avg %.1f us  best %.1f us  (%.1f GB/s)\n", name, pieces, nstreams, sum / reps, best,
           bytes / (sum / reps) * 1e-3);
  };
  run("hipHostMalloc", pinned, 1, 1);
  run("hipHostRegister'd malloc", reg, 1, 1);
  run("pageable malloc", pageable, 1, 1);
  run("hipHostMalloc", pinned, 2, 1);
  run("hipHostMalloc", pinned, 2, 2);
  run("hipHostMalloc", pinned, 4, 1);
  run("hipHostMalloc", pinned, 4, 4);
  run("hipHostRegister'd malloc", reg, 4, 4);
  run("hipHostRegister'd malloc", reg, 2, 2);
  run("pageable malloc", pageable, 4, 4);
  // a kernel, then the copy behind it on the same stream, vs the copy alone: the seam
  for (int iters : {0, 50, 150}) {
    if (iters > 0) break;
    double sum = 0;
    for (int r = 0; r < 23; ++r) {
      hipDeviceSynchronize();
      const double a = now_us();
      if (iters) hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s[0], d, iters);
      hipMemcpyAsync(reg, d, bytes, hipMemcpyDeviceToHost, s[0]);
      hipStreamSynchronize(s[0]);
      if (r >= 3) sum += now_us() - a;
    }
    printf("kernel of ~%d us + 12 MB copy behind it (registered): %.1f us\n", iters, sum / 20);
  }
  // THE question of pk_decodable_init's pipeline: does a copy on stream B, released by an event of stream A, run under
  // the kernels stream A goes on with?  A: K(100 us) ev K(120 us);  B: wait ev, copy 12 MB (220 us).
  // overlapped = 100 + 220 = 320 us; serialised = 440 us.
  {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    struct { const char *name; float *dst; } kinds[3] = {{"hipHostMalloc", pinned}, {"registered malloc", reg}, {"pageable malloc", pageable}};
    for (auto &k : kinds) {
      double sum = 0, sum_call = 0;
      for (int r = 0; r < 23; ++r) {
        hipDeviceSynchronize();
        const double a = now_us();
        hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s[0], d, 100);
        hipEventRecord(ev, s[0]);
        hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s[0], d, 120);
        hipStreamWaitEvent(s[1], ev, 0);
        const double c0 = now_us();
        hipMemcpyAsync(k.dst, d, bytes, hipMemcpyDeviceToHost, s[1]);
        const double c1 = now_us();
        hipStreamSynchronize(s[0]);
        hipStreamSynchronize(s[1]);
        if (r >= 3) { sum += now_us() - a; sum_call += c1 - c0; }
      }
      printf("A: K100 ev K120 | B: wait ev, copy 12 MB -> %-18s: total %.1f us (320 = overlapped, 440 = serialised); the hipMemcpyAsync call itself %.1f us\n",
             k.name, sum / 20, sum_call / 20);
    }
    // the same on ONE stream, for reference: K100, copy, (no second kernel)
    double sum = 0;
    for (int r = 0; r < 23; ++r) {
      hipDeviceSynchronize();
      const double a = now_us();
      hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s[0], d, 100);
      hipMemcpyAsync(pinned, d, bytes, hipMemcpyDeviceToHost, s[0]);
      hipStreamSynchronize(s[0]);
      if (r >= 3) sum += now_us() - a;
    }
    printf("one stream: K100 then copy 12 MB -> hipHostMalloc: total %.1f us (320 = no seam)\n", sum / 20);
  }
  // zero-copy: a kernel writes the host block directly
  float *dev_view = nullptr;
  if (hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_view), pinned, 0) == hipSuccess) {
    double sum = 0;
    for (int r = 0; r < 23; ++r) {
      hipDeviceSynchronize();
      const double a = now_us();
      hipMemcpyAsync(dev_view, d, bytes, hipMemcpyDeviceToDevice, s[0]);   // blit kernel writing over the link
      hipStreamSynchronize(s[0]);
      if (r >= 3) sum += now_us() - a;
    }
    printf("device-side copy into the mapped pinned block (kernel writes over the link): %.1f us\n", sum / 20);
  }
  return 0;
}
