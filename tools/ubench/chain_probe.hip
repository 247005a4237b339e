// Developer probe: dependent-chain latency (cycles per step, one wave alone on its SIMD) of the
// candidate forms of the CMVN window-sum recurrence (cmvn.cc:44-70):
//   A  s = float(double(s) + double(x))                       (reference form, window filling)
//   B  s = float(double(s) + double(x) + -double(xo))         (reference form, window sliding)
//   C  s = s + x                                              (f32 add; == A for all floats)
//   D  s = float(double(s) + xd), xd a ready f64              (A with x pre-widened off the chain)
//   E  s = float((double(s) + xd) + nxo), both ready f64      (B with operands pre-widened)
//   F  like E, plus one ds_write_b32 of s and two ds_read_b64 per step (the real loop's LDS traffic)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_probe.hip -o chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)
template <int MODE>
__global__ __launch_bounds__(64) void Chain(const float *x, const float *xo, float *out, long long *cyc, int n) {
  __shared__ float sx[4096];
  __shared__ double sd[2][2048];
  __shared__ float ss[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) sx[i] = x[i];
  for (int i = lane; i < 2048; i += 64) { sd[0][i] = (double)x[i]; sd[1][i] = -(double)xo[i]; }
  __syncthreads();
  float s = x[lane];
  float xv[16], ov[16];
  double xd[16], od[16];
  for (int j = 0; j < 16; ++j) { xv[j] = x[lane + 64 * j]; ov[j] = xo[lane + 64 * j]; xd[j] = xv[j]; od[j] = -(double)ov[j]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE <= 1) asm volatile("" : "+v"(xv[j]), "+v"(ov[j]));     // not loop-invariant: the widening stays in the loop
      if (MODE == 0) { double a = s; a += xv[j]; s = (float)a; }
      if (MODE == 1) { double a = s; a += xv[j]; a += -1.0 * (double)ov[j]; s = (float)a; }
      if (MODE == 2) { s = s + xv[j]; }
      if (MODE == 3) { double a = s; a += xd[j]; s = (float)a; }
      if (MODE == 4) { double a = s; a += xd[j]; a += od[j]; s = (float)a; }
      if (MODE == 5) {
        const int idx = ((i + j) & 31) * 64 + lane;
        double a = s; a += sd[0][idx & 2047]; a += sd[1][idx & 2047]; s = (float)a; ss[idx] = s;
      }
      if (MODE == 6) {   // f32 add + the LDS traffic of the real loop
        const int idx = ((i + j) & 31) * 64 + lane;
        s = s + sx[idx]; ss[idx] = s;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = s + ss[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int n = 16000;
  std::vector<float> hx(4096), ho(4096);
  for (int i = 0; i < 4096; ++i) { hx[i] = 12.0f + 0.001f * (i % 977); ho[i] = 11.5f + 0.002f * (i % 811); }
  float *x, *xo, *out; long long *cyc;
  hipMalloc(&x, 4096 * 4); hipMalloc(&xo, 4096 * 4); hipMalloc(&out, 64 * 4 * 8); hipMalloc(&cyc, 8 * 8);
  hipMemcpy(x, hx.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(xo, ho.data(), 4096 * 4, hipMemcpyHostToDevice);
  const char *names[] = {"A cvt,add,cvt (x widened on chain wave)", "B cvt,add,add,cvt (x, xo widened on chain wave)", "C f32 add",
                         "D cvt,add,cvt (x ready f64)", "E cvt,add,add,cvt (ready f64)", "F = E + 2 ds_read_b64 + ds_write_b32",
                         "G f32 add + ds_read_b32 + ds_write_b32"};
  for (int m = 0; m < 7; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (m) {
        case 0: hipLaunchKernelGGL(Chain<0>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 1: hipLaunchKernelGGL(Chain<1>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 2: hipLaunchKernelGGL(Chain<2>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 3: hipLaunchKernelGGL(Chain<3>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 4: hipLaunchKernelGGL(Chain<4>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 5: hipLaunchKernelGGL(Chain<5>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
        case 6: hipLaunchKernelGGL(Chain<6>, dim3(1), dim3(64), 0, 0, x, xo, out, cyc, n); break;
      }
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %.1f cycles/step\n", names[m], (double)c / n);
  }
  return 0;
}
