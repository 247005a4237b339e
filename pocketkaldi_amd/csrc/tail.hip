// tail.hip -- the non-affine layers and the log-likelihood tail for gfx950:
// ReLU (nnet.cc:49-60), Normalize (nnet.cc:62-75), Softmax (nnet.cc:38-47 ->
// vector.cc:265-277), AcousticModel::Compute's floor/log/prior step
// (am.cc:106-112) and the acoustic scale (decodable.cc:15), plus the two layout
// changes between feature-major panels and frame-major rows.  All HBM-bound.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "pk_expf.h"
#include "pk_kernels.h"
#include "pk_logf.h"
#include "pk_tail_wave.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// the C library's logf / expf tables (pk_logf.h, pk_expf.h)
__device__ const double kLogfTab[kLogfTableDoubles] = PK_LOGF_TABLE_INIT;
__device__ const uint64_t kExpfTab[kExpfTableWords] = PK_EXPF_TABLE_INIT;

__global__ void ReluKernel(float *__restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t n4 = n / 4;
  f32x4 *x4 = reinterpret_cast<f32x4 *>(x);
  for (int64_t v = i; v < n4; v += stride) {
    f32x4 t = x4[v];
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = t[c] < 0.0f ? 0.0f : t[c];
    x4[v] = t;
  }
  for (int64_t v = n4 * 4 + i; v < n; v += stride) x[v] = x[v] < 0.0f ? 0.0f : x[v];
}

// One lane per frame, features walked in ascending order: the float sum of
// squares is the reference's sequential VecVec (vector.cc:252-262).  Feature-major panels
// (stride_f == 1): the 64 lanes of a wave read 64 consecutive frames of one feature -- coalesced.
__global__ void NormalizeKernel(float *__restrict__ x, int frames, int dim, int64_t stride_f,
                                int64_t stride_d) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  float *row = x + (int64_t)f * stride_f;
  float ss = 0.0f;
  for (int d = 0; d < dim; ++d) {
    const float v = row[(int64_t)d * stride_d];
    ss += v * v;
  }
  const float D = static_cast<float>(dim);
  const double squared_sum = ss;
  const float scale = static_cast<float>(sqrt(D / squared_sum));     // nnet.cc:71-72
  for (int d = 0; d < dim; ++d) row[(int64_t)d * stride_d] *= scale;
}

// The same for frame-major rows (stride_d == 1), where a lane per frame would walk its own row:
// a workgroup owns 64 frames, 64 x 64 tiles of them are loaded with coalesced 256-byte row pieces
// and transposed through LDS, and one wave -- lane = frame -- adds the squares tile after tile in
// feature order (the order is the reference's; only the loads are reshaped).  The scaling pass
// is elementwise and runs coalesced on all four waves.
__global__ __launch_bounds__(256) void NormalizeRowsKernel(float *__restrict__ x, int frames, int dim, int64_t ld) {
  __shared__ float tile[64][65];
  __shared__ float s_scale[64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f0 = blockIdx.x * 64;
  float ss = 0.0f;
  for (int d0 = 0; d0 < dim; d0 += 64) {
    for (int r = wv; r < 64; r += 4) {
      const int f = f0 + r, d = d0 + lane;
      tile[r][lane] = (f < frames && d < dim) ? x[(int64_t)f * ld + d] : 0.0f;   // + 0 * 0 leaves the sum as it is
    }
    __syncthreads();
    if (wv == 0) {
      const int nd = dim - d0 < 64 ? dim - d0 : 64;
      for (int c = 0; c < nd; ++c) {
        const float v = tile[lane][c];
        ss += v * v;
      }
    }
    __syncthreads();
  }
  if (wv == 0) {
    const float D = static_cast<float>(dim);
    const double squared_sum = ss;
    s_scale[lane] = static_cast<float>(sqrt(D / squared_sum));        // nnet.cc:71-72
  }
  __syncthreads();
  for (int r = wv; r < 64; r += 4) {
    const int f = f0 + r;
    if (f >= frames) break;
    const float sc = s_scale[r];
    float *row = x + (int64_t)f * ld;
    for (int d = lane; d < dim; d += 64) row[d] *= sc;
  }
}

// 32 x 32 tile transpose through LDS (+1 padding), 256 threads.
__global__ void TransposeKernel(const float *__restrict__ in, int64_t ld_in, int rows_in,
                                int cols_in, float *__restrict__ out, int64_t ld_out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < rows_in && cc < cols_in) ? in[(int64_t)rr * ld_in + cc] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int orow = c0 + r, ocol = r0 + tx;                // out[c][r] = in[r][c]
    if (orow < cols_in && ocol < rows_in) out[(int64_t)orow * ld_out + ocol] = tile[tx][r];
  }
}

__device__ __forceinline__ float WaveMax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float WaveSum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int kTailThreads = 256;
constexpr int kTailCacheMax = 8;   // float4s per thread kept in registers (n <= 8192)

// A workgroup walks rows blockIdx.x, + gridDim.x, ...  A row is read once (16-byte loads),
// kept in registers, reduced with wavefront shuffles + a 4-entry LDS exchange, written
// once; the loads of the next row are issued before the current row is reduced, so every
// CU keeps ~2 rows per resident workgroup in flight towards HBM.
template <int MODE, int kTailCache>   // kTailCache float4s per thread hold one row (n <= 1024 kTailCache)
__global__ __launch_bounds__(kTailThreads) void TailKernel(const float *__restrict__ in,
                                                           int64_t ld_in, int rows, int n,
                                                           const float *__restrict__ log_prior,
                                                           float scale, float *__restrict__ out,
                                                           int64_t ld_out) {
  __shared__ float red[2][kTailThreads / 64];
  const int tid = threadIdx.x;
  const int n4 = (n + 3) / 4;                    // ld_in is padded, tail lanes masked below
  const float kLogFloor = -46.051701859880914f;  // logf(1e-20f), am.cc:109-110
  const bool vec_out = ((ld_out & 3) == 0) && ((n & 3) == 0);

  // log priors of this thread's columns do not change from row to row
  f32x4 lp[kTailCache];
#pragma unroll
  for (int c = 0; c < kTailCache; ++c) {
    const int q = tid + c * kTailThreads;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      lp[c][e] = (MODE != kTailSoftmaxProb && q < n4 && q * 4 + e < n) ? log_prior[q * 4 + e] : 0.0f;
  }

  f32x4 v[kTailCache], nv[kTailCache];
  auto load_row = [&](int row, f32x4 (&dst)[kTailCache]) {
    const f32x4 *x = reinterpret_cast<const f32x4 *>(in + (int64_t)row * ld_in);
#pragma unroll
    for (int c = 0; c < kTailCache; ++c) {
      const int q = tid + c * kTailThreads;
      if (q < n4) dst[c] = __builtin_nontemporal_load(&x[q]);
    }
  };
  int row = blockIdx.x;
  if (row < rows) load_row(row, v);
  for (; row < rows; row += gridDim.x) {
    const int next = row + gridDim.x;
    if (next < rows) load_row(next, nv);
    float *y = out + (int64_t)row * ld_out;

    float m = -INFINITY, lse = 0.0f, inv = 0.0f;
    if (MODE != kTailLoglik) {
#pragma unroll
      for (int c = 0; c < kTailCache; ++c) {
        const int q = tid + c * kTailThreads;
        if (q < n4) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < n) m = fmaxf(m, v[c][e]);
        }
      }
      m = WaveMax(m);
      if ((tid & 63) == 0) red[0][tid >> 6] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < kTailCache; ++c) {
        const int q = tid + c * kTailThreads;
        if (q < n4) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < n) s += expf(v[c][e] - m);
        }
      }
      s = WaveSum(s);
      if ((tid & 63) == 0) red[1][tid >> 6] = s;
      __syncthreads();
      s = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      lse = m + logf(s);
      inv = 1.0f / s;
    }

#pragma unroll
    for (int c = 0; c < kTailCache; ++c) {
      const int q = tid + c * kTailThreads;
      if (q < n4) {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[c][e];
          if (MODE == kTailSoftmaxProb) {
            t = expf(t - m) * inv;
          } else {
            if (MODE == kTailSoftmaxLoglik) t = t - lse;          // log softmax, stable form
            else t = LogfRestated(t < 1.0e-20f ? 1.0e-20f : t, kLogfTab);   // am.cc:109-110, libm's logf
            if (t < kLogFloor) t = kLogFloor;                      // floor of am.cc:109 in the log domain
            t = (t + -1.0f * lp[c][e]) * scale;                    // am.cc:111, decodable.cc:15
          }
          r[e] = t;
        }
        if (vec_out) {
          __builtin_nontemporal_store(r, &reinterpret_cast<f32x4 *>(y)[q]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < n) y[q * 4 + e] = r[e];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kTailCache; ++c) v[c] = nv[c];
  }
}

// Rows wider than the register cache of TailKernel (n > 8192): the same arithmetic in three
// passes over the row (max, sum of exponentials, output); the second and third pass re-read the
// row from L2.  Outside the BASELINE models (3 000 / 8 000 pdfs); kept simple.
template <int MODE>
__global__ __launch_bounds__(kTailThreads) void TailWideKernel(const float *__restrict__ in, int64_t ld_in,
                                                               int rows, int n,
                                                               const float *__restrict__ log_prior, float scale,
                                                               float *__restrict__ out, int64_t ld_out) {
  __shared__ float red[2][kTailThreads / 64];
  const int tid = threadIdx.x;
  const float kLogFloor = -46.051701859880914f;  // logf(1e-20f), am.cc:109-110
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const float *x = in + (int64_t)row * ld_in;
    float *y = out + (int64_t)row * ld_out;
    float m = -INFINITY, lse = 0.0f, inv = 0.0f;
    if (MODE != kTailLoglik) {
      for (int c = tid; c < n; c += kTailThreads) m = fmaxf(m, x[c]);
      m = WaveMax(m);
      if ((tid & 63) == 0) red[0][tid >> 6] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
      float s = 0.0f;
      for (int c = tid; c < n; c += kTailThreads) s += expf(x[c] - m);
      s = WaveSum(s);
      if ((tid & 63) == 0) red[1][tid >> 6] = s;
      __syncthreads();
      s = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      lse = m + logf(s);
      inv = 1.0f / s;
    }
    for (int c = tid; c < n; c += kTailThreads) {
      float t = x[c];
      if (MODE == kTailSoftmaxProb) {
        t = expf(t - m) * inv;
      } else {
        if (MODE == kTailSoftmaxLoglik) t = t - lse;
        else t = LogfRestated(t < 1.0e-20f ? 1.0e-20f : t, kLogfTab);
        if (t < kLogFloor) t = kLogFloor;
        t = (t + -1.0f * log_prior[c]) * scale;
      }
      y[c] = t;
    }
    __syncthreads();          // red[] is reused by the next row
  }
}

// decodable.cc:24-31 for many (frame, transition-id) pairs at once, device side:
// out[i] = ll[frame[i]][tid2pdf[tid[i]]]
__global__ void GatherKernel(const float *__restrict__ ll, int64_t ld, const int32_t *__restrict__ tid2pdf,
                             int num_tids, const int32_t *__restrict__ frames,
                             const int32_t *__restrict__ tids, int n, float *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int tid = tids[i];
    const int pdf = tid2pdf ? tid2pdf[tid < num_tids ? tid : 0] : tid;
    out[i] = ll[(int64_t)frames[i] * ld + pdf];
  }
}

}  // namespace

void LaunchGather(const float *ll, int64_t ld, const int32_t *tid2pdf, int num_tids,
                  const int32_t *frames, const int32_t *tids, int n, float *out, hipStream_t stream) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(GatherKernel, dim3(blocks), dim3(256), 0, stream, ll, ld, tid2pdf, num_tids, frames,
                     tids, n, out);
}

void LaunchRelu(float *x, int64_t n, hipStream_t stream) {
  if (n <= 0) return;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ReluKernel, dim3((int)blocks), dim3(256), 0, stream, x, n);
}

void LaunchNormalize(float *x, int frames, int dim, int64_t stride_f, int64_t stride_d,
                     hipStream_t stream) {
  if (frames <= 0) return;
  if (stride_d == 1 && stride_f != 1)
    hipLaunchKernelGGL(NormalizeRowsKernel, dim3((frames + 63) / 64), dim3(256), 0, stream, x, frames, dim, stride_f);
  else
    hipLaunchKernelGGL(NormalizeKernel, dim3((frames + 63) / 64), dim3(64), 0, stream, x, frames,
                       dim, stride_f, stride_d);
}

void LaunchTransposeToRows(const float *in, int64_t ld_in, int dim, int frames, float *out,
                           int64_t ld_out, hipStream_t stream) {
  if (dim <= 0 || frames <= 0) return;
  dim3 grid((frames + 31) / 32, (dim + 31) / 32);
  hipLaunchKernelGGL(TransposeKernel, grid, dim3(256), 0, stream, in, ld_in, dim, frames, out,
                     ld_out);
}

void LaunchTransposeToCols(const float *in, int64_t ld_in, int frames, int dim, float *out,
                           int64_t ld_out, hipStream_t stream) {
  if (dim <= 0 || frames <= 0) return;
  dim3 grid((dim + 31) / 32, (frames + 31) / 32);
  hipLaunchKernelGGL(TransposeKernel, grid, dim3(256), 0, stream, in, ld_in, frames, dim, out,
                     ld_out);
}

// ---- the softmax tail exactly as the reference computes it (opt-in, PK_MI355_SOFTMAX_REFERENCE).
// vector.cc:265-277: e_j = expf(x_j); sum = e_0 + e_1 + ... accumulated in float, in order;
// p_j = e_j / sum.  am.cc:106-112: logf(max(p, 1e-20)) - log_prior; decodable.cc:15: x scale.
// With the C library's expf / logf restated (pk_expf.h, pk_logf.h) every operation is the
// reference's, so the log-likelihoods are its bit patterns (overflow for logits > 88.7 and all).
//
// The sum is a 3 000-term serial chain per frame, so frames go on lanes: a workgroup owns 64
// frames; four producer waves stream 64 x 64 tiles of logits (coalesced 256-byte row pieces),
// exponentiate them and leave e in LDS, and a fifth wave -- lane = frame -- adds tile after
// tile in column order, one tile behind the producers.  A second pass re-reads the logits
// (the 64 x n block cannot stay on chip) and finishes every element in parallel.
// (Measured alternative: two frames per workgroup with their exponentials kept in LDS -- one
// pass over HBM, no second expf -- is slower, 4.3 vs 3.4 ms: its serial sum is exposed.)

constexpr int kXRows = 64, kXCols = 64, kXLd = kXCols + 1;
constexpr int kXProducers = 4;                         // waves; each owns 16 rows of a tile

template <int MODE>
__global__ __launch_bounds__(64 * (kXProducers + 1)) void TailExactKernel(
    const float *__restrict__ in, int64_t ld_in, int rows, int n, const float *__restrict__ log_prior,
    float scale, float *__restrict__ out, int64_t ld_out) {
  __shared__ float tile[2][kXRows * kXLd];
  __shared__ float s_sum[kXRows];
  __shared__ double s_logf[kLogfTableDoubles];
  __shared__ uint64_t s_expf[kExpfTableWords];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // 0..3 producers, 4 the chain
  if (threadIdx.x < kLogfTableDoubles) s_logf[threadIdx.x] = kLogfTab[threadIdx.x];
  if (threadIdx.x < kExpfTableWords) s_expf[threadIdx.x] = kExpfTab[threadIdx.x];
  const int r0 = blockIdx.x * kXRows;
  const int ntiles = (n + kXCols - 1) / kXCols;
  constexpr int kPer = kXRows / kXProducers;            // rows of a tile per producer wave
  // this thread's rows (clamped: a partial block recomputes its last row, stores are masked)
  const float *xrow[kPer];
  if (wv < kXProducers) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      int r = r0 + wv * kPer + i;
      if (r > rows - 1) r = rows - 1;
      xrow[i] = in + (int64_t)r * ld_in;
    }
  }
  __syncthreads();

  // ---- pass 1: sum_j expf(x_j) in column order
  float s = 0.0f;
  float xv[kPer];
  auto load_tile = [&](int t) {
    const int c = t * kXCols + lane;
#pragma unroll
    for (int i = 0; i < kPer; ++i) xv[i] = c < n ? __builtin_nontemporal_load(xrow[i] + c) : 0.0f;
  };
  if (wv < kXProducers) load_tile(0);
  for (int t = 0; t <= ntiles; ++t) {
    if (wv < kXProducers) {
      if (t < ntiles) {
        float ev[kPer];
        const bool live = t * kXCols + lane < n;
#pragma unroll
        for (int i = 0; i < kPer; ++i) ev[i] = live ? ExpfRestatedWave(xv[i], s_expf) : 0.0f;   // + 0 leaves a sum as it is
        if (t + 1 < ntiles) load_tile(t + 1);       // (requesting these before the arithmetic measured slower)
        float *dst = tile[t & 1] + (wv * kPer) * kXLd + lane;
#pragma unroll
        for (int i = 0; i < kPer; ++i) dst[i * kXLd] = ev[i];
      }
    } else if (t >= 1) {
      const float *src = tile[(t - 1) & 1] + lane * kXLd;
#pragma unroll 16
      for (int c = 0; c < kXCols; ++c) s += src[c];                    // vector.cc:271
    }
    __syncthreads();
  }
  if (wv == kXProducers) s_sum[lane] = s;
  __syncthreads();
  if (wv == kXProducers) return;

  // ---- pass 2: p = e / sum (vector.cc:274-276) and the log-likelihood arithmetic
  float sum_r[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) sum_r[i] = s_sum[wv * kPer + i];
  for (int t = 0; t < ntiles; ++t) {
    const int c = t * kXCols + lane;
    if (c >= n) break;
    const float lp = MODE == kTailSoftmaxLoglik ? log_prior[c] : 0.0f;
#pragma unroll
    for (int i = 0; i < kPer; ++i) xv[i] = xrow[i][c];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      float p = ExpfRestatedWave(xv[i], s_expf);
      p /= sum_r[i];
      if (MODE == kTailSoftmaxLoglik) {
        if (p < 1.0e-20f) p = 1.0e-20f;                               // am.cc:109
        p = LogfRestatedWave(p, s_logf);                                   // am.cc:110
        p = (p + -1.0f * lp) * scale;                                  // am.cc:111, decodable.cc:15
      }
      const int r = r0 + wv * kPer + i;
      if (r < rows) __builtin_nontemporal_store(p, out + (int64_t)r * ld_out + c);
    }
  }
}

// ---- the same arithmetic with e kept ON CHIP (rows of at most 3 008 columns: the 3 000-pdf model of the
// BASELINE configs).  TailExactKernel reads every logit twice and exponentiates it twice because a
// 64-row block of e does not fit in LDS; it fits in REGISTERS: a workgroup of sixteen waves owns 32
// rows, wave w holds e of rows 2 w and 2 w + 1 (47 column tiles x 2 rows = 94 registers per lane;
// lane = column inside a tile, so a wave streams its two rows through memory 256 contiguous bytes at
// a time -- whole rows in order, the pattern HBM likes; the two-pass kernel touches 64 rows x 256 bytes
// per step and reaches 3 TB/s).  For the column-order sum each tile of e also passes through LDS
// (32 x 64, double-buffered) and ONE wave -- a different one per tile, so nobody falls behind -- adds
// it to the running sums (lane = row; the sums travel from duty wave to duty wave through LDS; one
// barrier per tile).  Then every wave finishes its own registers: e / sum, floor, logf, prior, scale --
// one read of the logits, one expf, one write: 24 kB/frame instead of 36.
// Resources (hipcc -Rpass-analysis=kernel-resource-usage, ROCm 7.2): 115 (probabilities) / 121
// (log-likelihoods) VGPRs of the 128 a 1024-thread workgroup may use, 0 AGPRs, ScratchSize 0 -- e[] does not
// spill; 20 / 24 SGPRs are spilled to VGPR lanes (no memory traffic).  Rows of 3 009 columns and more take
// TailExactKernel (the switch-over is a test case: N = 3008 | 3009).
constexpr int kRegTiles = 47;                          // 47 x 64 = 3 008 columns at most
constexpr int kRegWaves = 16;
constexpr int kRegRows = 2 * kRegWaves;                // rows per workgroup
constexpr int kRegGroup = 1;                           // column tiles per barrier (2 measured no faster: 2.43 vs 2.33 ms)

template <int MODE>
__global__ __launch_bounds__(64 * kRegWaves) void TailExactRegKernel(
    const float *__restrict__ in, int64_t ld_in, int rows, int n, const float *__restrict__ log_prior,
    float scale, float *__restrict__ out, int64_t ld_out) {
  __shared__ float tile[2][kRegGroup * kRegRows * kXLd];
  __shared__ float s_sum[kRegRows];
  __shared__ double s_logf[kLogfTableDoubles];
  __shared__ uint64_t s_expf[kExpfTableWords];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < kLogfTableDoubles) s_logf[threadIdx.x] = kLogfTab[threadIdx.x];
  if (threadIdx.x < kExpfTableWords) s_expf[threadIdx.x] = kExpfTab[threadIdx.x];
  if (threadIdx.x < kRegRows) s_sum[threadIdx.x] = 0.0f;
  const int r0 = blockIdx.x * kRegRows + 2 * wv;
  // this wave's two rows (clamped: a partial block recomputes the last row, its stores are masked).
  // Whole tiles are read: the launcher guarantees ld_in >= 64 ntiles (the logits rows are padded)
  const float *x0 = in + (int64_t)(r0 < rows ? r0 : rows - 1) * ld_in + lane;
  const float *x1 = in + (int64_t)(r0 + 1 < rows ? r0 + 1 : rows - 1) * ld_in + lane;
  const int ntiles = (n + kXCols - 1) / kXCols;         // <= kRegTiles (the launcher checks)
  __syncthreads();

  float e0[kRegTiles], e1[kRegTiles];
  float xa = __builtin_nontemporal_load(x0), xb = __builtin_nontemporal_load(x1);
  // kRegGroup tiles per barrier (sixteen waves meet at every barrier: one per tile cost 15 %)
#pragma unroll
  for (int jg = 0; jg < kRegTiles; jg += kRegGroup) {
    if (jg < ntiles) {                                  // uniform over the workgroup
      float *buf = tile[(jg / kRegGroup) & 1];
#pragma unroll
      for (int jj = 0; jj < kRegGroup; ++jj) {
        const int j = jg + jj;
        if (j < kRegTiles && j < ntiles) {
          const float ca = xa, cb = xb;
          if (j + 1 < ntiles) {
            xa = __builtin_nontemporal_load(x0 + (j + 1) * kXCols);
            xb = __builtin_nontemporal_load(x1 + (j + 1) * kXCols);
          }
          const bool live = j * kXCols + lane < n;      // false only in the last tile's padding columns
          const float ea = ExpfRestatedWave(ca, s_expf), eb = ExpfRestatedWave(cb, s_expf);
          e0[j] = live ? ea : 0.0f;                     // + 0 leaves a sum as it is
          e1[j] = live ? eb : 0.0f;
          float *dst = buf + jj * (kRegRows * kXLd) + (2 * wv) * kXLd + lane;
          dst[0] = e0[j];
          dst[kXLd] = e1[j];
        }
      }
      __syncthreads();
      if (wv == ((jg / kRegGroup) & (kRegWaves - 1)) && lane < kRegRows) {   // this group's duty wave: lane = row
        float s = s_sum[lane];
#pragma unroll
        for (int jj = 0; jj < kRegGroup; ++jj) {
          if (jg + jj < kRegTiles && jg + jj < ntiles) {
            const float *src = buf + jj * (kRegRows * kXLd) + lane * kXLd;
#pragma unroll
            for (int c = 0; c < kXCols; ++c) s += src[c];              // vector.cc:271
          }
        }
        s_sum[lane] = s;
      }
    }
  }
  __syncthreads();
  const float sum0 = s_sum[2 * wv], sum1 = s_sum[2 * wv + 1];
  float *o0 = out + (int64_t)r0 * ld_out + lane, *o1 = o0 + ld_out;
  const float *lpp = log_prior + lane;
  const bool st0 = r0 < rows, st1 = r0 + 1 < rows;
#pragma unroll
  for (int j = 0; j < kRegTiles; ++j) {
    if (j < ntiles) {
      const bool live = j * kXCols + lane < n;
      float p0 = e0[j] / sum0, p1 = e1[j] / sum1;                      // vector.cc:274-276
      if (MODE == kTailSoftmaxLoglik) {
        const float lp = live ? lpp[j * kXCols] : 0.0f;
        if (p0 < 1.0e-20f) p0 = 1.0e-20f;                              // am.cc:109
        if (p1 < 1.0e-20f) p1 = 1.0e-20f;
        p0 = LogfRestatedWave(p0, s_logf);                             // am.cc:110
        p1 = LogfRestatedWave(p1, s_logf);
        p0 = (p0 + -1.0f * lp) * scale;                                // am.cc:111, decodable.cc:15
        p1 = (p1 + -1.0f * lp) * scale;
      }
      if (live && st0) __builtin_nontemporal_store(p0, o0 + j * kXCols);
      if (live && st1) __builtin_nontemporal_store(p1, o1 + j * kXCols);
    }
  }
}

// The wave-per-row tail as a launch of its own (pk_tail_wave.h; the arithmetic of the fused last-layer launches of
// gemm.hip, so a model's log-likelihoods do not depend on which form ran).  Four waves per workgroup; the log priors
// sit in LDS.
template <int C, bool PAIR, bool EXACT, bool LDS_PRIOR>
__global__ __launch_bounds__(256) void TailWaveKernel(const float *__restrict__ in, int64_t ld_in, int rows, int n,
                                                      const float *__restrict__ log_prior, float scale,
                                                      float *__restrict__ out, int64_t ld_out) {
  extern __shared__ __attribute__((aligned(16))) float s_prior[];     // 16 floats of pair exchange, then (LDS_PRIOR) the table
  const int wave = threadIdx.x >> 6;
  const int per_wg = PAIR ? 2 : 4;                                    // rows a workgroup works on at a time
  const int slot = PAIR ? wave >> 1 : wave;
  const int step = gridDim.x * per_wg;
  const int iters = (rows + step - 1) / step;                          // the same for every wave of every workgroup
  if (LDS_PRIOR) {
    const int n4 = (n + 3) >> 2;
    for (int i = threadIdx.x; i < 4 * n4; i += blockDim.x) s_prior[16 + i] = i < n ? TailWavePriorEntry(log_prior[i], scale) : 0.0f;
    __syncthreads();
    TailWaveRows<C, false, PAIR, EXACT>(in, ld_in, blockIdx.x * per_wg + slot, step, iters, rows, n,
                                        TailWaveLdsPrior{(const twf4 __attribute__((address_space(3))) *)(s_prior + 16)},
                                        scale, out, ld_out, threadIdx.x & 63, wave & 1, s_prior + 4 * slot);
  } else {
    TailWaveRows<C, false, PAIR, EXACT>(in, ld_in, blockIdx.x * per_wg + slot, step, iters, rows, n,
                                        TailWaveGlobalPrior{log_prior, scale}, scale, out, ld_out, threadIdx.x & 63, wave & 1,
                                        s_prior + 4 * slot);
  }
}

template <int MODE>
static void LaunchTailMode(const float *in, int64_t ld_in, int rows, int n, const float *log_prior,
                           float scale, float *out, int64_t ld_out, hipStream_t stream) {
  const int cap = 16384;   // measured: 4096 -> 1.08 ms, 16384 -> 1.00 ms, one row per workgroup -> 1.17 ms
  dim3 grid(rows < cap ? rows : cap), block(kTailThreads);
  const int n4 = (n + 3) / 4;
  if (n4 <= kTailThreads)
    hipLaunchKernelGGL((TailKernel<MODE, 1>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
  else if (n4 <= 3 * kTailThreads)
    hipLaunchKernelGGL((TailKernel<MODE, 3>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
  else if (n4 <= kTailCacheMax * kTailThreads)
    hipLaunchKernelGGL((TailKernel<MODE, kTailCacheMax>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
  else
    hipLaunchKernelGGL((TailWideKernel<MODE>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
}

bool LaunchTailWave(const float *in, int64_t ld_in, int rows, int n, const float *log_prior, float scale, float *out,
                    int64_t ld_out, hipStream_t stream) {
  const int n4 = (n + 3) / 4;
  if (n4 > 64 * kTailWaveMaxChunks) return false;
  if (rows <= 0 || n <= 0) return true;
  const bool pair = TailWavePair(n);
  const int per_wg = pair ? 2 : 4;
  const int wgs = (rows + per_wg - 1) / per_wg < 8192 ? (rows + per_wg - 1) / per_wg : 8192;
  dim3 grid(wgs), block(256);
  // few rows per workgroup (a single utterance): the log priors straight from global memory, no table copy
  const bool lds_prior = rows > 8 * wgs * per_wg;
  const size_t lds = sizeof(float) * (16 + (lds_prior ? 4 * (size_t)n4 : 0));
#define PK_TW(CC, P, X) do { if (lds_prior) hipLaunchKernelGGL((TailWaveKernel<CC, P, X, true>), grid, block, lds, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out); \
                             else hipLaunchKernelGGL((TailWaveKernel<CC, P, X, false>), grid, block, lds, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out); } while (0)
  const int chunks = TailWaveChunks(n);
  if (TailWaveExact(n)) {
    if (chunks == 4) PK_TW(4, false, true);
    else if (chunks == 8) PK_TW(8, false, true);
    else if (chunks == 12) PK_TW(12, false, true);
    else if (chunks == 16) PK_TW(16, false, true);
    else if (chunks == 24) PK_TW(12, true, true);
    else PK_TW(16, true, true);
  } else if (chunks <= 4) PK_TW(4, false, false);
  else if (chunks <= 8) PK_TW(8, false, false);
  else if (chunks <= 12) PK_TW(12, false, false);
  else if (chunks <= 16) PK_TW(16, false, false);
  else if (chunks <= 24) PK_TW(12, true, false);
  else PK_TW(16, true, false);
#undef PK_TW
  return true;
}

void LaunchTail(int mode, bool reference_exact, const float *in, int64_t ld_in, int rows, int n,
                const float *log_prior, float scale, float *out, int64_t ld_out, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return;
  static const bool two_pass_only = [] { const char *e = getenv("PK_MI355_EXACT_TAIL_TWO_PASS"); return e && e[0] == '1'; }();
  if (reference_exact && mode != kTailLoglik && n <= kRegTiles * kXCols && ld_in >= (n + kXCols - 1) / kXCols * kXCols &&
      !two_pass_only) {
    dim3 grid((rows + kRegRows - 1) / kRegRows), block(64 * kRegWaves);
    if (mode == kTailSoftmaxProb)
      hipLaunchKernelGGL((TailExactRegKernel<kTailSoftmaxProb>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
    else
      hipLaunchKernelGGL((TailExactRegKernel<kTailSoftmaxLoglik>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
    return;
  }
  if (reference_exact && mode != kTailLoglik) {
    dim3 grid((rows + kXRows - 1) / kXRows), block(64 * (kXProducers + 1));
    if (mode == kTailSoftmaxProb)
      hipLaunchKernelGGL((TailExactKernel<kTailSoftmaxProb>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
    else
      hipLaunchKernelGGL((TailExactKernel<kTailSoftmaxLoglik>), grid, block, 0, stream, in, ld_in, rows, n, log_prior, scale, out, ld_out);
    return;
  }
  switch (mode) {
    case kTailSoftmaxProb:
      LaunchTailMode<kTailSoftmaxProb>(in, ld_in, rows, n, log_prior, scale, out, ld_out, stream);
      break;
    case kTailSoftmaxLoglik:
      LaunchTailMode<kTailSoftmaxLoglik>(in, ld_in, rows, n, log_prior, scale, out, ld_out, stream);
      break;
    default:
      LaunchTailMode<kTailLoglik>(in, ld_in, rows, n, log_prior, scale, out, ld_out, stream);
      break;
  }
}

}  // namespace pkmi
