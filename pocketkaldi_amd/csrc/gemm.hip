// gemm.hip -- the affine layers (nnet.cc:22-36 -> matrix.cc:418-436 -> gemm.cc /
// gemm_haswell.cc) as an fp32 MFMA GEMM for gfx950.
//
//   C[i][j] = sum_k P[k][i] * Q[k][j]  (+ bias) (ReLU)
//
// Both operands are "k-major" panels (the output index is the contiguous one), so
// a 16 x 128 slab of either is a set of 512-byte rows: coalesced 16-byte loads,
// conflict-free LDS image [k][128], no transposes anywhere.  Hidden layers run
// with P = W^T ([in][out]) and Q = activations ([feature][frame]) and store the
// next layer's [feature][frame] panel directly; the last affine layer swaps the
// roles and stores frame-major rows for the softmax.
//
// Numerics: v_mfma_f32_32x32x2_f32 is a k-ordered chain of f32 fused
// multiply-adds, one accumulator per output element -- the same chain the
// reference's AVX2 micro-kernel builds (gemm_haswell.cc:122-282).  The reference
// blocks k by KC = 512 (gemm.h:50) and adds the chunks through C (gemm.cc:100),
// so we restart the accumulator every 512 k's and add the finished chunk into a
// second register set: results are bit-identical to the reference's SGEMM.
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave =
// 2 x 2 MFMA tiles), BK = 16, double-buffered LDS (32 KiB), two workgroups per CU.
#include <hip/hip_runtime.h>

#include "pk_kernels.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kSlab = kBK * kTile;          // floats per operand per stage

struct Stage {                               // one k-step of both operands in registers
  f32x4 p[2];
  f32x4 q[2];
};

template <bool SPLICE>
__device__ __forceinline__ void LoadStage(const GemmArgs &a, const float *__restrict__ pg,
                                          const float *__restrict__ qg, int k0, int kr, Stage *s) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = k0 + kr + 8 * h;
    s->p[h] = *reinterpret_cast<const f32x4 *>(pg + (int64_t)k * a.ldp);
    if (SPLICE) {
      // am.cc:65-88 without materialising: operand row k is feature (k % D) of the
      // padded feature-major matrix, shifted by (k / D) frames.  Only 4-byte aligned.
      const int c = k / a.splice_dim, d = k - c * a.splice_dim;
      const float *src = qg + (int64_t)d * a.ldq + c;
      s->q[h] = f32x4{src[0], src[1], src[2], src[3]};
    } else {
      s->q[h] = *reinterpret_cast<const f32x4 *>(qg + (int64_t)k * a.ldq);
    }
  }
}

__device__ __forceinline__ void StoreStage(float *ps, float *qs, int kr, int c4, const Stage &s) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    *reinterpret_cast<f32x4 *>(ps + (kr + 8 * h) * kTile + c4) = s.p[h];
    *reinterpret_cast<f32x4 *>(qs + (kr + 8 * h) * kTile + c4) = s.q[h];
  }
}

template <bool SPLICE, bool MULTICHUNK>
__global__ __launch_bounds__(kThreads, 2) void GemmKernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][kSlab];

  // ---- workgroup -> tile.  Workgroups b and b+8 share an XCD (round-robin
  // dispatch), so give each XCD a contiguous run of ids, and walk ids through
  // 8 x 8 super-tiles: the 64 workgroups resident on one XCD then share 8 P
  // panels and 8 Q panels in that XCD's L2.  Placement affects speed only.
  const int nblk = gridDim.x;                       // multiple of 64
  const int b = blockIdx.x;
  const int wg = (b % 8) * (nblk / 8) + b / 8;
  const int super_i = (a.tiles_i + 7) / 8;
  const int s = wg / 64, w = wg % 64;
  const int ti = (s % super_i) * 8 + (w % 8);
  const int tj = (s / super_i) * 8 + (w / 8);
  if (ti >= a.tiles_i || tj >= a.tiles_j) return;
  const int i0 = ti * kTile, j0 = tj * kTile;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;

  // global -> LDS staging role of this thread: rows kr and kr+8, 4 columns at c4
  const int kr = tid >> 5, c4 = (tid & 31) * 4;
  const float *pg = a.P + i0 + c4;
  const float *qg = a.Q + j0 + c4;

  f32x16 acc[2][2], done[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      acc[x][y] = f32x16{0};
      done[x][y] = f32x16{0};
    }

  const int nkt = a.K / kBK;
  Stage st;
  LoadStage<SPLICE>(a, pg, qg, 0, kr, &st);
  StoreStage(smem[0][0], smem[0][1], kr, c4, st);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) LoadStage<SPLICE>(a, pg, qg, (kt + 1) * kBK, kr, &st);

    const float *ps = smem[cur][0] + wi * 64 + l31;
    const float *qs = smem[cur][1] + wj * 64 + l31;
#pragma unroll
    for (int ks = 0; ks < kBK / 2; ++ks) {
      const int kk = (2 * ks + half) * kTile;
      const float p0 = ps[kk], p1 = ps[kk + 32];
      const float q0 = qs[kk], q1 = qs[kk + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q1, acc[1][1], 0, 0, 0);
    }

    if (MULTICHUNK && more && ((kt + 1) * kBK) % kChunkK == 0) {
      // gemm.cc:95-123: a finished 512-chunk is added into C (first chunk: stored)
      const bool first = (kt + 1) * kBK == kChunkK;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          done[x][y] = first ? acc[x][y] : done[x][y] + acc[x][y];
          acc[x][y] = f32x16{0};
        }
    }

    if (more) StoreStage(smem[cur ^ 1][0], smem[cur ^ 1][1], kr, c4, st);
    __syncthreads();
  }

  // ---- epilogue: D[i][j], column j on the lane, rows i in the 16 registers
  // (i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); 128-byte row segments per store.
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int jj = j0 + wj * 64 + y * 32 + l31;
      const float bj = a.bias_on_j ? a.bias[jj] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[x][y][r];
        if (MULTICHUNK && a.K > kChunkK) v = done[x][y][r] + v;
        v += a.bias_on_j ? bj : a.bias[ii];            // nnet.cc:32-35
        if (a.relu) v = v < 0.0f ? 0.0f : v;           // nnet.cc:56-58
        a.out[(int64_t)ii * a.ldo + jj] = v;
      }
    }
}

}  // namespace

void LaunchGemm(const GemmArgs &a, hipStream_t stream) {
  const int super_i = (a.tiles_i + 7) / 8, super_j = (a.tiles_j + 7) / 8;
  const int nblk = super_i * super_j * 64;
  const bool multi = a.K > kChunkK;
  dim3 grid(nblk), block(kThreads);
  if (a.splice_dim > 0) {
    if (multi) hipLaunchKernelGGL((GemmKernel<true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<true, false>), grid, block, 0, stream, a);
  } else {
    if (multi) hipLaunchKernelGGL((GemmKernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<false, false>), grid, block, 0, stream, a);
  }
}

}  // namespace pkmi
