// gemm.hip -- the affine layers (nnet.cc:22-36 -> matrix.cc:418-436 -> gemm.cc /
// gemm_haswell.cc) as an fp32 MFMA GEMM for gfx950.
//
//   C[i][j] = sum_k P[k][i] * Q[k][j]  (+ bias) (ReLU)
//
// Both operands are "k-major" panels (the output index is the contiguous one), so
// a 16 x 128 slab of either is a set of 512-byte rows.  Slabs go HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write), the LDS image
// [k][128] is read conflict-free by the MFMA operand pattern, nothing is ever
// transposed.  Hidden layers run with P = W^T ([in][out]) and Q = activations
// ([feature][frame]) and store the next layer's [feature][frame] panel directly;
// the last affine layer swaps the roles and stores frame-major rows for the softmax.
//
// Numerics: v_mfma_f32_32x32x2_f32 is a k-ordered chain of f32 fused
// multiply-adds, one accumulator per output element -- the same chain the
// reference's AVX2 micro-kernel builds (gemm_haswell.cc:122-282).  The reference
// blocks k by KC = 512 (gemm.h:50) and adds the chunks through C (gemm.cc:100),
// so we restart the accumulator every 512 k's and add the finished chunk into a
// second register set: results are bit-identical to the reference's SGEMM.
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave =
// 2 x 2 MFMA tiles), BK = 16, a ring of three LDS slabs (48 KiB) with two slabs of
// DMA in flight, one raw s_barrier per slab.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "pk_kernels.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kSlab = kBK * kTile;          // floats per operand per slab
constexpr int kRing = 3;                    // LDS slabs in the ring
constexpr int kDmaPerWave = 4;              // x4 DMA instructions one wave issues per slab

typedef const __attribute__((address_space(1))) void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;

// One slab (16 k-rows of both operands) goes into ring slot `slot` as kDma DMA
// "pieces" per wave, so that the pieces can be spread between MFMAs.  A x4
// wave-instruction moves 64 lanes x 16 B = two 512-byte k-rows; wave w owns rows
// 4w..4w+3 of P and of Q.  The LDS destination is wave-uniform base + lane * 16
// (hardware), the global source is per lane.
template <bool SPLICE, bool SPLICE_X4>
__device__ __forceinline__ void IssuePiece(const GemmArgs &a, const float *__restrict__ pg,
                                           const float *__restrict__ qg, int k0, int wave, int lane,
                                           float *smem, int slot, int piece) {
  float *ps = smem + (slot * 2 + 0) * kSlab;
  float *qs = smem + (slot * 2 + 1) * kSlab;
  const int lrow = lane >> 5, lcol = (lane & 31) * 4;
  if (piece < 2) {                                    // P rows, x4
    const int row = wave * 4 + piece * 2;             // wave-uniform
    const int k = k0 + row + lrow;
    __builtin_amdgcn_global_load_lds((GlobalPtr)(pg + (int64_t)k * a.ldp + lcol),
                                     (LdsPtr)(ps + row * kTile), 16, 0, 0);
  } else if (!SPLICE || SPLICE_X4) {                  // Q rows, x4
    const int row = wave * 4 + (piece - 2) * 2;
    const int k = k0 + row + lrow;
    if (!SPLICE) {
      __builtin_amdgcn_global_load_lds((GlobalPtr)(qg + (int64_t)k * a.ldq + lcol),
                                       (LdsPtr)(qs + row * kTile), 16, 0, 0);
    } else {
      // am.cc:65-88 without materialising: operand row k is feature (k % D) of the
      // padded feature-major matrix, shifted by (k / D) frames (4-byte aligned source).
      const int c = k / a.splice_dim, d = k - c * a.splice_dim;
      __builtin_amdgcn_global_load_lds((GlobalPtr)(qg + (int64_t)d * a.ldq + c + lcol),
                                       (LdsPtr)(qs + row * kTile), 16, 0, 0);
    }
  } else {                                            // spliced Q rows, dword DMA: 64 frames of one k-row
    const int row = wave * 4 + (piece - 2) / 2;
    const int k = k0 + row;
    const int c = k / a.splice_dim, d = k - c * a.splice_dim;
    const int off = ((piece - 2) & 1) * 64;
    __builtin_amdgcn_global_load_lds((GlobalPtr)(qg + (int64_t)d * a.ldq + c + off + lane),
                                     (LdsPtr)(qs + row * kTile + off), 4, 0, 0);
  }
}

template <bool SPLICE, bool SPLICE_X4>
constexpr int DmaCount() { return (SPLICE && !SPLICE_X4) ? 2 + 8 : kDmaPerWave; }   // pieces per wave per slab

template <bool SPLICE, bool SPLICE_X4, bool MULTICHUNK, bool BIAS_J, bool RELU>
__global__ __launch_bounds__(kThreads, 2) void GemmKernel(GemmArgs a) {
  // ALL LDS in one array (a second __shared__ object makes hipcc drain the DMA
  // queue before every LDS read)
  __shared__ __attribute__((aligned(16))) float smem[kRing * 2 * kSlab];

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
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const float *pg = a.P + i0;
  const float *qg = a.Q + j0;

  f32x16 acc[2][2], done[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      acc[x][y] = f32x16{0};
      done[x][y] = f32x16{0};
    }

  const int nkt = a.K / kBK;
  constexpr int kDma = DmaCount<SPLICE, SPLICE_X4>();

  auto issue_slab = [&](int kt, int slot) {
#pragma unroll
    for (int p = 0; p < kDma; ++p)
      IssuePiece<SPLICE, SPLICE_X4>(a, pg, qg, kt * kBK, wave, lane, smem, slot, p);
  };
  // fragments of k-steps [first, first+4) of the slab in `slot`
  auto read_frags = [&](int slot, int first, float (&pf)[kBK / 4][2], float (&qf)[kBK / 4][2]) {
    const float *ps = smem + (slot * 2 + 0) * kSlab + wi * 64 + l31 + half * kTile;
    const float *qs = smem + (slot * 2 + 1) * kSlab + wj * 64 + l31 + half * kTile;
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      const int kk = 2 * (first + ks) * kTile;
      pf[ks][0] = ps[kk]; pf[ks][1] = ps[kk + 32];
      qf[ks][0] = qs[kk]; qf[ks][1] = qs[kk + 32];
    }
  };
  auto mfma4 = [&](const float (&pf)[kBK / 4][2], const float (&qf)[kBK / 4][2], int ks) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ks][0], qf[ks][0], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ks][0], qf[ks][1], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ks][1], qf[ks][0], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ks][1], qf[ks][1], acc[1][1], 0, 0, 0);
  };

  // ---- software pipeline.  Per slab kt (A = k-steps 0..3, B = k-steps 4..7):
  //   16 MFMA on A(kt) with the reads of frags B(kt) and the DMA pieces of slab kt+2 between them
  //   | wait DMA(kt+1), barrier | read frags A(kt+1) | 16 MFMA on B(kt)
  // Two workgroups share each SIMD and run in lockstep, so anything outside the MFMA
  // stream is idle matrix-pipe time: DMA issue and fragment reads are tucked under
  // MFMAs, and only the barrier itself is exposed.
  // Slot reuse: DMA(kt+2) overwrites the slot of slab kt-1, whose last reads every
  // wave completed (lgkmcnt(0)) before the barrier of slab kt-1.
  float fa_p[kBK / 4][2], fa_q[kBK / 4][2], fb_p[kBK / 4][2], fb_q[kBK / 4][2];
  issue_slab(0, 0);
  if (nkt > 1) {
    issue_slab(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0, fa_p, fa_q);

  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    int slot1 = slot + 1 == kRing ? 0 : slot + 1;
    int slot2 = slot1 + 1 == kRing ? 0 : slot1 + 1;
    const bool dma = kt + 2 < nkt;
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      mfma4(fa_p, fa_q, ks);
      __builtin_amdgcn_sched_barrier(0);
      // B fragments are fetched behind the first MFMAs (issued before them, hipcc
      // waits for them -- lgkmcnt(0) -- ahead of the first MFMA)
      if (ks == 0) read_frags(slot, kBK / 4, fb_p, fb_q);
      if (dma) {
#pragma unroll
        for (int p = ks * kDma / 4; p < (ks + 1) * kDma / 4; ++p)
          IssuePiece<SPLICE, SPLICE_X4>(a, pg, qg, (kt + 2) * kBK, wave, lane, smem, slot2, p);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // slab kt+1 must have landed (this wave's DMAs; the barrier covers the others');
    // slab kt+2 stays in flight across the barrier.  Raw s_barrier: __syncthreads()
    // would drain the DMA queue.
    // (the LDS wait is the builtin form so that hipcc's own wait bookkeeping sees it:
    // after an asm wait it re-waits lgkmcnt(0) behind the next reads)
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0) only
    if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(slot1, 0, fa_p, fa_q);   // unconditional (stale LDS on the last slab, unused): a branch
                                        // here makes hipcc wait lgkmcnt(0) ahead of the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) mfma4(fb_p, fb_q, ks);

    if (MULTICHUNK && kt + 1 < nkt && ((kt + 1) * kBK) % kChunkK == 0) {
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
    __builtin_amdgcn_sched_barrier(0);
    slot = slot1;
  }

  // ---- epilogue: D[i][j], column j on the lane, rows i in the 16 registers
  // (i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); 128-byte row segments per store.
  // The bias values are fetched in one batch (a per-element runtime select makes
  // hipcc branch around every load and wait for each one).
  float bj[2] = {0.0f, 0.0f};
  float bi[2][16];
  if (BIAS_J) {
#pragma unroll
    for (int y = 0; y < 2; ++y) bj[y] = a.bias[j0 + wj * 64 + y * 32 + l31];
  } else {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        bi[x][r] = a.bias[i0 + wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int jj = j0 + wj * 64 + y * 32 + l31;
      float *orow = a.out + (int64_t)(i0 + wi * 64 + x * 32 + 4 * half) * a.ldo + jj;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[x][y][r];
        if (MULTICHUNK) v = done[x][y][r] + v;
        v += BIAS_J ? bj[y] : bi[x][r];                // nnet.cc:32-35
        if (RELU) v = v < 0.0f ? 0.0f : v;             // nnet.cc:56-58
        orow[(int64_t)((r & 3) + 8 * (r >> 2)) * a.ldo] = v;
      }
    }
}

template <bool SPLICE, bool SPLICE_X4, bool MULTICHUNK>
void LaunchVariant(const GemmArgs &a, dim3 grid, dim3 block, hipStream_t stream) {
  if (a.bias_on_j) {
    if (a.relu) hipLaunchKernelGGL((GemmKernel<SPLICE, SPLICE_X4, MULTICHUNK, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<SPLICE, SPLICE_X4, MULTICHUNK, true, false>), grid, block, 0, stream, a);
  } else {
    if (a.relu) hipLaunchKernelGGL((GemmKernel<SPLICE, SPLICE_X4, MULTICHUNK, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<SPLICE, SPLICE_X4, MULTICHUNK, false, false>), grid, block, 0, stream, a);
  }
}

}  // namespace

void LaunchGemm(const GemmArgs &a, hipStream_t stream) {
  const int super_i = (a.tiles_i + 7) / 8, super_j = (a.tiles_j + 7) / 8;
  const int nblk = super_i * super_j * 64;
  const bool multi = a.K > kChunkK;
  dim3 grid(nblk), block(kThreads);
  static const bool splice_x4 = getenv("PK_MI355_SPLICE_DWORD") == nullptr;
  if (a.splice_dim > 0) {
    if (splice_x4) {
      if (multi) LaunchVariant<true, true, true>(a, grid, block, stream);
      else LaunchVariant<true, true, false>(a, grid, block, stream);
    } else {
      if (multi) LaunchVariant<true, false, true>(a, grid, block, stream);
      else LaunchVariant<true, false, false>(a, grid, block, stream);
    }
  } else {
    if (multi) LaunchVariant<false, false, true>(a, grid, block, stream);
    else LaunchVariant<false, false, false>(a, grid, block, stream);
  }
}

}  // namespace pkmi
