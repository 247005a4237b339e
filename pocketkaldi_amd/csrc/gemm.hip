// gemm.hip -- the affine layers (nnet.cc:22-36 -> matrix.cc:418-436 -> gemm.cc /
// gemm_haswell.cc) as an fp32 MFMA GEMM for gfx950.
//
//   C[i][j] = sum_k P[k][i] * Q[k][j]  (+ bias) (ReLU)
//
// Both operands are "k-major" panels (the output index is the contiguous one), so
// a 16 x 128 slab of either is a set of 512-byte rows.  Slabs go HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) in its scalar-base
// form (pk_dma.h: row base in SGPRs + a 32-bit lane offset; 64-bit per-lane addresses cost
// 6 % of the kernel), the LDS image [k][128] is read conflict-free by the MFMA operand
// pattern, nothing is ever transposed.  Hidden layers run with P = W^T ([in][out]) and Q = activations
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
// 2 x 2 MFMA tiles with interleaved rows), BK = 16, a ring of three LDS slabs
// (48 KiB) with two slabs of DMA in flight, one raw s_barrier per slab, three
// workgroups per CU.  When a launch has too few 128 x 128 tiles to fill the chip
// (a single utterance: 8 x 8 tiles for 256 CUs) the same kernel runs with 64 x 64
// tiles (one MFMA tile per wave, S = 1) -- four times the workgroups -- and, when k spans
// 2..4 chunks of 512, with one group of four waves per chunk (KG) inside the workgroup.
// The k-loop itself is straight-line code: no branches, no vector address arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pk_dma.h"
#include "pk_kernels.h"
#include "pk_tail_wave.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;

typedef const __attribute__((address_space(1))) void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;

// Geometry for S x S MFMA tiles per wave (S = 2: 128 x 128 block tile, S = 1: 64 x 64).
template <int S>
struct Geo {
  static constexpr int kBT = 64 * S;               // block tile edge
  static constexpr int kSlab = kBK * kBT;          // floats per operand per slab
  static constexpr int kPieceRows = 4 / S;         // k-rows one 1 KiB DMA piece covers
  static constexpr int kLanesPerRow = 16 * S;      // lanes (x 16 B) per k-row of a piece
  static constexpr int kDma = 2 * S;               // pieces one wave issues per slab (S for P, S for Q)
};

// One slab (16 k-rows of both operands) goes into ring slot `slot` as kDma DMA "pieces"
// per wave, so that the pieces can be spread between MFMAs.  A x4 wave-instruction moves
// 64 lanes x 16 B = 1 KiB = kPieceRows k-rows; wave w owns rows 4w..4w+3 of P and of Q.
// The LDS destination is wave-uniform base + lane * 16 (hardware), the global source is
// per lane.
template <int S, bool SPLICE>
__device__ __forceinline__ void IssuePiece(const GemmArgs &a, const float *__restrict__ pg,
                                           const float *__restrict__ qg, int k0, int wave, int lane,
                                           uint32_t lane_off_p, uint32_t lane_off_q, float *smem,
                                           int slot, int piece, int ushift = 0) {
  using G = Geo<S>;
  float *ps = smem + (slot * 2 + 0) * G::kSlab;
  float *qs = smem + (slot * 2 + 1) * G::kSlab;
  // Plain panels: wave-uniform 64-bit row base (scalar registers) + a loop-invariant
  // 32-bit per-lane byte offset, so a piece costs scalar address math only.
  if (piece < S) {                                    // P rows
    const int row = wave * 4 + piece * G::kPieceRows; // wave-uniform
    const char *base = reinterpret_cast<const char *>(pg + (int64_t)(k0 + row) * a.ldp);
    DmaScalarBase(ps + row * G::kBT, base, lane_off_p);
  } else if (!SPLICE) {                               // Q rows
    const int row = wave * 4 + (piece - S) * G::kPieceRows;
    const char *base = reinterpret_cast<const char *>(qg + (int64_t)(k0 + row) * a.ldq);
    DmaScalarBase(qs + row * G::kBT, base, lane_off_q);
  } else {
    // am.cc:65-88 without materialising: operand row k is feature (k % D) of the
    // padded feature-major matrix, shifted by (k / D) frames.  The source is only
    // 4-byte aligned; x4 LDS-DMA takes that on gfx950 (bit-exact in the parity tests).
    const int row = wave * 4 + (piece - S) * G::kPieceRows;
    const int k = k0 + row + lane / G::kLanesPerRow;
    const int c = k / a.splice_dim, d = k - c * a.splice_dim;
    // rows past the last context frame (K is padded to the slab) come from zeros, not from the frame behind the
    // context: their weights are zero, but 0 * NaN is NaN, and the reference has no such row
    const float *src = c < a.splice_ctx ? qg + (int64_t)d * a.ldq + c : a.splice_zero;
    DmaVectorAddr(qs + row * G::kBT, src + (lane % G::kLanesPerRow) * 4 + ushift);
  }
}

// KG > 1 (small launches only, S = 1): the workgroup has KG groups of four waves and group g
// takes the g-th 512-chunk of k -- the reference's own blocking (gemm.h:50) -- through its own
// LDS ring; the groups' accumulators are then added in chunk order (gemm.cc:95-123 adds chunk
// after chunk into C), so the result is the same bit pattern with KG times the waves in flight.
// R = LDS slabs in a group's ring; R - 1 slabs of DMA are in flight.  R = 3 everywhere: the big
// tiles spend 6 000 cycles on a slab with three workgroups per CU, and a deeper ring (R = 8)
// bought the small launches nothing either (measured; they are not latency-bound).
// The log-likelihood tail of the 128 finished rows of a row tile, by the workgroup that completed it (TAIL variant):
// pk_tail_wave.h, one wave per row, four rows at a time; sc1 loads (the rows were written by other CUs, possibly on
// other XCDs: from memory, never from a stale line of this CU's L1 or this XCD's L2).
template <int C, bool PAIR>
__device__ __forceinline__ void FusedTailTile(const GemmArgs &a, int i0, float *smem, int tid) {
  const int nrows = a.tail_rows - i0 < kTile ? a.tail_rows - i0 : kTile;       // workgroup-uniform
  if (nrows <= 0) return;
  float *s_prior = smem + 64;                                   // [16, 64): pair exchange; [64, ..): the log priors
  const int n4 = (a.tail_n + 3) >> 2;
  for (int i = tid; i < 4 * n4; i += kThreads) s_prior[i] = i < a.tail_n ? TailWavePriorEntry(a.tail_log_prior[i], a.tail_scale) : 0.0f;
  __syncthreads();
  const int wave = tid >> 6;
  const int per_wg = PAIR ? 2 : 4;
  const int slot = PAIR ? wave >> 1 : wave;
  TailWaveRows<C, true, PAIR, true>(a.out + (int64_t)i0 * a.ldo, a.ldo, slot, per_wg, (nrows + per_wg - 1) / per_wg, nrows, a.tail_n,
                                    TailWaveLdsPrior{(const twf4 __attribute__((address_space(3))) *)(s_prior)},
                                    a.tail_scale, a.tail_out + (int64_t)i0 * a.tail_ld, a.tail_ld,
                                    tid & 63, wave & 1, smem + 16 + 4 * slot);
}

template <int S, bool SPLICE, bool MULTICHUNK, bool BIAS_J, bool RELU, int KG = 1, int R = 3, bool TAIL = false>
__global__ __launch_bounds__(kThreads * KG, (KG == 1 && R == 3) ? 3 : (KG == 1 && R == 2) ? 4 : 1) void GemmKernel(GemmArgs a) {
  static_assert(!TAIL || (S == 2 && BIAS_J && !RELU && !SPLICE && KG == 1), "fused tail: frame-major logits of big tiles");
  using G = Geo<S>;
  constexpr int kBT = G::kBT, kSlab = G::kSlab, kDma = G::kDma;
  constexpr int kRing = R, kAhead = R - 1;
  static_assert((kAhead - 1) * kDma <= 63, "vmcnt is a 6-bit counter");
  static_assert(KG == 1 || (S == 1 && !SPLICE && !MULTICHUNK), "k-groups: small plain launches only");
  // ALL LDS in one array (a second __shared__ object makes hipcc drain the DMA
  // queue before every LDS read)
  __shared__ __attribute__((aligned(16))) float smem_all[KG * kRing * 2 * kSlab];

  // ---- workgroup -> tile.  Workgroups b and b+8 share an XCD (round-robin
  // dispatch), so give each XCD a contiguous run of ids, and walk ids through
  // 8 x 8 super-tiles: the 64 workgroups resident on one XCD then share 8 P
  // panels and 8 Q panels in that XCD's L2.  Placement affects speed only.
  // (args count 128-tiles; half_j: the small-tile launch covers ONE 64-column strip -- the ragged end of a wide layer)
  const int tiles_i = a.tiles_i * (2 / S), tiles_j = (S == 1 && a.half_j) ? 1 : a.tiles_j * (2 / S);
  const int nblk = gridDim.x;                       // multiple of 64
  const int b = blockIdx.x;
  const int wg = (b % 8) * (nblk / 8) + b / 8;
  const int super_i = (tiles_i + 7) / 8;
  const int walk_j = TAIL ? a.walk_j : 8;              // the tail variant's super-tiles span a whole row of tiles, so
                                                       // that rows complete all through the launch, not in its last part
  const int per = 8 * walk_j;
  const int s = wg / per, w = wg % per;
  const bool strip = S == 1 && a.half_j;               // one column of tiles: ids are row tiles
  const int ti = strip ? wg : (s % super_i) * 8 + (w % 8);
  const int tj = strip ? 0 : (s / super_i) * walk_j + (w / 8);
  if (ti >= tiles_i || tj >= tiles_j) return;
  const int i0 = ti * kBT, j0 = tj * kBT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KG == 1 ? 0 : wave_all >> 2;        // k-group of this wave
  const int wave = KG == 1 ? wave_all : (wave_all & 3);
  float *smem = smem_all + grp * (kRing * 2 * kSlab);
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const float *pg = a.P + i0 + (int64_t)grp * kChunkK * a.ldp;
  const float *qg = a.Q + j0 + (int64_t)grp * kChunkK * a.ldq;
  // byte offset of this lane inside a DMA piece (kPieceRows k-rows, 16 bytes per lane)
  const uint32_t lane_off_p = (uint32_t)(((lane / G::kLanesPerRow) * a.ldp + (lane % G::kLanesPerRow) * 4) * sizeof(float));
  const uint32_t lane_off_q = (uint32_t)(((lane / G::kLanesPerRow) * a.ldq + (lane % G::kLanesPerRow) * 4) * sizeof(float));

  f32x16 acc[S][S], done[S][S];
#pragma unroll
  for (int x = 0; x < S; ++x)
#pragma unroll
    for (int y = 0; y < S; ++y) {
      acc[x][y] = f32x16{0};
      done[x][y] = f32x16{0};
    }

  const int nkt = (KG == 1 ? a.K : kChunkK) / kBK;

  // spliced operand, compact rows: this lane's four columns sit `ushift` columns further right in Q than their number
  // says (the context pads of the utterances before them); constant per lane and tile
  int ushift = 0;
  if (SPLICE && a.splice_shift) ushift = a.splice_shift[(j0 >> 2) + (lane % G::kLanesPerRow)];
  auto issue_slab = [&](int kt, int slot) {
#pragma unroll
    for (int p = 0; p < kDma; ++p)
      IssuePiece<S, SPLICE>(a, pg, qg, kt * kBK, wave, lane, lane_off_p, lane_off_q, smem, slot, p, ushift);
  };
  // fragments of k-steps [first, first+4) of the slab in `slot`.  Sub-tile a of a wave
  // takes the rows I0 + S i' + a (i' = lane & 31), so the S values of P (and of Q) a lane
  // needs are adjacent: one LDS read each (8 bytes for S = 2), conflict-free, all k-steps
  // reachable through the instruction's immediate offset.
  auto read_frags = [&](int slot, int first, float (&pf)[kBK / 4][S], float (&qf)[kBK / 4][S]) {
    const float *ps = smem + (slot * 2 + 0) * kSlab + wi * (32 * S) + S * l31 + half * kBT;
    const float *qs = smem + (slot * 2 + 1) * kSlab + wj * (32 * S) + S * l31 + half * kBT;
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      const int kk = 2 * (first + ks) * kBT;
      if (S == 2) {
        const f32x2 p = *reinterpret_cast<const f32x2 *>(ps + kk);
        const f32x2 q = *reinterpret_cast<const f32x2 *>(qs + kk);
        pf[ks][0] = p[0]; pf[ks][S - 1] = p[1];
        qf[ks][0] = q[0]; qf[ks][S - 1] = q[1];
      } else {
        pf[ks][0] = ps[kk];
        qf[ks][0] = qs[kk];
      }
    }
  };
  auto mfma_step = [&](const float (&pf)[kBK / 4][S], const float (&qf)[kBK / 4][S], int ks) {
#pragma unroll
    for (int x = 0; x < S; ++x)
#pragma unroll
      for (int y = 0; y < S; ++y)
        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ks][x], qf[ks][y], acc[x][y], 0, 0, 0);
  };

  // ---- software pipeline.  Per slab kt (A = k-steps 0..3, B = k-steps 4..7):
  //   MFMAs on A(kt) with the reads of frags B(kt) and the DMA pieces of slab kt+kAhead between them
  //   | wait DMA(kt+1), barrier | read frags A(kt+1) | MFMAs on B(kt)
  // Workgroups sharing a SIMD run in lockstep, so anything outside the MFMA stream is
  // idle matrix-pipe time: DMA issue and fragment reads are tucked under MFMAs, and only
  // the barrier itself is exposed.
  // Slot reuse: DMA(kt+kAhead) overwrites the slot of slab kt-1, whose last reads every
  // wave completed (lgkmcnt(0)) before the barrier of slab kt-1.
  float fa_p[kBK / 4][S], fa_q[kBK / 4][S], fb_p[kBK / 4][S], fb_q[kBK / 4][S];
#pragma unroll
  for (int s0 = 0; s0 < kAhead; ++s0)
    if (s0 < nkt) issue_slab(s0, s0);
  if (nkt >= kAhead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * kDma) : "memory");   // slab 0 is in
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0, fa_p, fa_q);

  // In the loop the plain panels are addressed through running row pointers (scalar registers,
  // one 64-bit add per piece and slab) instead of a 64-bit multiply per piece.
  const char *prun[S], *qrun[S];
#pragma unroll
  for (int p = 0; p < S; ++p) {
    const int64_t row = kAhead * kBK + wave * 4 + p * G::kPieceRows;
    prun[p] = reinterpret_cast<const char *>(pg + row * a.ldp);
    qrun[p] = reinterpret_cast<const char *>(qg + row * a.ldq);
  }
  const int64_t pstride = (int64_t)kBK * a.ldp * sizeof(float), qstride = (int64_t)kBK * a.ldq * sizeof(float);
  auto issue_running = [&](int slot_to, int p) {
    const int row = wave * 4 + (p < S ? p : p - S) * G::kPieceRows;
    if (p < S) {
      DmaScalarBase(smem + (slot_to * 2 + 0) * kSlab + row * kBT, prun[p], lane_off_p);
      prun[p] += pstride;
    } else {
      DmaScalarBase(smem + (slot_to * 2 + 1) * kSlab + row * kBT, qrun[p - S], lane_off_q);
      qrun[p - S] += qstride;
    }
  };

  // Spliced operand, 128-wide tiles: a piece is the two operand rows k, k + 1 = features
  // (k % D, (k + 1) % D) of the padded feature-major matrix shifted by (k / D, (k + 1) / D) frames
  // (am.cc:65-88).  The lower of the two row addresses is the scalar base, the other row's lanes
  // carry the distance; (k % D, k / D) advance by 16 per slab without a division.
  int sd[S], sc[S];
  if (SPLICE && S == 2) {
#pragma unroll
    for (int p = 0; p < S; ++p) {
      const int k = kAhead * kBK + wave * 4 + p * G::kPieceRows;
      sc[p] = k / a.splice_dim;
      sd[p] = k - sc[p] * a.splice_dim;
    }
  }
  const uint32_t ushift_bytes = (uint32_t)ushift * 4u;
  auto issue_splice = [&](int slot_to, int p) {       // p in [S, 2 S)
    const int j = p - S;
    const int row = wave * 4 + j * G::kPieceRows;
#ifdef PK_EXP_SPLICE_ALIGNED
    const int d0 = sd[j], c0 = sc[j] & ~3;            // TIMING ONLY (wrong results): every shift a multiple of four columns = 16-byte aligned rows
#else
    const int d0 = sd[j], c0 = sc[j];
#endif
    const bool wrap = d0 + 1 == a.splice_dim;
    // (operand rows past the last context frame -- K is padded to the slab -- come from zeros: see IssuePiece)
    const int64_t zoff = a.splice_zero - qg;
    const int64_t in0 = (int64_t)d0 * a.ldq + c0;
    const int64_t off0 = c0 < a.splice_ctx ? in0 : zoff;
    const int64_t off1 = (wrap ? c0 + 1 : c0) < a.splice_ctx ? (wrap ? (int64_t)c0 + 1 : in0 + a.ldq) : zoff;
    const int64_t lo = off1 < off0 ? off1 : off0;
    const uint32_t dist = (uint32_t)((off1 < off0 ? off0 - off1 : off1 - off0) * sizeof(float));
    const bool second = lane >= G::kLanesPerRow;       // this lane fetches row k + 1
    const uint32_t voff = (lane % G::kLanesPerRow) * 16 + ((second != (off1 < off0)) ? dist : 0u) + ushift_bytes;
    DmaScalarBase(smem + (slot_to * 2 + 1) * kSlab + row * kBT, reinterpret_cast<const char *>(qg + lo), voff);
    sd[j] += kBK;
    while (sd[j] >= a.splice_dim) { sd[j] -= a.splice_dim; ++sc[j]; }   // (once, unless the dimension is below 16)
  };

  // The bias values are fetched in one batch (a per-element runtime select makes hipcc branch
  // around every load and wait for each one).  The small tiles (S = 1: launches of a few
  // microseconds, one or two waves per SIMD) request them here, in front of the k loop, so that
  // the round trip to L2 is not added to the epilogue; the big tiles have no registers to park
  // them in (three workgroups per CU) and other workgroups to cover for them.
  const int I0 = i0 + wi * (32 * S), J0 = j0 + wj * (32 * S);
  float bj[S];
  float bi[S][16];
  auto load_bias = [&]() {
    if (BIAS_J) {
#pragma unroll
      for (int y = 0; y < S; ++y) bj[y] = a.bias[J0 + S * l31 + y];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int x = 0; x < S; ++x)
          bi[x][r] = a.bias[I0 + S * ((r & 3) + 8 * (r >> 2) + 4 * half) + x];
    }
  };
  if (S == 1) load_bias();

  // The slabs that still have DMA to issue and the last kAhead ones run through two copies of
  // the body, so that "is there a slab to fetch" is never a run-time predicate in the loop
  // (hipcc turned it into vector compares that write a register the MFMAs are still reading).
  const bool skip_last_half = SPLICE && a.K - a.splice_ctx * a.splice_dim >= kBK / 2;   // (wave-uniform, from arguments)
  int slot = 0;
  auto slab_step = [&](int kt, auto dma_tag) {
    constexpr bool dma = decltype(dma_tag)::value;
    int slot1 = slot + 1 == kRing ? 0 : slot + 1;
    int slot2 = slot == 0 ? kRing - 1 : slot - 1;     // slab kt + kAhead takes the slot of slab kt - 1
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      mfma_step(fa_p, fa_q, ks);
      __builtin_amdgcn_sched_barrier(0);
      // B fragments are fetched behind the first MFMAs (issued before them, hipcc
      // waits for them -- lgkmcnt(0) -- ahead of the first MFMA)
      if (ks == 0) read_frags(slot, kBK / 4, fb_p, fb_q);
      if (dma) {
#pragma unroll
        for (int p = ks * kDma / 4; p < (ks + 1) * kDma / 4; ++p) {
          if (SPLICE && p >= S && S == 2) issue_splice(slot2, p);
          else if (SPLICE && p >= S) IssuePiece<S, SPLICE>(a, pg, qg, (kt + kAhead) * kBK, wave, lane, lane_off_p, lane_off_q, smem, slot2, p, ushift);
          else issue_running(slot2, p);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // slab kt+1 must have landed (this wave's DMAs; the barrier covers the others');
    // the later slabs stay in flight across the barrier.  Raw s_barrier: __syncthreads()
    // would drain the DMA queue.
    // (the LDS wait is the builtin form so that hipcc's own wait bookkeeping sees it:
    // after an asm wait it re-waits lgkmcnt(0) behind the next reads)
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0) only
    if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * kDma) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(slot1, 0, fa_p, fa_q);   // unconditional (stale LDS on the last slab, unused): a branch
                                        // here makes hipcc wait lgkmcnt(0) ahead of the MFMAs below
    asm volatile("" ::: "memory");      // keeps the reads HERE: LLVM otherwise sinks them to the end of
                                        // the body, next to their first use
    __builtin_amdgcn_sched_barrier(0);
    // spliced operand: when the last slab's second half is nothing but K padding (K = 440: rows 440..447), its sixteen
    // MFMAs would add 0 x 0 to every accumulator -- the reference has no such terms; skip them (a scalar branch, last
    // slab only)
    if (!(SPLICE && !dma && skip_last_half && kt == nkt - 1)) {
#pragma unroll
      for (int ks = 0; ks < kBK / 4; ++ks) mfma_step(fb_p, fb_q, ks);
    }

    if (MULTICHUNK && kt + 1 < nkt && ((kt + 1) * kBK) % kChunkK == 0) {
      // gemm.cc:95-123: a finished 512-chunk is added into C (first chunk: stored)
      const bool first = (kt + 1) * kBK == kChunkK;
#pragma unroll
      for (int x = 0; x < S; ++x)
#pragma unroll
        for (int y = 0; y < S; ++y) {
          done[x][y] = first ? acc[x][y] : done[x][y] + acc[x][y];
          acc[x][y] = f32x16{0};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    slot = slot1;
  };
  int kt = 0;
  for (; kt + kAhead < nkt; ++kt) slab_step(kt, std::true_type());
  for (; kt < nkt; ++kt) slab_step(kt, std::false_type());

  if (KG > 1) {
    // every wave is past its last barrier, so no slab is read any more: the rings become the
    // exchange area, [group - 1][wave][register][lane]
    float *xch = smem_all;
    if (grp > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(((grp - 1) * 4 + wave) * 16 + r) * 64 + lane] = acc[0][0][r];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][0][r] + xch[(((g - 1) * 4 + wave) * 16 + r) * 64 + lane];
  }

  // ---- epilogue.  Accumulator acc[x][y], register r, lane (l31, half) holds
  // D[I0 + S i' + x][J0 + S l31 + y] with i' = (r & 3) + 8 (r >> 2) + 4 half: the y values
  // are adjacent in memory, so (S = 2) a row is written as 32 lanes x 8 bytes = 256
  // contiguous bytes per store.
  if (S == 2) load_bias();        // (S = 1 fetched it before the k loop)
#pragma unroll
  for (int x = 0; x < S; ++x) {
    float *obase = a.out + (int64_t)(I0 + S * 4 * half + x) * a.ldo + J0 + S * l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v[S];
#pragma unroll
      for (int y = 0; y < S; ++y) {
        float t = acc[x][y][r];
        if (MULTICHUNK) t = done[x][y][r] + t;
        t += BIAS_J ? bj[y] : bi[x][r];                // nnet.cc:32-35
        if (RELU) t = t < 0.0f ? 0.0f : t;             // nnet.cc:56-58
        v[y] = t;
      }
      float *dst = obase + (int64_t)(S * ((r & 3) + 8 * (r >> 2))) * a.ldo;
      if (TAIL) {
        // write-through (agent-coherent) store: the rows are read back by whichever workgroup completes the row
        // tile, possibly on another XCD, whose L2 this store must not bypass silently
        const f32x2 vv = f32x2{v[0], v[S - 1]};
#ifdef PK_EXP_TAIL_NT
        asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(dst), "v"(vv) : "memory");
#else
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(vv) : "memory");
#endif
      } else if (S == 2) {
        *reinterpret_cast<f32x2 *>(dst) = f32x2{v[0], v[S - 1]};
      } else {
        dst[0] = v[0];
      }
    }
  }

  if (TAIL) {
    // Count this tile in once its stores have left (sc1 stores + a drained vmcnt + a relaxed device-scope counter:
    // placement-independent, no cache maintenance).  The workgroup that brings the row tile's count to tiles_j owns
    // the 128 finished rows.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned *mail = reinterpret_cast<unsigned *>(smem_all);        // the rings are idle now
    // (the thread id is rebuilt from the wave number, a scalar, and the lane count: keeping threadIdx.x alive across
    // the k loop costs the 167-register kernel a spill)
    const int tid2 = wave_all * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (tid2 == 0) {
      // (the host zeroes the counters in front of every launch: capi_exec.hip, RunLayers)
      mail[0] = __hip_atomic_fetch_add(a.row_done + ti, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned ticket = mail[0];
    __syncthreads();
    if (ticket != (unsigned)tiles_j - 1u) return;
#ifdef PK_MI355_DIAG
    if (a.dbg & 16) return;                                         // hand-off only: a measurement switch (results are NOT made)
    const unsigned long long ph_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int chunks = TailWaveChunks(a.tail_n);                    // the EXACT instantiations of tail.hip: LaunchTailWave
    if (chunks == 4) FusedTailTile<4, false>(a, i0, smem_all, tid2);
    else if (chunks == 8) FusedTailTile<8, false>(a, i0, smem_all, tid2);
    else if (chunks == 12) FusedTailTile<12, false>(a, i0, smem_all, tid2);
    else if (chunks == 16) FusedTailTile<16, false>(a, i0, smem_all, tid2);
    else if (chunks == 24) FusedTailTile<12, true>(a, i0, smem_all, tid2);
    else FusedTailTile<16, true>(a, i0, smem_all, tid2);
#ifdef PK_MI355_DIAG
    if (a.dbg_counters && tid2 == 0) {                               // measurement: phases and their total duration (100 MHz ticks)
      atomicAdd(a.dbg_counters, __builtin_amdgcn_s_memrealtime() - ph_t0);
      atomicAdd(a.dbg_counters + 1, 1ull);
    }
#endif
  }
}

template <int S, bool SPLICE, bool MULTICHUNK, int KG = 1, int R = 3>
void LaunchVariant(const GemmArgs &a, dim3 grid, dim3 block, hipStream_t stream) {
  if constexpr (S == 2 && !SPLICE && KG == 1) {
    if (a.tail_out && a.bias_on_j && !a.relu) {
      hipLaunchKernelGGL((GemmKernel<S, SPLICE, MULTICHUNK, true, false, KG, R, true>), grid, block, 0, stream, a);
      return;
    }
  }
  if (a.bias_on_j) {
    if (a.relu) hipLaunchKernelGGL((GemmKernel<S, SPLICE, MULTICHUNK, true, true, KG, R>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<S, SPLICE, MULTICHUNK, true, false, KG, R>), grid, block, 0, stream, a);
  } else {
    if (a.relu) hipLaunchKernelGGL((GemmKernel<S, SPLICE, MULTICHUNK, false, true, KG, R>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((GemmKernel<S, SPLICE, MULTICHUNK, false, false, KG, R>), grid, block, 0, stream, a);
  }
}

template <int S>
void LaunchGeo(const GemmArgs &a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  const int ti = a.tiles_i * (2 / S), tj = (S == 1 && a.half_j) ? 1 : a.tiles_j * (2 / S);
  const bool tail = S == 2 && a.tail_out != nullptr;
  // the tail variant: one super-tile column = the whole row of tiles, so that rows complete all through the launch
  // (tail_walk > 0: an A/B switch, super-tile columns of 8 x tail_walk tiles walked row-super-tile first)
  a.walk_j = tail ? (a.tail_walk > 0 && a.tail_walk < tj ? a.tail_walk : tj) : 8;
#ifdef PK_MI355_DIAG
  // measurement hooks of the fused tail (diagnostic builds only: -DPK_MI355_DIAG; never in the product library)
  if (tail) {
    if (const char *de = getenv("PK_DEBUG_TAILFLAGS")) a.dbg = atoi(de);
    if (getenv("PK_DEBUG_TAILTIME")) {
      static unsigned long long *ctr = nullptr;
      static int calls = 0;
      if (!ctr) { (void)hipMalloc(&ctr, 16); (void)hipMemset(ctr, 0, 16); }
      a.dbg_counters = ctr;
      if (++calls % 16 == 0) {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[2];
        (void)hipMemcpy(h, ctr, 16, hipMemcpyDeviceToHost);
        fprintf(stderr, "fused tail phases so far: %llu, average %.1f us\n", h[1], h[1] ? h[0] * 0.01 / h[1] : 0.0);
      }
    }
  }
#endif
  const int super_i = (ti + 7) / 8, super_j = (tj + a.walk_j - 1) / a.walk_j;
  const int nblk = (S == 1 && a.half_j) ? (ti + 63) / 64 * 64 : (super_i * super_j * 8 * a.walk_j + 63) / 64 * 64;
  const bool multi = a.K > kChunkK;
  dim3 grid(nblk), block(kThreads);
  // a launch that leaves most SIMDs with one wave: one k-group per 512-chunk (KG x the waves)
  if (S == 1 && a.splice_dim == 0 && nblk <= 768 && a.K % kChunkK == 0) {
    const int kg = a.K / kChunkK;
    if (kg == 2) return LaunchVariant<1, false, false, 2>(a, grid, dim3(kThreads * 2), stream);
    if (kg == 3) return LaunchVariant<1, false, false, 3>(a, grid, dim3(kThreads * 3), stream);
    if (kg == 4) return LaunchVariant<1, false, false, 4>(a, grid, dim3(kThreads * 4), stream);
  }
  constexpr int kR = 3;
  if (a.splice_dim > 0) {
    if (multi) LaunchVariant<S, true, true, 1, kR>(a, grid, block, stream);
    // K <= 512 (no second accumulator set: 112 registers) leaves room for FOUR workgroups per CU if the ring has two slabs
    // instead of three (32 KiB of LDS each): the first layer's short tiles (27.5 slabs) hide their fixed part better behind
    // a fourth workgroup than behind a third slab of prefetch -- 1.84 -> 1.81 ms (profiles/r05_l1_ring_ab.txt)
    else if (S == 2 && a.ring == 2) LaunchVariant<S, true, false, 1, 2>(a, grid, block, stream);
    else LaunchVariant<S, true, false, 1, kR>(a, grid, block, stream);
  } else {
    if (multi) LaunchVariant<S, false, true, 1, kR>(a, grid, block, stream);
    else LaunchVariant<S, false, false, 1, kR>(a, grid, block, stream);
  }
}

}  // namespace

bool GemmFusesTail(const GemmArgs &a, int min_tiles) {
  if (min_tiles < 384) min_tiles = 384;                               // (below 384 tiles the small-tile kernel runs: no tail variant)
  return (int64_t)a.tiles_i * a.tiles_j >= min_tiles && a.bias_on_j && !a.relu && a.splice_dim == 0 && a.tail_n > 0 &&
         TailWaveExact(a.tail_n);      // (the fused form exists for rows of exactly 4 / 8 / 12 / 16 / 24 / 32 chunks of 256 columns)
}

void LaunchGemm(const GemmArgs &a, hipStream_t stream) {
  // fewer 128 x 128 tiles than ~1.5 per CU: quarter the tile, quadruple the workgroups
  if (a.half_j || a.tiles_i * a.tiles_j < 384) LaunchGeo<1>(a, stream);
  else LaunchGeo<2>(a, stream);
}

}  // namespace pkmi
