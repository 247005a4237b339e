// gemm_f16.hip -- the affine layers on the fp16 matrix cores at fp32-class accuracy
// ("f16x3" precision mode; the default mode is the bit-exact fp32 kernel in gemm.hip).
//
// Every fp32 operand is carried as a pair of fp16 values, x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi)   (x - hi is exact in fp32), and
//
//   D[m][n] = sum_k Xhi Whi + Xhi Wlo + Xlo Whi          (Xlo Wlo ~ 2^-22, dropped)
//
// three fp16 MFMAs per block into ONE fp32 accumulator (fp16 products are exact in fp32; the
// cross terms sit 11 bits below the main term and fit the accumulator).  Two kernels share the
// layouts, the tile and the LDS-DMA ring described here:
//   GemmF16K32Kernel  v_mfma_f32_16x16x32_f16, k32 steps -- what f16x3 runs on (round 3; further down)
//   GemmF16Kernel     v_mfma_f32_32x32x16_f16, k16 steps -- the round-2 form, kept for the plain-fp16
//                     mode (one MFMA per product), which is bound by the operand stream, not by the clock.  Measured error vs the fp32 reference chain is ~1e-6
// relative on log-likelihoods -- inside the 1e-4 contract, not bit-exact.  Range: fp16
// saturates at 65504; the split clamps instead of overflowing.
//
// Layout (everything frame-major here): X = [rows][K] (rows = frames, k contiguous),
// W = [N][K] -- the model file's own [out][in] order -- D = [rows][N].  A row stores its
// (hi, lo) pairs interleaved in chunks of 8 k's: [hi k0..7][lo k0..7][hi k8..15][lo ...],
// so the 64 bytes one k16 step needs of a row are contiguous.  Layer 1 reads the CMVN
// output [frames][40] (5 chunks per frame, row stride 80 halves) with K = 440: the
// splice (am.cc:65-88) is again just an address function.
//
// Tile 256 x 256 per 512-thread workgroup (8 waves as 2 x 4, 128 x 64 per wave = 4 x 2
// MFMA tiles, 128 accumulator registers).  The k loop advances in k16 "half-slabs" of
// 32 KiB (256 X rows + 256 W rows x 64 bytes) through a ring of FOUR LDS buffers filled
// by LDS-DMA three half-slabs ahead (the DMA wait + barrier was where the waves sat:
// 33 % of their lifetime with a one-slab lead).  XOR swizzle on the DMA SOURCE address
// (linear LDS destination) and on the ds_read_b128 fragment reads (conflict-free).  The
// two column sub-tiles of a wave take interleaved columns, so a lane holds adjacent output
// columns; the epilogue stages each wave's rows through the (by then idle) LDS ring and
// writes them as whole 256-byte row pieces, 16 bytes per lane.
// (Measured alternative: 256 x 128 tiles, 4 waves, two workgroups per CU -- independent
// barriers, one tile's epilogue under the other's MFMAs -- 5 % slower: the operand traffic
// L2 -> LDS per MFMA is 1.5 x, and the chip is power-limited in this mode.)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "pk_dma.h"
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

#include "pk_kernels.h"

namespace pkmi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(1))) void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;

constexpr int kT = kTileF16;                  // 256: tile edge
constexpr int kStepK = 16;                    // k per half-slab = one MFMA k16 step
constexpr int kThreadsF16 = 512;
constexpr int kOperandBytes = kT * 64;        // one operand of one half-slab: 256 rows x 64 B
constexpr int kHalfSlabBytes = 2 * kOperandBytes;   // X rows, then W rows: 32 KiB
constexpr int kRingF16 = 4;
constexpr int kPiecesPerWave = 4;             // 32 x 1 KiB pieces per half-slab / 8 waves

// LDS row (0..255) of the W tile -> column n of the tile.  Within each block of 64
// columns sub-tile y of a wave owns columns 2 i' + y, stored as 32 consecutive LDS rows.
__device__ __forceinline__ int WRowToCol(int row) {
  return (row & ~63) + 2 * (row & 31) + ((row >> 5) & 1);
}

// byte offset of logical 16-byte position q of LDS row `row`; the four positions of a
// 64-byte row are [k0..7 hi][k0..7 lo][k8..15 hi][k8..15 lo]
__device__ __forceinline__ int SwzOff(int row, int q) {
  return row * 64 + ((q ^ ((row >> 2) & 3)) << 4);
}

struct SplitOut {
  _Float16 hi, lo;
};
__device__ __forceinline__ SplitOut Split(float v) {
  v = fminf(fmaxf(v, -65504.0f), 65504.0f);
  SplitOut s;
  s.hi = static_cast<_Float16>(v);
  s.lo = static_cast<_Float16>(v - static_cast<float>(s.hi));
  return s;
}

// 2^e as a float (|e| <= 126: the host clamps the exponents it stores)
__device__ __forceinline__ float Pow2(int e) { return __int_as_float((127 + e) << 23); }

// Range word of an operand: the largest |hi| some wave wrote, kept as the bit pattern of a non-negative float
// (unsigned order = float order).  One relaxed read + at most one fire-and-forget atomic per wave; the words of
// an operand are spread over kRangeSlots addresses so the atomics of a launch do not queue on one L2 line.
// The host reads them after the call: >= 65504 means the split clamped (saturation), a positive value below
// 2^-5 means every lo half of the operand was subnormal (capi_exec.hip: EvalRange).
__device__ __forceinline__ void PublishRange(uint32_t *range, int slot, float wave_max) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) wave_max = fmaxf(wave_max, __shfl_xor(wave_max, m));
  if ((threadIdx.x & 63) == 0) {
    const uint32_t bits = __float_as_uint(wave_max);
    uint32_t *w = range + (slot & (kRangeSlots - 1));
    if (bits > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, bits);
  }
}

// Running maximum of |hi| over packed halves, taken on the BIT PATTERNS as signed 16-bit integers (v_pk_max_i16:
// half an instruction per value, no NaN canonicalisation, no dependent fp latency): for non-negative halves the
// integer order is the float order, and -0.0 (0x8000) sorts below everything.  NONNEG: the values went through
// a ReLU; otherwise the sign bits are cleared first.
template <bool NONNEG>
__device__ __forceinline__ s16x4 RangeAcc(s16x4 m, f16x4 hi) {
  s16x4 b = __builtin_bit_cast(s16x4, hi);
  if (!NONNEG) b &= (short)0x7fff;
  return __builtin_elementwise_max(m, b);
}
template <bool NONNEG>
__device__ __forceinline__ s16x2 RangeAcc(s16x2 m, f16x2 hi) {
  s16x2 b = __builtin_bit_cast(s16x2, hi);
  if (!NONNEG) b &= (short)0x7fff;
  return __builtin_elementwise_max(m, b);
}
__device__ __forceinline__ float RangeValue(short bits) {      // the fp16 whose (non-negative) pattern won
  return static_cast<float>(__builtin_bit_cast(_Float16, bits));
}

// The three exponent words of a launch (wave-uniform, fetched before the first DMA is issued: left to the
// compiler they became vector loads at the top of the epilogue, a full memory round trip with the matrix pipes idle).
struct Exps { int e_in_w, e_out; };
__device__ __forceinline__ Exps LoadExps(const GemmF16Args &a, bool last) {
  Exps e;
  e.e_in_w = __builtin_amdgcn_readfirstlane(*a.e_in + *a.e_w);
  e.e_out = last ? 0 : __builtin_amdgcn_readfirstlane(*a.e_out);
  return e;
}

// Diagnostic build only (tools/ubench/f16_gemm_probe.hip defines PK_F16_STAMPS): s_memtime stamps
// around the three waits of a step, summed per wave.  In the product build PK_STAMP is nothing.
#ifdef PK_F16_STAMPS
#define PK_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define PK_STAMP(i) do { } while (0)
#endif

// TERMS = 3: the split-fp16 arithmetic above.  TERMS = 1 (PK_MI355_PRECISION_F16): the hi halves only --
// plain fp16 operands, one MFMA per product; same layouts (the lo halves travel unused), so the mode
// is bound by the L2 -> LDS operand stream rather than by the matrix pipes.
template <bool RELU, bool LAST, int TERMS>
__global__ __launch_bounds__(kThreadsF16, 2) void GemmF16Kernel(GemmF16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ring: 4 x 32 KiB
#ifdef PK_F16_STAMPS
  const long long st_kernel = __builtin_amdgcn_s_memtime();
#endif

  // XCD-aware tile walk, as in gemm.hip: contiguous ids per XCD, walk_m x walk_n (4 x 4) super-tiles
  const int nblk = gridDim.x;                       // multiple of 8
  const int b = blockIdx.x;
  const int wg = (b % 8) * (nblk / 8) + b / 8;
  const int super_m = (a.tiles_m + a.walk_m - 1) / a.walk_m;
  const int per = a.walk_m * a.walk_n;
  const int s = wg / per, w = wg % per;
  const int tm = (s % super_m) * a.walk_m + (w % a.walk_m);
  const int tn = (s / super_m) * a.walk_n + (w / a.walk_m);
  if (tm >= a.tiles_m || tn >= a.tiles_n) return;
  const int m0 = tm * kT, n0 = tn * kT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves
  const int l31 = lane & 31, kg = lane >> 5;
  const Exps ex = LoadExps(a, LAST);

  // ---- DMA role of this lane: pieces 2 wave, 2 wave + 1 of X and of W; a piece is 16
  // LDS rows, lane -> row (lane >> 2), stored position (lane & 3)
  // Source = wave-uniform base (SGPRs) + per-lane byte offset (32 bits): the rows of a piece
  // differ by lane >> 2 only, and the W row -> column map is affine inside a piece.
  const char *sbase[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row0 = (wave * 2 + p) * 16;
    sbase[p][0] = reinterpret_cast<const char *>(a.X + (int64_t)(m0 + row0) * a.ldx);
    sbase[p][1] = reinterpret_cast<const char *>(a.W + (int64_t)(n0 + WRowToCol(row0)) * a.ldw);
  }
  const int lr = lane >> 2;
  const int q = (lane & 3) ^ ((lr >> 2) & 3);       // logical position landing at this lane's slot
  const uint32_t voff[2] = {(uint32_t)(((int64_t)lr * a.ldx + q * 8) * sizeof(_Float16)),
                            (uint32_t)(((int64_t)2 * lr * a.ldw + q * 8) * sizeof(_Float16))};
  // compact rows (spliced first layer of the batch scorer): row r of this launch is row r + row_shift4[r / 4] of X
  uint32_t xoff[2] = {voff[0], voff[0]};
  if (a.row_shift4) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      xoff[p] += (uint32_t)((int64_t)a.row_shift4[(m0 + (wave * 2 + p) * 16 + lr) >> 2] * a.ldx * sizeof(_Float16));
  }
  // one of this wave's four DMA pieces of k16 step h (32 halves of every row): piece = 2 op + p
  auto issue_piece = [&](int h, int slot, int piece) {
    const int op = piece >> 1, p = piece & 1;
    unsigned char *dst = smem + slot * kHalfSlabBytes + op * kOperandBytes + (wave * 2 + p) * 1024;
    DmaScalarBase(reinterpret_cast<const float *>(dst), sbase[p][op] + h * 64, op == 0 ? xoff[p] : voff[1]);
  };
  auto issue_step = [&](int h, int slot) {
#pragma unroll
    for (int piece = 0; piece < kPiecesPerWave; ++piece) issue_piece(h, slot, piece);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f32x16{0};

  // fragment byte offsets inside a half-slab: A sub-tile x = LDS row wm*128 + 32x + l31 of
  // the X rows, B sub-tile y = LDS row wn*64 + 32y + l31 of the W rows; hi at logical
  // position 2 kg, lo at 2 kg + 1
  int aoff_h[4], aoff_l[4], boff_h[2], boff_l[2];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    aoff_h[x] = SwzOff(wm * 128 + 32 * x + l31, 2 * kg);
    aoff_l[x] = SwzOff(wm * 128 + 32 * x + l31, 2 * kg + 1);
  }
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    boff_h[y] = kOperandBytes + SwzOff(wn * 64 + 32 * y + l31, 2 * kg);
    boff_l[y] = kOperandBytes + SwzOff(wn * 64 + 32 * y + l31, 2 * kg + 1);
  }

  // ---- main loop over k16 steps h.  Ring of four half-slabs, DMA three steps ahead.  A
  // step is four groups (x = 0..3) of six MFMAs; the A fragments of the next group are
  // fetched while the current group's MFMAs issue.  Before the LAST group of step h:
  // every fragment of step h is in registers (lgkmcnt(0)), this wave's pieces of step
  // h+1 have landed (counted vmcnt), barrier.  Everything else sits UNDER the MFMA stream (an MFMA
  // occupies the matrix pipe for 32 cycles and the wave's issue port for 8 of them): the four DMA
  // pieces of step h+4 -- into step h's buffer, which nobody reads any more -- are spread over the
  // last group of step h and the first of step h+1, the first fragments of step h+1 are read behind
  // the first MFMA after the barrier.  (Issued back to back behind the barrier, by both waves of a
  // SIMD at once, the pieces cost 4-5 %.)
  const int nsteps = a.K / kStepK;
  issue_step(0, 0);
  if (nsteps > 1) issue_step(1, 1);
  if (nsteps > 2) issue_step(2, 2);
  if (nsteps > 3) issue_step(3, 3);
  if (nsteps > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kPiecesPerWave) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  f16x8 ah[2], al[2];          // A fragments: current group / next group
  f16x8 bh[2][2], bl[2][2];    // B fragments of the current / next step: [parity][y]
  auto read_a = [&](const unsigned char *base, int x, f16x8 &h, f16x8 &l) {
    h = *reinterpret_cast<const f16x8 *>(base + aoff_h[x]);
    if (TERMS == 3) l = *reinterpret_cast<const f16x8 *>(base + aoff_l[x]);
  };
  auto read_b = [&](const unsigned char *base, int par) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      bh[par][y] = *reinterpret_cast<const f16x8 *>(base + boff_h[y]);
      if (TERMS == 3) bl[par][y] = *reinterpret_cast<const f16x8 *>(base + boff_l[y]);
    }
  };
  read_b(smem, 0);
  read_a(smem, 0, ah[0], al[0]);

#ifdef PK_F16_STAMPS
  long long st_acc[4] = {0, 0, 0, 0};
  long long st_last = __builtin_amdgcn_s_memtime();
  const long long st_begin = st_last;
#endif
  auto mfma3 = [&](int x, int y, int cur, int par, int which) {
    // the three products of one (x, y) tile and k16 step: hi lo, lo hi, hi hi -- in this order
    if (TERMS == 3 && which == 0) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[par][y], acc[x][y], 0, 0, 0);
    if (TERMS == 3 && which == 1) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[par][y], acc[x][y], 0, 0, 0);
    if (which == 2) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[par][y], acc[x][y], 0, 0, 0);
  };
  // (Measured and dropped: waves 4-7 issuing their pieces in the second and third group instead,
  // so that the two waves of a SIMD never issue DMA at the same time: 1-2 % slower.  One barrier
  // per PAIR of steps, the next pair's DMA issued right behind it: 3.5 % slower -- the barrier wait
  // of the older wave of a SIMD is the time its partner needs the matrix pipe for, not lost time.)
  auto step = [&](int h, const int par) {             // par = h & 1, compile-time in the body
    const int slot = h & (kRingF16 - 1);
    const int prev_slot = (h - 1) & (kRingF16 - 1);
    const unsigned char *base = smem + slot * kHalfSlabBytes;
    const unsigned char *next = smem + ((h + 1) & (kRingF16 - 1)) * kHalfSlabBytes;
    const bool dma_tail = h >= 1 && h + 3 < nsteps;   // pieces 2, 3 of step h + 3 (its pieces 0, 1 went out in step h - 1)
    const bool dma_head = h + 4 < nsteps;             // pieces 0, 1 of step h + 4
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int cur = x & 1;
      if (x == 3) {
        PK_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): all of step h is in registers
        PK_STAMP(1);
        // this wave's pieces of step h+1 landed; steps h+2, h+3 may stay in flight
        if (h + 3 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPiecesPerWave) : "memory");
        else if (h + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PK_STAMP(2);
        __builtin_amdgcn_s_barrier();
        PK_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 0, cur, par, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_b(next, par ^ 1);                        // stale on the last step, unused
        read_a(next, 0, ah[cur ^ 1], al[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 0, cur, par, 1);
        mfma3(x, 0, cur, par, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (dma_head) issue_piece(h + 4, slot, 0);    // nobody reads step h's buffer any more
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 1, cur, par, 0);
        mfma3(x, 1, cur, par, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (dma_head) issue_piece(h + 4, slot, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 1, cur, par, 2);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        read_a(base, x + 1, ah[cur ^ 1], al[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 0, cur, par, 0);
        mfma3(x, 0, cur, par, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (x == 0 && dma_tail) issue_piece(h + 3, prev_slot, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 0, cur, par, 2);
        mfma3(x, 1, cur, par, 0);
        mfma3(x, 1, cur, par, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (x == 0 && dma_tail) issue_piece(h + 3, prev_slot, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma3(x, 1, cur, par, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // two steps per trip: the B-fragment parity is then a compile-time constant
  // (4 groups per step keep the A current/next parity aligned by themselves)
  for (int h = 0; h < nsteps; h += 2) {
    step(h, 0);
    if (h + 1 < nsteps) step(h + 1, 1);
  }

#ifdef PK_F16_STAMPS
  const long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: acc[x][y][r] = D[M0 + 32x + i'][N0 + y], N0 = n0 + wn*64 + 2 l31,
  // i' = (r & 3) + 8 (r >> 2) + 4 kg; bias (nnet.cc:32-35), ReLU (nnet.cc:56-58).
  // A lane holds two adjacent columns of 64 rows: stored from the registers that is 128 four-byte
  // stores per lane (the whole epilogue of a tile then costs as much as 18 k16 steps, with the
  // matrix pipes of the CU idle).  Instead each wave stages 32 rows (one x) at a time in its own
  // 2 x 8 KiB of the LDS ring -- idle now: every wave passed the last barrier with the last step's
  // fragments in registers, and all DMA has landed -- as the rows will lie in memory, 256 bytes
  // per row (64 fp32 logits, or 64 x (hi, lo) halves in chunks of 8), and writes them out 16 bytes
  // per lane: one store instruction = four whole 256-byte row pieces.  No barrier: a wave reads
  // back only what it wrote itself.
  const int M0 = m0 + wm * 128, N0 = n0 + wn * 64 + 2 * l31;
  // operand exponents (gemm_f16.hip header of GemmF16K32Kernel's epilogue): v = acc * 2^(e_out - e_in - e_w) + bias * 2^e_out
  const float acc_scale = Pow2(ex.e_out - ex.e_in_w);
  const f32x2 bias = *reinterpret_cast<const f32x2 *>(a.bias + N0) * Pow2(ex.e_out);
  s16x2 hmax = s16x2{0, 0};
  unsigned char *stage = smem + wave * (2 * 8192);
  // byte offset of this lane's pair in a staged row: fp32 pair, or the (hi, lo) halves of chunk l31 / 4
  const int pair_byte = LAST ? l31 * 8 : (l31 >> 2) * 32 + (l31 & 3) * 4;
  // this wave's 64 columns start at byte 256 * (column block) of an output row in both formats
  unsigned char *out_rows = LAST ? reinterpret_cast<unsigned char *>(a.out_f32 + (int64_t)M0 * a.ldo + n0 + wn * 64)
                                 : reinterpret_cast<unsigned char *>(a.out + (int64_t)M0 * a.ldo + 2 * (n0 + wn * 64));
  const int64_t row_bytes = a.ldo * (LAST ? (int64_t)sizeof(float) : (int64_t)sizeof(_Float16));
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    unsigned char *buf = stage + (x & 1) * 8192;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
      float v0 = __builtin_fmaf(acc[x][0][r], acc_scale, bias[0]), v1 = __builtin_fmaf(acc[x][1][r], acc_scale, bias[1]);
      if (RELU) {               // one v_max_f32 (folds into the split's clamp); NaN / -0.0 are not carried by this mode
        v0 = fmaxf(v0, 0.0f);
        v1 = fmaxf(v1, 0.0f);
      }
      if (LAST) {
        *reinterpret_cast<f32x2 *>(buf + row * 256 + pair_byte) = f32x2{v0, v1};
      } else {
        const SplitOut s0 = Split(v0), s1 = Split(v1);
        const f16x2 hi = f16x2{s0.hi, s1.hi};
        hmax = RangeAcc<RELU>(hmax, hi);
        *reinterpret_cast<f16x2 *>(buf + row * 256 + pair_byte) = hi;
        *reinterpret_cast<f16x2 *>(buf + row * 256 + pair_byte + 16) = f16x2{s0.lo, s1.lo};
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // this wave's LDS writes are in
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {                       // 8 KiB = 8 x (64 lanes x 16 bytes); lane -> row j*4 + lane/16
      const int row = j * 4 + (lane >> 4);
      const f32x4v v = *reinterpret_cast<const f32x4v *>(buf + row * 256 + (lane & 15) * 16);
      *reinterpret_cast<f32x4v *>(out_rows + (int64_t)(32 * x + row) * row_bytes + (lane & 15) * 16) = v;
    }
  }
  if (!LAST && a.range)
    PublishRange(a.range, blockIdx.x * 8 + wave, RangeValue(hmax[0] > hmax[1] ? hmax[0] : hmax[1]));
#ifdef PK_F16_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
    long long *o = pk_f16_stamps + (blockIdx.x * 8 + wave) * 8;
    o[0] = st_loop_end - st_begin;                       // k loop, cycles
    o[1] = st_acc[0];                                    // MFMA / issue segments (barrier release -> next lgkm wait)
    o[2] = st_acc[1];                                    // lgkmcnt(0) wait
    o[3] = st_acc[2];                                    // vmcnt wait
    o[4] = st_acc[3];                                    // barrier wait
    o[5] = st_begin - st_kernel;                         // prologue (start -> first fragments requested)
    o[6] = __builtin_amdgcn_s_memtime() - st_loop_end;   // epilogue (stores issued)
  }
#endif
}

// ======================================================================= 16 x 16 x 32 form
// GemmF16K32Kernel -- the same tile, ring, DMA and data layouts on v_mfma_f32_16x16x32_f16.
// Why: this chip holds its clock down under fp16 MFMA load, and the clock it holds depends on the
// MFMA shape (MI355X_MICROARCH.md, DVFS give-back item 7; tools/ubench/mfma_shape_probe.hip on this
// pool: 16x16x32 1.91 GHz / 1906 TFLOP/s against 1.65 GHz / 1688 for 32x32x16 in bare register
// loops at equal cycles per flop).  The CDNA3-era k16 forms run at half rate here (same probe), so
// the k32 of an instruction has to come from two k16 half-slabs:
//   * a STEP is a PAIR of half-slabs (k32); the ring of four half-slabs is two pair slots; one
//     barrier per step; the pair after next is fetched into the slot a step has just released;
//   * lane l of a fragment owns LDS row (l & 15) of its 16-row tile and k-group g = l >> 4:
//     g = 0, 1 -> k 0..7, 8..15 of the first half-slab, g = 2, 3 -> of the second.  A fragment is still
//     one ds_read_b128 per lane (hi at logical position 2 (g & 1), lo one further);
//   * the XOR swizzle becomes q ^ gray((row >> 2) & 3): with the four 16-lane groups of ds_read_b128
//     ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) every group then covers all 64 banks
//     (the plain (row >> 2) & 3 of the 32 x 32 form is 2-way for this access);
//   * a wave's 128 x 64 is 8 x 4 tiles of 16 x 16; the four column tiles take interleaved columns
//     (tile y owns columns 4 j + y of the wave's 64), so a lane holds four adjacent output columns of
//     four rows per row tile and stages them 16 bytes (fp32) or 8 + 8 bytes (hi, lo halves) at a time.
__device__ __forceinline__ int Gray2(int c) { return c ^ (c >> 1); }
__device__ __forceinline__ int WRowToCol16(int row) {
  return (row & ~63) + 4 * (row & 15) + ((row >> 4) & 3);
}
__device__ __forceinline__ int SwzOff16(int row, int q) {
  return row * 64 + ((q ^ Gray2((row >> 2) & 3)) << 4);
}

template <bool RELU, bool LAST, int TERMS>
__global__ __launch_bounds__(kThreadsF16, 2) void GemmF16K32Kernel(GemmF16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ring: 4 x 32 KiB = 2 pair slots
  constexpr int kPairBytes = 2 * kHalfSlabBytes;
#ifdef PK_F16_STAMPS
  const long long st_kernel = __builtin_amdgcn_s_memtime();
#endif

  // XCD-aware tile walk: workgroup ids are made contiguous per XCD and walk walk_m x walk_n super-tiles (4 x 4: the
  // 32 workgroups resident on an XCD then cover 8 x 4 tiles and share 8 X and 4 W panels in that XCD's L2)
  const int nblk = gridDim.x;                       // multiple of 8
  const int b = blockIdx.x;
  const int wg = (b % 8) * (nblk / 8) + b / 8;
  const int super_m = (a.tiles_m + a.walk_m - 1) / a.walk_m;
  const int per = a.walk_m * a.walk_n;
  const int s = wg / per, w = wg % per;
  const int tm = (s % super_m) * a.walk_m + (w % a.walk_m);
  const int tn = (s / super_m) * a.walk_n + (w / a.walk_m);
  if (tm >= a.tiles_m || tn >= a.tiles_n) return;
  const int m0 = tm * kT, n0 = tn * kT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves
  const int l15 = lane & 15, g = lane >> 4;
  const Exps ex = LoadExps(a, LAST);

  // ---- DMA role of this lane: as in the 32 x 32 form (pieces 2 wave, 2 wave + 1 of X and of W of
  // every half-slab; a piece is 16 LDS rows, lane -> row lane >> 2, stored position lane & 3); the W rows
  // of a piece are columns 4 apart
  const char *sbase[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row0 = (wave * 2 + p) * 16;
    sbase[p][0] = reinterpret_cast<const char *>(a.X + (int64_t)(m0 + row0) * a.ldx);
    sbase[p][1] = reinterpret_cast<const char *>(a.W + (int64_t)(n0 + WRowToCol16(row0)) * a.ldw);
  }
  const int lr = lane >> 2;
  const int q = (lane & 3) ^ Gray2((lr >> 2) & 3);  // logical position landing at this lane's slot
  const uint32_t voff[2] = {(uint32_t)(((int64_t)lr * a.ldx + q * 8) * sizeof(_Float16)),
                            (uint32_t)(((int64_t)4 * lr * a.ldw + q * 8) * sizeof(_Float16))};
  // compact rows (spliced first layer of the batch scorer): row r of this launch is row r + row_shift4[r / 4] of X
  uint32_t xoff[2] = {voff[0], voff[0]};
  if (a.row_shift4) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      xoff[p] += (uint32_t)((int64_t)a.row_shift4[(m0 + (wave * 2 + p) * 16 + lr) >> 2] * a.ldx * sizeof(_Float16));
  }
  // piece 0..7 of pair P: half-slab 2 P + (piece >> 2), operand (piece >> 1) & 1, p = piece & 1
  auto issue_piece = [&](int P, int piece) {
    const int hs = 2 * P + (piece >> 2), op = (piece >> 1) & 1, p = piece & 1;
    unsigned char *dst = smem + (hs & (kRingF16 - 1)) * kHalfSlabBytes + op * kOperandBytes + (wave * 2 + p) * 1024;
    DmaScalarBase(reinterpret_cast<const float *>(dst), sbase[p][op] + hs * 64, op == 0 ? xoff[p] : voff[1]);
  };

  f32x4v acc[8][4];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

  // fragment byte offsets inside a pair slot; row tile x / column tile y add 1024 x / 1024 y
  const int half = (g >> 1) * kHalfSlabBytes;
  const int aoff_h = half + SwzOff16(wm * 128 + l15, 2 * (g & 1));
  const int aoff_l = half + SwzOff16(wm * 128 + l15, 2 * (g & 1) + 1);
  const int boff_h = half + kOperandBytes + SwzOff16(wn * 64 + l15, 2 * (g & 1));
  const int boff_l = half + kOperandBytes + SwzOff16(wn * 64 + l15, 2 * (g & 1) + 1);

  const int npairs = a.K / 32;
#pragma unroll
  for (int piece = 0; piece < 8; ++piece) issue_piece(0, piece);
  if (npairs > 1) {
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) issue_piece(1, piece);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  f16x8 ah[2], al[2];          // A fragments: current row tile / next
  f16x8 bh[2][4], bl[2][4];    // B fragments of the current / next step: [parity][y]
  auto read_a = [&](const unsigned char *base, int x, f16x8 &h, f16x8 &l) {
    h = *reinterpret_cast<const f16x8 *>(base + aoff_h + 1024 * x);
    if (TERMS == 3) l = *reinterpret_cast<const f16x8 *>(base + aoff_l + 1024 * x);
  };
  auto read_b = [&](const unsigned char *base, int par, int y) {
    bh[par][y] = *reinterpret_cast<const f16x8 *>(base + boff_h + 1024 * y);
    if (TERMS == 3) bl[par][y] = *reinterpret_cast<const f16x8 *>(base + boff_l + 1024 * y);
  };
#pragma unroll
  for (int y = 0; y < 4; ++y) read_b(smem, 0, y);
  read_a(smem, 0, ah[0], al[0]);

#ifdef PK_F16_STAMPS
  long long st_acc[4] = {0, 0, 0, 0};
  long long st_last = __builtin_amdgcn_s_memtime();
  const long long st_begin = st_last;
#endif
  // one term of the four column tiles of row tile x: consecutive MFMAs are independent
  auto mfma_row = [&](int x, int cur, int par, int which) {
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (TERMS == 3 && which == 0) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bl[par][y], acc[x][y], 0, 0, 0);
      if (TERMS == 3 && which == 1) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur], bh[par][y], acc[x][y], 0, 0, 0);
      if (which == 2) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bh[par][y], acc[x][y], 0, 0, 0);
    }
  };
  // ---- main loop over k32 steps P.  A step is eight groups (row tiles x) of twelve MFMAs; the A
  // fragments of the next group are fetched under the current group's MFMAs.  Before the LAST group of
  // step P: every fragment of step P is in registers (lgkmcnt(0)), pair P + 1 has landed (vmcnt(0): it
  // is the only DMA in flight), barrier.  Behind it, under the MFMAs of that group and of the first
  // groups of step P + 1: the B fragments and the first A fragments of step P + 1, and the eight DMA
  // pieces of pair P + 2 into the slot step P has just released.
  auto step = [&](int P, const int par) {             // par = P & 1, compile-time in the body
    const unsigned char *base = smem + par * kPairBytes;
    const unsigned char *next = smem + (par ^ 1) * kPairBytes;
    const bool dma_late = P >= 1 && P + 1 < npairs;   // the pieces of pair P + 1 that did not go out in step P - 1
    const bool dma_head = P + 2 < npairs;             // the first pieces of pair P + 2
    // Where the eight pieces go (A/B on one box, profiles/r03_f16_dma_schedule.txt): two behind the barrier, then
    // ONE per group in groups 0-5 -- 500 / 475 TFLOP/s (two boxes) against 480 / 463 with two per group in groups
    // 0-2: all eight waves of the CU issue in step, so pieces close together queue up in the texture path.
    // One behind the barrier and one in each of groups 0-6 is no better, four behind the barrier is worse (462).
    constexpr int kHead = 2;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int cur = x & 1;
      if (x == 7) {
        PK_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): all of step P is in registers
        PK_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PK_STAMP(2);
        __builtin_amdgcn_s_barrier();
        PK_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        if (TERMS == 3) {
          mfma_row(x, cur, par, 0);
          __builtin_amdgcn_sched_barrier(0);
          read_b(next, par ^ 1, 0);                   // stale on the last step, unused
          read_b(next, par ^ 1, 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_row(x, cur, par, 1);
          __builtin_amdgcn_sched_barrier(0);
          read_b(next, par ^ 1, 2);
          read_b(next, par ^ 1, 3);
          read_a(next, 0, ah[cur ^ 1], al[cur ^ 1]);
          if (dma_head) issue_piece(P + 2, 0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_row(x, cur, par, 2);
          __builtin_amdgcn_sched_barrier(0);
          if (kHead == 2 && dma_head) issue_piece(P + 2, 1);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          mfma_row(x, cur, par, 2);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int y = 0; y < 4; ++y) read_b(next, par ^ 1, y);
          read_a(next, 0, ah[cur ^ 1], al[cur ^ 1]);
          if (dma_head) {
#pragma unroll
            for (int i = 0; i < kHead; ++i) issue_piece(P + 2, i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        read_a(base, x + 1, ah[cur ^ 1], al[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(x, cur, par, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (x < 6 && dma_late) issue_piece(P + 1, 2 + x);                         // pieces 2..7
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(x, cur, par, 1);
        mfma_row(x, cur, par, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int P = 0; P < npairs; P += 2) {
    step(P, 0);
    if (P + 1 < npairs) step(P + 1, 1);
  }

#ifdef PK_F16_STAMPS
  const long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: acc[x][y][r] = D[M0 + 16 x + 4 g + r][N0 + 4 l15 + y]; bias (nnet.cc:32-35), ReLU
  // (nnet.cc:56-58).  Staged through this wave's own 2 x 8 KiB of the idle ring, 32 rows (two row tiles)
  // at a time, as the rows will lie in memory -- 256 bytes per row: 64 fp32 logits, or 64 x (hi, lo)
  // halves in chunks of 8 -- and written out 16 bytes per lane: one store instruction = four whole
  // 256-byte row pieces.  No barrier: a wave reads back only what it wrote itself.
  const int M0 = m0 + wm * 128, N0 = n0 + wn * 64 + 4 * l15;
  // Operand exponents: the accumulators hold sum (X 2^e_in)(W 2^e_w); what is written is
  //   v 2^e_out,  v = acc 2^-(e_in + e_w) + bias        (nnet.cc:32-35)
  // as ONE fma per value, acc * 2^(e_out - e_in - e_w) + bias * 2^e_out: every factor is a power of two, so this is
  // the rounding of v itself, scaled exactly (ReLU commutes with a positive factor).  Costs what the plain bias
  // add cost (v_pk_fma_f32 for v_pk_add_f32).  e_out is 0 for fp32 output.
  const float acc_scale = Pow2(ex.e_out - ex.e_in_w);
  const f32x4v bias = *reinterpret_cast<const f32x4v *>(a.bias + N0) * Pow2(ex.e_out);
  const f32x4v acc_scale4 = f32x4v{acc_scale, acc_scale, acc_scale, acc_scale};
  unsigned char *stage = smem + wave * (2 * 8192);
  // byte offset of this lane's four columns in a staged row: fp32 quad, or the hi halves of chunk l15 / 2
  const int quad_byte = LAST ? l15 * 16 : (l15 >> 1) * 32 + (l15 & 1) * 8;
  unsigned char *out_rows = LAST ? reinterpret_cast<unsigned char *>(a.out_f32 + (int64_t)M0 * a.ldo + n0 + wn * 64)
                                 : reinterpret_cast<unsigned char *>(a.out + (int64_t)M0 * a.ldo + 2 * (n0 + wn * 64));
  const int64_t row_bytes = a.ldo * (LAST ? (int64_t)sizeof(float) : (int64_t)sizeof(_Float16));
  // bias and ReLU on whole accumulator vectors: the add pairs up (v_pk_add_f32), the ReLU is one v_max_f32.  (max
  // instead of the reference's x < 0 ? 0 : x differs for NaN and -0.0 only, which this mode's saturating split
  // does not carry anyway.)
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      acc[x][y] = __builtin_elementwise_fma(acc[x][y], acc_scale4, f32x4v{bias[y], bias[y], bias[y], bias[y]});
      if (RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[x][y][r] = fmaxf(acc[x][y][r], 0.0f);
      }
    }
  s16x4 hmax[2] = {s16x4{0, 0, 0, 0}, s16x4{0, 0, 0, 0}};   // range word of the operand written here: max |hi| (RangeAcc), two chains
#pragma unroll
  for (int xp = 0; xp < 4; ++xp) {
    unsigned char *buf = stage + (xp & 1) * 8192;
#pragma unroll
    for (int xx = 0; xx < 2; ++xx) {
      const int x = 2 * xp + xx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * xx + 4 * g + r;
        float v[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) v[y] = acc[x][y][r];
        if (LAST) {
          *reinterpret_cast<f32x4v *>(buf + row * 256 + quad_byte) = f32x4v{v[0], v[1], v[2], v[3]};
        } else {
          const SplitOut s0 = Split(v[0]), s1 = Split(v[1]), s2 = Split(v[2]), s3 = Split(v[3]);
          const f16x4 hi = f16x4{s0.hi, s1.hi, s2.hi, s3.hi};
          hmax[r & 1] = RangeAcc<RELU>(hmax[r & 1], hi);
          *reinterpret_cast<f16x4 *>(buf + row * 256 + quad_byte) = hi;
          *reinterpret_cast<f16x4 *>(buf + row * 256 + quad_byte + 16) = f16x4{s0.lo, s1.lo, s2.lo, s3.lo};
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // this wave's LDS writes are in
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {                       // 8 KiB = 8 x (64 lanes x 16 bytes); lane -> row j*4 + lane/16
      const int row = j * 4 + (lane >> 4);
      const f32x4v v = *reinterpret_cast<const f32x4v *>(buf + row * 256 + (lane & 15) * 16);
      *reinterpret_cast<f32x4v *>(out_rows + (int64_t)(32 * xp + row) * row_bytes + (lane & 15) * 16) = v;
    }
  }
  if (!LAST && a.range) {
    const s16x4 h4 = __builtin_elementwise_max(hmax[0], hmax[1]);
    const s16x2 h2 = __builtin_elementwise_max(s16x2{h4[0], h4[1]}, s16x2{h4[2], h4[3]});
    PublishRange(a.range, blockIdx.x * 8 + wave, RangeValue(h2[0] > h2[1] ? h2[0] : h2[1]));
  }
#ifdef PK_F16_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
    long long *o = pk_f16_stamps + (blockIdx.x * 8 + wave) * 8;
    o[0] = st_loop_end - st_begin;                       // k loop, cycles
    o[1] = st_acc[0];                                    // MFMA / issue segments (barrier release -> next lgkm wait)
    o[2] = st_acc[1];                                    // lgkmcnt(0) wait
    o[3] = st_acc[2];                                    // vmcnt wait
    o[4] = st_acc[3];                                    // barrier wait
    o[5] = st_begin - st_kernel;                         // prologue
    o[6] = __builtin_amdgcn_s_memtime() - st_loop_end;   // epilogue (stores issued)
  }
#endif
}

// fp32 -> interleaved (hi, lo) fp16 rows.  in: element (r, c) at in[r * stride_r + c *
// stride_c]; out row r starts at out + r * ld_out (halves); logical column c lives at
// (c / 8) * 16 + c % 8 (hi) and 8 halves further (lo); columns cols..cols_pad-1 are zero.
__global__ void SplitKernel(const float *__restrict__ in, int64_t stride_r, int64_t stride_c,
                            int rows, int cols, int cols_pad, _Float16 *__restrict__ out,
                            int64_t ld_out, const int32_t *__restrict__ e_x, uint32_t *range) {
  const int64_t total = (int64_t)rows * cols_pad;
  const float sc = e_x ? Pow2(*e_x) : 1.0f;
  float hmax = 0.0f;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = idx / cols_pad, c = idx % cols_pad;
    const float v = c < cols ? in[(int64_t)r * stride_r + (int64_t)c * stride_c] * sc : 0.0f;
    const SplitOut s = Split(v);
    hmax = fmaxf(hmax, fabsf(static_cast<float>(s.hi)));
    _Float16 *o = out + (int64_t)r * ld_out + (c >> 3) * 16 + (c & 7);
    o[0] = s.hi;
    o[8] = s.lo;
  }
  if (range) PublishRange(range, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), hmax);
}

}  // namespace

namespace {
// a zero exponent word for callers that scale nothing (probes, tests): one per device, made on first use
const int32_t *ZeroWord() {
  static std::mutex mu;
  static int32_t *word[64] = {};
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  if (dev < 0 || dev >= 64) return nullptr;
  if (!word[dev] && hipMalloc(&word[dev], sizeof(int32_t)) == hipSuccess) hipMemset(word[dev], 0, sizeof(int32_t));
  return word[dev];
}
}  // namespace

void LaunchGemmF16(const GemmF16Args &a_in, hipStream_t stream) {
  GemmF16Args a = a_in;
  if (!a.e_w) a.e_w = ZeroWord();
  if (!a.e_in) a.e_in = ZeroWord();
  if (!a.e_out) a.e_out = ZeroWord();
  // PK_MI355_F16_WALK=MxN: the super-tile shape of the 16x16x32 kernel's tile walk, a measurement switch
  // (profiles/r04_f16_walk_ab.txt).  The grid is rounded up to a multiple of 8 (one share per XCD); ids past the
  // last super-tile fall outside the tile range and return at once.
  static const int walk[2] = {[] { const char *e = getenv("PK_MI355_F16_WALK"); int m = 4, n = 4; if (e) sscanf(e, "%dx%d", &m, &n); return m > 0 && m <= 64 ? m : 4; }(),
                              [] { const char *e = getenv("PK_MI355_F16_WALK"); int m = 4, n = 4; if (e) sscanf(e, "%dx%d", &m, &n); return n > 0 && n <= 64 ? n : 4; }()};
  a.walk_m = walk[0];
  a.walk_n = walk[1];
  const int super_m = (a.tiles_m + a.walk_m - 1) / a.walk_m, super_n = (a.tiles_n + a.walk_n - 1) / a.walk_n;
  const int nblk = (super_m * super_n * a.walk_m * a.walk_n + 7) / 8 * 8;
  dim3 grid(nblk), block(kThreadsF16);
  const size_t lds = kRingF16 * kHalfSlabBytes;
  // the 128 KiB dynamic-LDS opt-in is a per-DEVICE function attribute: set it once for every device a
  // thread launches on (models and batches may be driven from several host threads)
  {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
#define PK_SET_LDS(R, L, T) do { \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmF16Kernel<R, L, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmF16K32Kernel<R, L, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); } while (0)
      PK_SET_LDS(true, false, 3); PK_SET_LDS(false, false, 3); PK_SET_LDS(true, true, 3); PK_SET_LDS(false, true, 3);
      PK_SET_LDS(true, false, 1); PK_SET_LDS(false, false, 1); PK_SET_LDS(true, true, 1); PK_SET_LDS(false, true, 1);
#undef PK_SET_LDS
      attr_set[dev] = true;
    }
  }
  // MFMA shape.  f16x3 (three MFMAs per product: bound by the matrix pipes and by the clock the chip holds) runs
  // on 16x16x32, k32 steps: 486-506 against 447 TFLOP/s algorithmic on the wide model, profiles/r03_f16_shape_ab.txt.
  // Plain fp16 (one MFMA per product: bound by the L2 -> LDS operand stream) keeps the round-2 32x32x16 form, whose
  // k16 steps fetch three steps ahead instead of one: 675 against 597, profiles/r03_f16_plain_shape_ab.txt.
  // PK_MI355_F16_SHAPE=16 / 32 forces one form for A/B measurements.  Same operands, layouts and tile; results
  // differ in the last bits only (the k order of the fp32 accumulation), so the choice is fixed per process.
  static const int forced = [] { const char *e = getenv("PK_MI355_F16_SHAPE"); return e ? atoi(e) : 0; }();
  const bool k32 = forced == 16 || (forced != 32 && a.terms == 3);
#define PK_LAUNCH(R, L, T) do { if (k32) hipLaunchKernelGGL((GemmF16K32Kernel<R, L, T>), grid, block, lds, stream, a); \
                                else hipLaunchKernelGGL((GemmF16Kernel<R, L, T>), grid, block, lds, stream, a); } while (0)
  if (a.terms == 1) {
    if (a.out_f32) { if (a.relu) PK_LAUNCH(true, true, 1); else PK_LAUNCH(false, true, 1); }
    else { if (a.relu) PK_LAUNCH(true, false, 1); else PK_LAUNCH(false, false, 1); }
  } else {
    if (a.out_f32) { if (a.relu) PK_LAUNCH(true, true, 3); else PK_LAUNCH(false, true, 3); }
    else { if (a.relu) PK_LAUNCH(true, false, 3); else PK_LAUNCH(false, false, 3); }
  }
#undef PK_LAUNCH
}

// NormalizeLayer (nnet.cc:62-75) between two f16x3 affine layers: fp32 rows in (what the GEMM wrote with
// its out_f32 epilogue), y = x * float(sqrt(D / sum x^2)), interleaved (hi, lo) rows out -- the next layer's
// operand.  One wave per row, a lane owns chunks of 8 columns (one 16-byte hi store + one lo store each);
// the row is read twice (the second time out of L2).  The sum is a tree, not the reference's sequential
// float sum: this mode is not bit-exact anyway (1e-4 contract).  Columns n .. npad - 1 are written as zeros.
namespace {
__global__ __launch_bounds__(256) void NormalizeSplitKernel(const float *__restrict__ in, int64_t ld_in, int rows, int n,
                                                            int npad, _Float16 *__restrict__ out, int64_t ld_out,
                                                            const int32_t *__restrict__ e_x, uint32_t *range) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;                      // wave-uniform
  const float *x = in + (int64_t)row * ld_in;
  float ssq = 0.0f;
  for (int c = lane * 8; c < npad; c += 64 * 8) {
    const f32x4v a = *reinterpret_cast<const f32x4v *>(x + c), b = *reinterpret_cast<const f32x4v *>(x + c + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < n) ssq += a[e] * a[e];
      if (c + 4 + e < n) ssq += b[e] * b[e];
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ssq += __shfl_xor(ssq, m);
  // nnet.cc:70-72.  An all-zero row makes the reference produce NaN (0 * inf, no epsilon: SURVEY a17); this mode's
  // operands cannot carry NaN (the split saturates, include/pk_mi355.h), so such a row stays all-zero here
  // the operand's exponent rides on the row's scale: (x * scale) * 2^e == x * (scale * 2^e) exactly
  const float scale = (ssq > 0.0f ? static_cast<float>(sqrt(static_cast<double>(n) / static_cast<double>(ssq))) : 0.0f) *
                      (e_x ? Pow2(*e_x) : 1.0f);
  float hmax = 0.0f;
  _Float16 *o = out + (int64_t)row * ld_out;
  for (int c = lane * 8; c < npad; c += 64 * 8) {
    const f32x4v a = *reinterpret_cast<const f32x4v *>(x + c), b = *reinterpret_cast<const f32x4v *>(x + c + 4);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = e < 4 ? a[e] : b[e - 4];
      const SplitOut sp = Split(c + e < n ? v * scale : 0.0f);
      hi[e] = sp.hi;
      lo[e] = sp.lo;
      hmax = fmaxf(hmax, fabsf(static_cast<float>(sp.hi)));
    }
    *reinterpret_cast<f16x8 *>(o + 2 * c) = hi;
    *reinterpret_cast<f16x8 *>(o + 2 * c + 8) = lo;
  }
  if (range) PublishRange(range, row, hmax);
}
}  // namespace

void LaunchNormalizeSplitF16(const float *in, int64_t ld_in, int rows, int n, int npad, _Float16 *out, int64_t ld_out,
                             const int32_t *e_x, uint32_t *range, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return;
  hipLaunchKernelGGL(NormalizeSplitKernel, dim3((rows + 3) / 4), dim3(256), 0, stream, in, ld_in, rows, n, npad, out, ld_out,
                     e_x, range);
}

void LaunchSplitF16(const float *in, int64_t stride_r, int64_t stride_c, int rows, int cols,
                    int cols_pad, _Float16 *out, int64_t ld_out, const int32_t *e_x, uint32_t *range,
                    hipStream_t stream) {
  if (rows <= 0 || cols_pad <= 0) return;
  int64_t n = (int64_t)rows * cols_pad;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(SplitKernel, dim3(blocks), dim3(256), 0, stream, in, stride_r, stride_c, rows,
                     cols, cols_pad, out, ld_out, e_x, range);
}

}  // namespace pkmi
