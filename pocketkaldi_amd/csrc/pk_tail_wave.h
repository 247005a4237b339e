// pk_tail_wave.h -- the log-likelihood tail of the fp32 mode, one WAVE per row (internal).
//
// SoftmaxLayer (nnet.cc:38-47) + the floor / log / prior step of AcousticModel::Compute (am.cc:106-112) + the
// acoustic scale (decodable.cc:15), overflow-safe:   out = scale * (max(x - lse, log 1e-20) - log prior),
// lse = m + log sum exp(x - m).  Lane l of the wave owns the 16-byte chunks q = l + 64 c (c < C) of the row, keeps
// them in registers and reduces on the vector ALU (DPP) -- no workgroup barrier, no LDS exchange for rows of up to
// 4 096 columns; wider rows are shared by two waves.  exp is the hardware's v_exp_f32 on fma(x, log2 e, -m log2 e)
// (1 ulp; two instructions, which keeps a row's code to a few hundred of them -- a tail phase inlined with the libm
// expf, per-chunk branches and spilled masks ran at 20 cycles per instruction: profiles/r04_f16_fused_tail.txt).
// Measured against the oracle: 1.2e-6 on log-likelihoods (TailKernel: 0.95e-6); the contract is 1e-4.
//
// Two users, ONE arithmetic, so that a model's results do not depend on which of them ran:
//   * the last affine layer's big-tile launches (gemm.hip, TAIL variant): the workgroup that completes a 128-row
//     tile of logits finishes those rows itself (SC1 = true: the rows were written by other CUs, possibly other XCDs);
//   * TailWaveKernel (tail.hip): the stand-alone launch for everything the fused form does not take.
#ifndef PK_TAIL_WAVE_H_
#define PK_TAIL_WAVE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pk_wave.h"

namespace pkmi {

typedef float twf4 __attribute__((ext_vector_type(4)));
typedef unsigned int twu4 __attribute__((ext_vector_type(4)));

constexpr int kTailWaveMaxChunks = 32;          // 32 x 64 lanes x 4 = 8 192 columns at most

// Rows first, first + step, ... of `in` (row r at in + r * ld_in; with SC1 the whole block must lie within 2 GB of
// `in`), `iters` of them per wave -- the SAME count for every wave of the workgroup (rows >= rows_end are skipped,
// their barriers are not).  prior: the table -log prior * scale as 16-byte chunks, zero-padded to a multiple of four
// (TailWavePriorEntry) -- a global pointer, or an LDS one (PriorPtr = const twf4 __attribute__((address_space(3))) *:
// ds_read_b128, no flat access).
// PAIR = false: a wave owns a whole row, chunks q = lane + 64 c, c < C (rows of at most 256 C columns).
// PAIR = true:  rows wider than 4 096 columns are shared by the two waves of a pair (wave parity h owns chunks
//               lane + 64 (C h + c)); the halves' maxima and sums meet in LDS (`xch`, 4 floats per pair and use)
//               across two workgroup barriers per row, combined in a fixed order.
// The next row's chunks are requested into a second register set before the current row is touched and moved over
// when it is done: a wave has a full row of compute between a request and its use.
// Inside a row nothing branches: chunks past the row's end are read from a clamped (valid) position and selected
// to -inf, their stores fall outside the row's buffer descriptor and are dropped by the hardware -- per-chunk
// `if`s cost the first version 90 exec-mask branches and 160 SGPR spills per row.
// EXACT = true: the row has exactly C (2 C for a pair) chunks, so only the very last one can reach past n and needs
// the selects; EXACT = false: C is rounded up and every chunk is tested (the selects of the first form cost a fifth of
// a row's instructions and 48 SGPR spills for the hoisted masks).
// The prior table -log prior * scale, two sources with the same values: copied into LDS once per workgroup
// (ds_read_b128; pays when a workgroup walks many rows), or formed per use from the log priors in global memory
// (small launches: no copy, no barrier).
struct TailWaveLdsPrior {
  const twf4 __attribute__((address_space(3))) *tab;
  __device__ __forceinline__ twf4 operator[](int q) const { return tab[q]; }
};
struct TailWaveGlobalPrior {
  const float *log_prior;      // zero-padded to a multiple of four
  float scale;
  __device__ __forceinline__ twf4 operator[](int q) const {
    const twf4 lp = *reinterpret_cast<const twf4 *>(log_prior + 4 * q);
    return twf4{-lp[0] * scale, -lp[1] * scale, -lp[2] * scale, -lp[3] * scale};
  }
};

template <int C, bool SC1, bool PAIR, bool EXACT, typename PriorPtr>
__device__ __forceinline__ void TailWaveRows(const float *__restrict__ in, int64_t ld_in, int first, int step, int iters,
                                             int rows_end, int n, PriorPtr prior, float scale,
                                             float *__restrict__ out, int64_t ld_out, int lane, int half, float *xch) {
  const float kLogFloor = -46.051701859880914f;   // logf(1e-20f), am.cc:109-110
  const float kLog2e = 1.4426950408889634f;
  // what is the same in every lane of the wave is said to be (the callers derive these from the wave's number, which
  // hipcc cannot know to be uniform: the row's buffer descriptor would otherwise be built per lane, in a waterfall loop)
  first = __builtin_amdgcn_readfirstlane(first);
  step = __builtin_amdgcn_readfirstlane(step);
  iters = __builtin_amdgcn_readfirstlane(iters);
  half = __builtin_amdgcn_readfirstlane(half);
  const int n4 = (n + 3) >> 2;
  const bool vec_out = ((ld_out & 3) == 0) && ((n & 3) == 0);          // wave-uniform
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 0x7fffffff, 0x00020000);
  const uint32_t pitch = (uint32_t)(ld_in * sizeof(float));
  const int q0 = lane + (PAIR ? 64 * C * half : 0);
  int qc[C];                                        // this lane's chunk positions, clamped into the row
#pragma unroll
  for (int c = 0; c < C; ++c) qc[c] = (q0 + 64 * c < n4) ? q0 + 64 * c : n4 - 1;
  auto load = [&](int row, int c) -> twf4 {
#ifdef PK_EXP_TAIL_NT
    if (SC1) return __builtin_bit_cast(twf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)row * pitch + qc[c] * 16, 0, 18 /* sc1 nt */));
#else
    if (SC1) return __builtin_bit_cast(twf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)row * pitch + qc[c] * 16, 0, 16 /* sc1 */));
#endif
    return *reinterpret_cast<const twf4 *>(in + (int64_t)row * ld_in + qc[c] * 4);
  };
  twf4 v[C], nv[C];
  if (iters > 0 && first < rows_end) {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = load(first, c);
  }
  for (int it = 0; it < iters; ++it) {
    const int row = first + it * step;
    const bool live = row < rows_end;                 // wave-uniform
    if (live && it + 1 < iters && row + step < rows_end) {
#pragma unroll
      for (int c = 0; c < C; ++c) nv[c] = load(row + step, c);
    }
    // columns past n (the row's padding, and the clamped chunks) must not take part: -inf for the maximum and the sum
    float m = -INFINITY;
    if (live) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (!EXACT || c == C - 1) {
          const int col = (q0 + 64 * c) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[c][e] = (col + e < n) ? v[c][e] : -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, v[c][e]);
      }
      m = TwWaveMax(m);
    }
    if (PAIR) {
      if (lane == 0) xch[half] = m;
      __syncthreads();
      m = fmaxf(xch[0], xch[1]);
    }
    float s = 0.0f;
    if (live) {
      const float nm = -m * kLog2e;                    // exp(x - m) = exp2(x log2e - m log2e): one fma per element
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f(__builtin_fmaf(v[c][e], kLog2e, nm));
      s = TwWaveSum(s);
    }
    if (PAIR) {
      if (lane == 0) xch[2 + half] = s;
      __syncthreads();
      s = xch[2] + xch[3];
    }
    if (live) {
      const float lse = m + logf(s);
      // A row with a NaN logit (or +inf, or nothing but -inf) has lse = NaN and must come out NaN, as the reference's
      // does (am.cc:109 `x < 1e-20` is false for NaN, log follows): v_max_f32(NaN, floor) would return the floor and
      // turn the row into plausible numbers.  One select per ROW instead of a compare-select per element: the floor
      // operand itself becomes NaN, and max(NaN, NaN) = NaN.  (lse is finite for every other row: s is in [1, 8192].)
      const float row_floor = (lse == lse) ? kLogFloor : lse;
      // the row as a buffer of n floats: a store past its end is dropped by the range check
      const __amdgpu_buffer_rsrc_t yrow = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)row * ld_out, 0, n * 4, 0x00020000);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const twf4 lp = prior[qc[c]];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[c][e] - lse;                                               // log softmax, stable form
          t = fmaxf(t, row_floor);                                               // floor of am.cc:109 in the log domain
          o[e] = __builtin_fmaf(t, scale, lp[e]);                                // (t - log prior) * scale (am.cc:111, decodable.cc:15);
                                                                                 // the table holds -log prior * scale
        }
        const int off = (q0 + 64 * c) * 16;                                      // (the unclamped position)
        if (vec_out) {
          const twu4 bits = twu4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
          __builtin_amdgcn_raw_buffer_store_b128(bits, yrow, off, 0, 2 /* nt */);
        } else {
          // (scalars, not elements of a vector: hipcc 7.2 stored element 0 four times from `bit_cast(o[e])` of an ext-vector)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[0]), yrow, off, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[1]), yrow, off + 4, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[2]), yrow, off + 8, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[3]), yrow, off + 12, 0, 2);
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = nv[c];
    }
  }
}

// One entry of the prior table both users build in LDS.
__device__ __forceinline__ float TailWavePriorEntry(float log_prior, float scale) { return -log_prior * scale; }

// Shape of a row: chunks per wave and whether two waves share it.  One place, so that the fused and the stand-alone
// form pick the same instantiation for the same n.
__host__ __device__ inline int TailWaveChunks(int n) { return (((n + 3) >> 2) + 63) >> 6; }        // 256-column chunks of a row
__host__ __device__ inline bool TailWavePair(int n) { return TailWaveChunks(n) > 16; }
// the chunk counts that have an EXACT instantiation (and with it a fused form): 1 024 k columns, 6 144, 8 192
__host__ __device__ inline bool TailWaveExact(int n) {
  const int c = TailWaveChunks(n);
  return c == 4 || c == 8 || c == 12 || c == 16 || c == 24 || c == 32;
}

}  // namespace pkmi

#endif  // PK_TAIL_WAVE_H_
