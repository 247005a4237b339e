// pk_wave.h -- wave-wide (64-lane) reductions on the vector ALU (internal): DPP row operations, the result taken
// from lane 63 and handed to every lane as a scalar.  No LDS permutes (ds_bpermute: an LDS instruction plus three
// vector instructions of index arithmetic per step, and six dependent round trips per reduction).
#ifndef PK_WAVE_H_
#define PK_WAVE_H_

#include <hip/hip_runtime.h>

namespace pkmi {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float TwDpp(float x, float identity) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
template <typename Op>
__device__ __forceinline__ float TwWaveReduce(float v, Op op, float identity) {
  v = op(v, TwDpp<0xB1, 0xf>(v, identity));      // quad_perm [1, 0, 3, 2]
  v = op(v, TwDpp<0x4E, 0xf>(v, identity));      // quad_perm [2, 3, 0, 1]
  v = op(v, TwDpp<0x141, 0xf>(v, identity));     // row_half_mirror
  v = op(v, TwDpp<0x140, 0xf>(v, identity));     // row_mirror
  v = op(v, TwDpp<0x142, 0xa>(v, identity));     // row_bcast:15 into rows 1 and 3
  v = op(v, TwDpp<0x143, 0xc>(v, identity));     // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float TwWaveMax(float v) {
  return TwWaveReduce(v, [](float a, float b) { return fmaxf(a, b); }, -INFINITY);
}
__device__ __forceinline__ float TwWaveSum(float v) {
  return TwWaveReduce(v, [](float a, float b) { return a + b; }, 0.0f);
}

}  // namespace pkmi

#endif  // PK_WAVE_H_
