// pk_dma.h -- LDS-DMA (global_load_lds_dwordx4) for gfx950, shared by the GEMM kernels.
#ifndef PK_DMA_H_
#define PK_DMA_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pkmi {

// LDS-DMA, 16 bytes per lane: LDS[m0 + lane * 16 ...] <- global.  Written as inline assembly so
// that the scalar-base form can be used: a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane
// byte offset.  With 64-bit per-lane addresses the same transfers cost the fp32 GEMM 6 % of its
// time (133 -> 140.7 TFLOP/s when they went).  A kernel that uses these issues ALL its LDS-DMA
// through them, so M0 is never shared with compiler-generated LDS-DMA.
__device__ __forceinline__ uint32_t LdsAddr(const float *p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));     // low half of the flat address
}
__device__ __forceinline__ void DmaScalarBase(const float *lds_dst, const char *uniform_base, uint32_t lane_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               ::"s"(LdsAddr(lds_dst)), "v"(lane_off), "s"(uniform_base)
               : "memory");
}
__device__ __forceinline__ void DmaVectorAddr(const float *lds_dst, const float *lane_ptr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               ::"s"(LdsAddr(lds_dst)), "v"(lane_ptr)
               : "memory");
}

}  // namespace pkmi

#endif  // PK_DMA_H_
