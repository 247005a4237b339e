// net.hip -- a run of affine (+ ReLU) layers on ONE utterance's worth of rows as a single
// persistent launch (nnet.cc:149-163 driving nnet.cc:22-36 / :49-60).
//
// Why: with one utterance on the chip every layer is a launch of a few microseconds whose fixed
// cost (dispatch, first operand round trip, drain, the boundary) nothing covers -- 5 launches cost
// the S model ~30 us on top of ~90 us of matrix work.  A layer boundary INSIDE a launch is only
// cheaper than a kernel boundary if producer and consumer share an L2: per-XCD L2s are not coherent
// with each other, and publishing a tile across XCDs (write-through stores or a release fence, a
// device-scope flag, an acquire on the reader) measures 7-24 us (tools/ubench/sync_probe.hip) against
// 1.9 us for the same hand-off between CUs of one XCD.  So the work is dealt by 64-row blocks to the
// XCDs: a workgroup asks the hardware which XCD it runs on (HW_REG_XCC_ID -- a fact, not an
// assumption about dispatch order), takes a ticket there, and computes only tiles of that XCD's row
// blocks, for every layer.  A layer's output tile is stored with plain stores (it stays in that
// XCD's L2), drained (s_waitcnt vmcnt(0)), and counted; the next layer's tile of the same row block
// waits for the count and fetches the panel with sc1 LDS-DMA (served by the L2, never by the CU's own
// vector L1).  The 0.5 MB of activations an XCD works on never leave its 4 MB L2.
//
// Arithmetic: exactly GemmKernel<S = 1, KG = 2> (gemm.hip): 64 x 64 tiles, one 32 x 32 MFMA
// accumulator per wave, k ascending inside a 512-chunk (gemm.h:50), one group of four waves per
// chunk, chunk sums added in chunk order (gemm.cc:95-123), bias, ReLU -- bit-identical to the
// reference's SGEMM + nnet.cc:32-35,56-58 like the multi-launch path.
//
// Progress: a tile of layer l + 1 waits only for tiles of layer l, every workgroup walks its
// tiles in layer order, and the grid (256 workgroups) is resident at once, so the lowest unfinished
// layer can always advance.  Every spin is bounded: a wait that runs out (an XCD that received
// fewer than 32 workgroups, which round-robin dispatch does not produce) raises a host-visible
// status word instead of hanging, and the host recomputes through the multi-launch path.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pk_dma.h"
#include "pk_kernels.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNT = 64;                         // tile edge
constexpr int kNSlab = kBK * kNT;               // floats per operand per slab (4 KiB)
constexpr int kNRing = 3, kNAhead = 2;          // as GemmKernel: three slabs, two in flight
constexpr int kNDma = 2;                        // DMA pieces per wave and slab: 4 k-rows of P, 4 of Q
constexpr int kGroupFloats = kNRing * 2 * kNSlab;
constexpr int kNetThreads = 512;                // two k-groups of four waves
constexpr int kWgPerXcd = 32;
constexpr int kSpinLimit = 1 << 16;             // x (~64 cycles of s_sleep + one L2 round trip) ~ 50 ms
constexpr int kStepsPerGroup = kChunkK / kBK;   // 32 slabs = one 512-chunk

struct Tile {                                   // one 64 x 64 output tile; everything wave-uniform
  const float *P; int64_t ldp;
  const float *Q; int64_t ldq;
  int K, splice_dim;
  const float *bias;
  int bias_on_j, relu;
  float *out; int64_t ldo;
  int i0, j0;
};

// COH: which operand was written earlier in this launch (0 none, 1 P, 2 Q) and is fetched with sc1.
template <bool SPLICE, int COH>
__device__ __forceinline__ void ComputeTile(const Tile &t, float *smem_all, int lane, int grp, int wave) {
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  float *smem = smem_all + grp * kGroupFloats;
  const int nkt_all = t.K / kBK;
  const int nsteps = nkt_all < kStepsPerGroup ? nkt_all : kStepsPerGroup;          // barriers per tile: both groups
  const int n_g = grp == 0 ? nsteps : (nkt_all > kStepsPerGroup ? nkt_all - kStepsPerGroup : 0);
  const int kbase = grp * kChunkK;
  const float *pg = t.P + t.i0 + (int64_t)kbase * t.ldp;
  const float *qg = SPLICE ? t.Q + t.j0 : t.Q + t.j0 + (int64_t)kbase * t.ldq;
  const uint32_t lane_off_p = (uint32_t)(((lane >> 4) * t.ldp + (lane & 15) * 4) * sizeof(float));
  const uint32_t lane_off_q = (uint32_t)(((lane >> 4) * t.ldq + (lane & 15) * 4) * sizeof(float));

  // slab kt of this group -> ring slot, as two 1 KiB pieces per wave (k-rows 4 wave .. 4 wave + 3)
  auto issue = [&](int kt, int slot, int piece) {
    const int row = wave * 4;
    if (piece == 0) {
      float *dst = smem + (slot * 2 + 0) * kNSlab + row * kNT;
      const char *base = reinterpret_cast<const char *>(pg + (int64_t)(kt * kBK + row) * t.ldp);
      if (COH == 1) DmaScalarBaseL2(dst, base, lane_off_p);
      else DmaScalarBase(dst, base, lane_off_p);
    } else if (!SPLICE) {
      float *dst = smem + (slot * 2 + 1) * kNSlab + row * kNT;
      const char *base = reinterpret_cast<const char *>(qg + (int64_t)(kt * kBK + row) * t.ldq);
      if (COH == 2) DmaScalarBaseL2(dst, base, lane_off_q);
      else DmaScalarBase(dst, base, lane_off_q);
    } else {
      // am.cc:65-88 as an address function: operand row k = feature k % D shifted by k / D frames
      const int k = kbase + kt * kBK + row + (lane >> 4);
      const int c = k / t.splice_dim, d = k - c * t.splice_dim;
      DmaVectorAddr(smem + (slot * 2 + 1) * kNSlab + row * kNT, qg + (int64_t)d * t.ldq + c + (lane & 15) * 4);
    }
  };
  auto read_frags = [&](int slot, int first, float (&pf)[kBK / 4], float (&qf)[kBK / 4]) {
    const float *ps = smem + (slot * 2 + 0) * kNSlab + wi * 32 + l31 + half * kNT;
    const float *qs = smem + (slot * 2 + 1) * kNSlab + wj * 32 + l31 + half * kNT;
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      pf[ks] = ps[2 * (first + ks) * kNT];
      qf[ks] = qs[2 * (first + ks) * kNT];
    }
  };

  f32x16 acc = {0};
  float fa_p[kBK / 4], fa_q[kBK / 4], fb_p[kBK / 4], fb_q[kBK / 4];
#pragma unroll
  for (int s0 = 0; s0 < kNAhead; ++s0)
    if (s0 < n_g) { issue(s0, s0, 0); issue(s0, s0, 1); }
  if (n_g >= kNAhead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kNAhead - 1) * kNDma) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0, fa_p, fa_q);

  // bias values in front of the k loop (their round trip is not added to the epilogue)
  const int I0 = t.i0 + wi * 32, J0 = t.j0 + wj * 32;
  float bj = 0.0f, bi[16];
  if (grp == 0) {
    if (t.bias_on_j) {
      bj = t.bias[J0 + l31];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bi[r] = t.bias[I0 + (r & 3) + 8 * (r >> 2) + 4 * half];
    }
  }

  int slot = 0;
  auto slab_step = [&](int kt, auto dma_tag) {
    constexpr bool dma = decltype(dma_tag)::value;
    const int slot1 = slot + 1 == kNRing ? 0 : slot + 1;
    const int slot2 = slot == 0 ? kNRing - 1 : slot - 1;       // slab kt + 2 takes the slot of slab kt - 1
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_p[ks], fa_q[ks], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) read_frags(slot, kBK / 4, fb_p, fb_q);
      if (dma && ks == 1) issue(kt + kNAhead, slot2, 0);
      if (dma && ks == 3) issue(kt + kNAhead, slot2, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0)
    if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kNAhead - 1) * kNDma) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(slot1, 0, fa_p, fa_q);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < kBK / 4; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb_p[ks], fb_q[ks], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    slot = slot1;
  };
  int kt = 0;
  for (; kt + kNAhead < n_g; ++kt) slab_step(kt, std::true_type());
  for (; kt < n_g; ++kt) slab_step(kt, std::false_type());
  for (; kt < nsteps; ++kt) __builtin_amdgcn_s_barrier();      // the other group's slabs: keep its barriers company

  // chunk sums in chunk order (gemm.cc:95-123): group 1 hands its accumulator over through the
  // rings (no slab is read any more: every wave is past the last barrier of the loop)
  if (nkt_all > kStepsPerGroup) {
    float *xch = smem_all;
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = acc[r];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if (grp == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] + xch[(wave * 16 + r) * 64 + lane];
    }
  }
  if (grp != 0) return;

  // register r of lane (l31, half) holds D[I0 + (r & 3) + 8 (r >> 2) + 4 half][J0 + l31]
  float *obase = t.out + (int64_t)(I0 + 4 * half) * t.ldo + J0 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r];
    v += t.bias_on_j ? bj : bi[r];                             // nnet.cc:32-35
    if (t.relu) v = v < 0.0f ? 0.0f : v;                       // nnet.cc:56-58
    obase[(int64_t)((r & 3) + 8 * (r >> 2)) * t.ldo] = v;
  }
}

__device__ unsigned long long g_stamps[256 * 16];

__global__ __launch_bounds__(kNetThreads, 1) void NetKernel(NetArgs a) {
  // ALL LDS in one array (gemm.hip: a second __shared__ object makes hipcc drain the DMA queue
  // in front of every LDS read); its first words carry the workgroup's placement at start-up
  __shared__ __attribute__((aligned(16))) float smem_all[2 * kGroupFloats];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave_all >> 2, wave = wave_all & 3;

  // the counters of the NEXT launch on this workspace (kernel boundaries order them)
  for (int i = blockIdx.x * kNetThreads + tid; i < kNetSyncWords; i += gridDim.x * kNetThreads) a.sync_next[i] = 0u;

  unsigned *place = reinterpret_cast<unsigned *>(smem_all);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;          // HW_REG_XCC_ID[3:0]
    place[0] = xcc;
    place[1] = xcc < 16u ? __hip_atomic_fetch_add(a.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  __syncthreads();
  const int xcc = __builtin_amdgcn_readfirstlane((int)place[0]);
  const int ticket = __builtin_amdgcn_readfirstlane((int)place[1]);
  __syncthreads();
  if (xcc >= a.num_xcd) {                        // not the device this launch was laid out for
    if (tid == 0) __hip_atomic_store(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if (ticket >= kWgPerXcd) return;               // more workgroups than tiles' worth on this XCD
  const int rb_lo = xcc * a.row_blocks / a.num_xcd, rb_hi = (xcc + 1) * a.row_blocks / a.num_xcd;
  unsigned *cnt = a.sync + 16;

  if (tid == 0) g_stamps[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
  for (int l = 0; l < a.num_layers; ++l) {
    if (tid == 0 && l > 0) g_stamps[blockIdx.x * 16 + l] = __builtin_amdgcn_s_memrealtime();
    const NetLayer &L = a.layer[l];
    const bool rows_out = a.rows_out_last && l == a.num_layers - 1;
    const float *act = l == 0 ? a.in : a.buf[(a.first_buf + l - 1) & 1];
    const int64_t ld_src = l == 0 ? a.ld_in : a.ld_act;
    float *dst = rows_out ? a.out_last : a.buf[(a.first_buf + l) & 1];
    const int ntiles = (rb_hi - rb_lo) * L.col_tiles;
    for (int q = ticket; q < ntiles; q += kWgPerXcd) {
      const int rb = rb_lo + q / L.col_tiles, c = q % L.col_tiles;
      if (l > 0 && tid == 0) {
        const unsigned need = (unsigned)a.layer[l - 1].col_tiles;
        int spins = 0;
        while (__hip_atomic_load(cnt + (l - 1) * kNetMaxRowBlocks + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          if (++spins > kSpinLimit) {
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();       // the row block is ready; every wave is done with the previous tile's LDS
      Tile t;
      t.K = L.K; t.splice_dim = l == 0 ? a.splice_dim : 0;
      t.bias = L.bias; t.relu = L.relu;
      t.out = dst;
      if (!rows_out) {       // out[feature][frame]
        t.P = L.wt; t.ldp = L.ldw; t.i0 = c * kNT;
        t.Q = act; t.ldq = ld_src; t.j0 = rb * kNT;
        t.bias_on_j = 0; t.ldo = a.ld_act;
        if (l == 0) {
          if (a.splice_dim > 0) ComputeTile<true, 0>(t, smem_all, lane, grp, wave);
          else ComputeTile<false, 0>(t, smem_all, lane, grp, wave);
        } else {
          ComputeTile<false, 2>(t, smem_all, lane, grp, wave);
        }
      } else {               // out[frame][feature]
        t.P = act; t.ldp = ld_src; t.i0 = rb * kNT;
        t.Q = L.wt; t.ldq = L.ldw; t.j0 = c * kNT;
        t.bias_on_j = 1; t.ldo = a.ld_last;
        if (l == 0) ComputeTile<false, 0>(t, smem_all, lane, grp, wave);
        else ComputeTile<false, 1>(t, smem_all, lane, grp, wave);
      }
      // publish: this workgroup's stores have reached the L2, then the row block's count goes up
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt + l * kNetMaxRowBlocks + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) { g_stamps[blockIdx.x * 16 + a.num_layers] = __builtin_amdgcn_s_memrealtime(); g_stamps[blockIdx.x * 16 + 15] = (unsigned long long)(xcc * 100 + ticket); }
}

}  // namespace

void LaunchNet(const NetArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(NetKernel, dim3(kWgPerXcd * a.num_xcd), dim3(kNetThreads), 0, stream, a);
  if (getenv("PK_MI355_NET_STAMPS")) {
    static int calls = 0;
    if (++calls == 30) {
      hipStreamSynchronize(stream);
      static unsigned long long h[256 * 16];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < 256; ++w) t0 = h[w * 16] < t0 ? h[w * 16] : t0;
      for (int l = 0; l <= a.num_layers; ++l) {
        double mn = 1e30, mx = 0, sum = 0; int n = 0;
        for (int w = 0; w < 256; ++w) {
          if (h[w * 16 + 15] % 100 >= 32) continue;
          double v = (h[w * 16 + l] - t0) * 0.01; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; ++n;
        }
        fprintf(stderr, "stamp %d (start of layer / end): min %.2f avg %.2f max %.2f us over %d workgroups\n", l, mn, sum / n, mx, n);
      }
    }
  }
}

}  // namespace pkmi
