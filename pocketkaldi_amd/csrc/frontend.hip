// frontend.hip -- fbank (window -> 512-point split-radix real FFT -> power ->
// mel -> log) and sliding-window CMVN for gfx950.
//
// Parity strategy: every float expression is evaluated with the same operations
// in the same order as the reference (fbank.cc, srfft.cc, cmvn.cc), so results
// are bit-identical wherever IEEE arithmetic is; the file is compiled with
// -ffp-contract=off so a*b+c is never fused where the reference's x86-64 build
// does not fuse.  The only non-IEEE step is the final natural log (fbank.cc:245):
// it is the C library's own table-driven logf restated for the device (pk_logf.h,
// checked against the system libm on every positive float), so it too is bit-identical.
//
// Execution shape: one 64-lane wavefront per frame, four independent waves per
// workgroup; frame and constant tables live in LDS, the eight split-radix passes each
// run <= 64 independent "L" butterflies -- one per lane -- with a wave-level barrier
// between passes.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pk_dma.h"
#include "pk_kernels.h"
#include "pk_logf.h"
#include "pk_wave.h"

#pragma clang fp contract(off)

namespace pkmi {

namespace {

constexpr int kWave = 64;

// One split-radix "L" butterfly on points n, n+m/4, n+m/2, n+3m/4 of the block at
// `base`: srfft.cc:163-173 (radix-2 step), :176-188 (+-j step), :198-222
// (twiddles).  The reference sweeps each step over the whole block; the quads
// are disjoint so doing all three steps per quad rounds identically.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The frame is kept as interleaved complex numbers z[n] = (re, im): the windowed samples
// y[0..511] ARE that array (srfft.cc:296-303 de-interleaves because its recursion wants
// split arrays; the arithmetic is the same), and every butterfly moves whole complex
// values, so each LDS access is one 8-byte word.
// A complex point of the frame by its LDS byte address (a 16-bit half of a packed pair, FftLane): the address goes
// into the ds instruction as it is -- one vector instruction to unpack a point instead of two (extract the slot
// byte, shift-and-add the frame's base), 28 fewer per frame over the seven passes.
typedef f32x2 __attribute__((address_space(3))) *LdsPoint;
__device__ __forceinline__ LdsPoint PointLo(unsigned pair) { return (LdsPoint)(pair & 0xffffu); }
__device__ __forceinline__ LdsPoint PointHi(unsigned pair) { return (LdsPoint)(pair >> 16); }
__device__ __forceinline__ unsigned LdsAddressOf(const void *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

template <int LOGM>
__device__ __forceinline__ void LButterfly(unsigned p01, unsigned p23, int n, const f32x2 (&tw)[3]) {
  constexpr int m = 1 << LOGM, m8 = m / 8;
  // the LDS addresses of points n, n + m/4 (p01) and n + m/2, n + 3m/4 (p23) of the block (FftLane::pts); unpacked per
  // frame (kept unpacked across the frame loop they cost 40 registers and a wave per SIMD)
  asm volatile("" : "+v"(p01), "+v"(p23));
  const LdsPoint i0 = PointLo(p01), i1 = PointHi(p01), i2 = PointLo(p23), i3 = PointHi(p23);
  const f32x2 z0 = *i0, z1 = *i1, z2 = *i2, z3 = *i3;

  // srfft.cc:163-173: (a, c) = (z0 + z2, z1 + z3) go back to points 0, 1; b = z0 - z2, d = z1 - z3 feed the +-j step.
  // Everything is kept as the PAIRS the packed fp32 instructions want (one instruction per pair, same roundings):
  //   x = (r2, q1) = (br - di, bi - dr)      y = (q2, r1) = (bi + dr, br + di)          (srfft.cc:176-188)
  // with r1 -> re[i2], q1 -> im[i2], r2 -> re[i3], q2 -> im[i3].
  const f32x2 a = z0 + z2, b = z0 - z2, c = z1 + z3, d = z1 - z3;
  *i0 = a;
  *i1 = c;
  const f32x2 ds = __builtin_shufflevector(d, d, 1, 0);
  f32x2 x = b - ds;
  const f32x2 yp = b + ds;
  f32x2 y = __builtin_shufflevector(yp, yp, 1, 0);

  // srfft.cc:198-222, the twiddles.  The general form runs on every lane of the butterfly; the lanes of the two special
  // positions then overwrite its result under their own execution masks (n = m/8: the +-sqrt(1/2) form; n = 0: no
  // twiddle) -- three packed instructions and two moves, written out because hipcc turns the same thing as C++ into
  // either nested branches with a dozen register copies at the joins, or sixteen selects.
  //   general:  t2 = (c3n, cn) (x + y);  o1 = (smc3n, spcn) y + t2 = (r2', q1');  o2 = (spc3n, smcn) x + t2 = (q2', r1')
  //   n = m/8:  o1 = sq (q2 - r2, q1 - r1);  o2 = (-sq, sq) (r2 + q2, q1 + r1)
  if (LOGM >= 3) {
    const f32x2 s = x + y;                       // (r2 + q2, q1 + r1)
    f32x2 o1 = x, o2 = y;
    if (LOGM >= 4) {
      const f32x2 t2 = tw[0] * s;
      o1 = tw[1] * y + t2;
      o2 = tw[2] * x + t2;
    }
    const float sq = 0.70710678118654752440;     // srfft.cc:41-43, narrowed like `float sqhalf`
    const f32x2 sq2 = f32x2{sq, sq};
    const unsigned long long at_m8 = __builtin_amdgcn_ballot_w64(n == m8);
    const unsigned long long at_0 = __builtin_amdgcn_ballot_w64(LOGM >= 4 && n == 0);    // (LOGM = 3: o already is (x, y))
    unsigned long long saved;
    f32x2 u;
    asm("s_and_saveexec_b64 %[sv], %[m8]\n\t"
        "v_pk_add_f32 %[u], %[y], %[x] neg_lo:[0,1] neg_hi:[1,0]\n\t"       // (q2 - r2, q1 - r1)
        "v_pk_mul_f32 %[o1], %[u], %[sq2]\n\t"
        "v_pk_mul_f32 %[o2], %[s], %[sq2] neg_lo:[0,1]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_and_saveexec_b64 %[sv], %[m0]\n\t"
        "v_mov_b64 %[o1], %[x]\n\t"
        "v_mov_b64 %[o2], %[y]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [o1] "+v"(o1), [o2] "+v"(o2), [u] "=&v"(u), [sv] "=&s"(saved)
        : [x] "v"(x), [y] "v"(y), [s] "v"(s), [sq2] "s"(sq2), [m8] "s"(at_m8), [m0] "s"(at_0)
        : "vcc", "scc");
    x = o1;
    y = o2;
  }
  // re[i2] = r1', im[i2] = q1', re[i3] = r2', im[i3] = q2': the halves sit in different register pairs, and
  // ds_write2_b32 takes its two words from two registers -- as one 8-byte store each pair would first be regrouped by
  // three moves a pass.  (WaveSync follows every pass: its lgkmcnt(0) covers these.)
  asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(i2), "v"(y[1]), "v"(x[1]) : "memory");
  asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(i3), "v"(x[0]), "v"(y[0]) : "memory");
}

// The constant tables of the front-end, copied once per workgroup into LDS (the
// kernel is latency-bound: a table value fetched from L2 inside every FFT pass costs
// more than the butterfly it feeds).  Layout and narrowing are the host's (pk_tables.h).
typedef FrontendLdsImage LdsTables;

// Per-wave work area: one frame.
struct alignas(16) FrameLds {
  float x[kFrameLength];       // general float input only: the samples in order, for the sequential DC sum
  f32x2 z[kFftCplx];           // the 256 complex points (= the 512 windowed samples)
  float pw[kFftCplx + 4];      // power spectrum (+ slack: a mel padding tap may read up to three slots behind bin 256)
};
// x and z are dead once the power spectrum is written: the mel products go there
static_assert(sizeof(float) * kMelProducts <= sizeof(float) * kFrameLength + sizeof(f32x2) * kFftCplx, "mel products fit in x + z");
static_assert((sizeof(float) * kFrameLength) % 16 == 0 && sizeof(FrameLds) % 16 == 0, "16-byte rows");

// The four waves of a workgroup work on different frames and never exchange data, so
// the barrier between FFT passes is wave-local: LDS operations of one wave complete in
// issue order; what is needed is that they are finished and not reordered by hipcc.
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What a lane needs from the tables for EVERY frame it transforms -- where its butterfly of each
// pass starts, that butterfly's six coefficients, its two-point blocks, its two bins of the real
// post-pass -- read from the LDS tables once, kept in registers.  (The kernel is LDS-bound:
// SQ_ACTIVE_INST_LDS fills the CU's LDS pipe, profiles/r02_S_pmc_valu.json; these reads were
// 60 of a frame's 154 LDS instructions.)
// LDS slot of complex point i.  With the plain layout the late passes (many small blocks, one lane
// each), the two-point blocks and above all the bit-reversed reads of the post-pass put 4 to 16 lanes
// of every 16 on the same banks: SQ_LDS_BANK_CONFLICT was 42 % of the kernel's CU-busy cycles.  The
// low four bits of the index are XOR-ed with a GF(2)-linear image of the high four; the map was
// chosen by enumerating all 65 536 of them against the kernel's exact access lists
// (tools/fft_lds_swizzle.py): 560 -> 60 conflict cycles per frame.
__device__ __forceinline__ int FftSlot(int i) {
  const int h = i >> 4;
  const int x = ((h & 1) ? 10 : 0) ^ ((h & 2) ? 5 : 0) ^ ((h & 4) ? 8 : 0) ^ ((h & 8) ? 4 : 0);
  return i ^ x;
}

// Points are held as LDS byte addresses of THIS wave's frame, two 16-bit halves per register.
struct FftLane {
  unsigned pts[kNumPasses - 1][2];   // passes LOGM = 8..2: this lane's four points (n, n + m/4 | n + m/2, n + 3m/4)
  unsigned active;                   // bit p: this lane has a butterfly in pass p; bits 8, 9: a two-point block
  f32x2 tw[5][3];                    // LOGM = 8..4, as the pairs the butterfly multiplies by (srfft.cc:45-93):
                                     // (c3n, cn), (s3-c3, -(s+c)), (-(s3+c3), s-c)
  unsigned two[2];                   // the two-point blocks (srfft.cc:140-150): (a0, a0 + 1) of block lane, of block lane + 64
  unsigned post[2];                  // post-pass: bins k = 1 + lane + 64 r and 256 - k, bit-reversed
  float kre[2], kim[2];              // exp(-2 pi i k / 512) as srfft.cc:385-394 builds it
  unsigned stage[2];                 // the points lane + 64 r the lane writes when it windows a frame (r = 0, 1 | 2, 3)
};

__device__ __forceinline__ void MakeFftLane(FftLane &c, int lane, const LdsTables &tab, const f32x2 *z) {
  const unsigned zb = LdsAddressOf(z);
  auto at = [&](int i) { return zb + 8u * (unsigned)FftSlot(i); };       // (the whole LDS of these kernels is below 64 KiB)
  c.active = 0;
#pragma unroll
  for (int pass = 0; pass < kNumPasses - 1; ++pass) {
    const int logm = kLogCplx - pass;
    const int q = (1 << logm) / 4;            // butterflies per block
    const int first = tab.pass_start[pass];
    const int nblk = tab.pass_start[pass + 1] - first;
    const int b = lane / q, n = lane % q;
    c.pts[pass][0] = c.pts[pass][1] = 0;
    if (b < nblk) {
      const int i0 = tab.blk_off[first + b] + n;
      c.pts[pass][0] = at(i0) | (at(i0 + q) << 16);
      c.pts[pass][1] = at(i0 + 2 * q) | (at(i0 + 3 * q) << 16);
      c.active |= 1u << pass;
    }
    if (logm >= 4) {
      const float *tw = tab.tw + tab.tw_off[logm];      // six arrays of q: cn, -(s+c), s-c, c3n, -(s3+c3), s3-c3
      c.tw[pass][0] = f32x2{tw[3 * q + n], tw[0 * q + n]};
      c.tw[pass][1] = f32x2{tw[5 * q + n], tw[1 * q + n]};
      c.tw[pass][2] = f32x2{tw[4 * q + n], tw[2 * q + n]};
    }
  }
  const int first = tab.pass_start[kNumPasses - 1];
  const int nblk = tab.pass_start[kNumPasses] - first;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int b = lane + kWave * r;
    c.two[r] = 0;
    if (b < nblk) {
      const int off = tab.blk_off[first + b];
      c.two[r] = at(off) | (at(off + 1) << 16);
      c.active |= 1u << (8 + r);
    }
    const int k = 1 + lane + kWave * r;
    c.post[r] = at(tab.bitrev[k]) | (at(tab.bitrev[kFftCplx - k]) << 16);
    c.kre[r] = tab.post_re[k];
    c.kim[r] = tab.post_im[k];
    c.stage[r] = at(lane + kWave * 2 * r) | (at(lane + kWave * (2 * r + 1)) << 16);
  }
}

template <int LOGM>
__device__ __forceinline__ void FftPass(int lane, const FftLane &c) {
  constexpr int pass = kLogCplx - LOGM;
  constexpr int q = (1 << LOGM) / 4;
  if (c.active & (1u << pass)) LButterfly<LOGM>(c.pts[pass][0], c.pts[pass][1], lane % q, c.tw[pass < 5 ? pass : 0]);
  WaveSync();
}

// One flat copy, 16 bytes per lane, every load in flight before the first store.
__device__ __forceinline__ void CopyTablesToLds(LdsTables &tab, const FrontendTables *__restrict__ gtab) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int kPieces = (int)(sizeof(LdsTables) / 16);
  constexpr int kPerThread = (kPieces + 63) / 64;            // enough for the smallest workgroup (one wave)
  const u32x4 *src = reinterpret_cast<const u32x4 *>(&gtab->lds);
  u32x4 *dst = reinterpret_cast<u32x4 *>(&tab);
  const int tid = threadIdx.x, nt = blockDim.x;
  u32x4 v[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int i = tid + k * nt;
    if (i < kPieces) v[k] = src[i];
  }
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int i = tid + k * nt;
    if (i < kPieces) dst[i] = v[k];
  }
}

// srfft.cc:95-237: the 256-point complex split-radix DIF on the interleaved frame in LDS, one
// wave, as eight passes over the block schedule; output in bit-reversed order.
__device__ __forceinline__ void ComplexFft256(int lane, const FftLane &c) {
  FftPass<8>(lane, c);
  FftPass<7>(lane, c);
  FftPass<6>(lane, c);
  FftPass<5>(lane, c);
  FftPass<4>(lane, c);
  FftPass<3>(lane, c);
  FftPass<2>(lane, c);
#pragma unroll
  for (int r = 0; r < 2; ++r) {   // two-point blocks, srfft.cc:140-150
    if (c.active & (1u << (8 + r))) {
      const LdsPoint a0 = PointLo(c.two[r]), a1 = PointHi(c.two[r]);
      const f32x2 u0 = *a0, u1 = *a1;
      *a0 = f32x2{u0[0] + u1[0], u0[1] + u1[1]};
      *a1 = f32x2{u0[0] - u1[0], u0[1] - u1[1]};
    }
  }
  WaveSync();
}

// The real-FFT post-pass for bin k = 1 + lane + 64 r (1..128) and its partner 256 - k:
// srfft.cc:389-436, with the bit-reversed read of srfft.cc:239-265 folded in.  (a_re, a_im) is bin k,
// (o_re, o_im) bin 256 - k.
struct RealBins { float a_re, a_im, o_re, o_im; };
__device__ __forceinline__ RealBins RealPostPass(const FftLane &c, int r) {
  const f32x2 zk = *PointLo(c.post[r]), zd = *PointHi(c.post[r]);
  const float bk_re = zk[0], bk_im = zk[1];
  const float bd_re = zd[0], bd_im = zd[1];
  const float kre = c.kre[r], kim = c.kim[r];
  const float ck_re = 0.5f * (bk_re + bd_re);
  const float ck_im = 0.5f * (bk_im - bd_im);
  const float dk_re = 0.5f * (bk_im + bd_im);
  const float dk_im = -0.5f * (bk_re - bd_re);
  RealBins rb;
  rb.a_re = ck_re; rb.a_im = ck_im;
  rb.a_re += kre * dk_re - kim * dk_im;
  rb.a_im += kre * dk_im + kim * dk_re;
  const float nk_re = -kre, ndk_im = -dk_im;
  rb.o_re = ck_re; rb.o_im = -ck_im;
  rb.o_re += nk_re * dk_re - kim * ndk_im;
  rb.o_im += nk_re * ndk_im + kim * dk_re;
  return rb;
}

constexpr int kFbankWaves = 4;

// One tap of a mel triangle: w * p, one rounding (vector.cc:252-262).  v_mul_legacy_f32 is v_mul_f32 except that
// 0 * x = 0 for EVERY x: a padding tap (weight 0) reads a power-spectrum slot outside its triangle, and must
// contribute +0 even when that slot is inf or NaN; a real tap's weight is never zero (fbank.cc:141: strict inequalities).
__device__ __forceinline__ float MelProduct(float w, float p) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(w), "v"(p));
  return r;
}

// Each wave of a workgroup walks frames t = blockIdx.x * 4 + wave, + gridDim.x * 4, ...
// of utterance blockIdx.y.
template <typename SampleT>
__global__ __launch_bounds__(kWave * kFbankWaves, 4) void FbankKernel(
    const SampleT *__restrict__ wave_pcm, UttLayout utts, const FrontendTables *__restrict__ gtab,
    float *__restrict__ raw) {
  __shared__ LdsTables tab;
  __shared__ FrameLds frames[kFbankWaves];

  // ---- tables: global -> LDS, once per workgroup
  CopyTablesToLds(tab, gtab);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  FrameLds &fr = frames[wv];
  FftLane fc;
  MakeFftLane(fc, lane, tab, fr.z);
  float *s_x = fr.x, *s_pow = fr.pw;
  f32x2 *s_z = fr.z;
  // mel stage (pk_tables.h: kMelProducts): this lane forms products 4 lane + e, 256 + 4 lane + e (e < 4) and 512 + lane;
  // lanes < 40 then add the products of their bin in order
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  float *s_prod = fr.x;
  const float *mel_p0 = s_pow + tab.mel_pgrp[lane], *mel_p1 = s_pow + tab.mel_pgrp[64 + lane];
  const float *mel_p2 = s_pow + tab.mel_pgrp[128 + (lane >> 2)] + (lane & 3);
  const float *mel_run = s_prod + (lane < kNumBins ? tab.mel_pbase[lane] : 0);
  const int mel_plen = lane < kNumBins ? tab.mel_plen[lane] : 0;
  const int utt = blockIdx.y;
  const int T = utts.num_frames[utt];
  const SampleT *w0 = wave_pcm + utts.wave_off[utt];
  float *out0 = raw + utts.raw_base[utt] * kNumBins;

  // ---- fbank.cc:74-100: a frame's 400 samples as 200 (even, odd) pairs, pair p = lane + 64 r
  // (r = 0..3): pair p IS point p of the half-size complex FFT (sample 2p real, 2p + 1 imaginary),
  // so a lane stores whole points, and the predecessor sample the pre-emphasis needs is the lane's
  // own even sample or its left neighbour's odd one -- a DPP shift, no trip through LDS.  The
  // samples of the wave's NEXT frame are fetched before the current frame is processed.
  const int t_step = gridDim.x * kFbankWaves;
  auto fetch = [&](int t, SampleT (&de)[4], SampleT (&dod)[4]) {
    const SampleT *w = w0 + (int64_t)t * kFrameShift;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pidx = lane + kWave * r;
      const bool in = t < T && 2 * pidx + 1 < kFrameLength;
      // one load per pair; an utterance may start at any sample, so the pair is only element-aligned
      struct __attribute__((packed, aligned(sizeof(SampleT)))) Pair { SampleT e, o; };
      Pair pr = {static_cast<SampleT>(0), static_cast<SampleT>(0)};
      if (in) pr = *reinterpret_cast<const Pair *>(w + 2 * pidx);
      de[r] = pr.e;
      dod[r] = pr.o;
    }
  };
  SampleT cur_e[4], cur_o[4], next_e[4], next_o[4];
  fetch(blockIdx.x * kFbankWaves + wv, cur_e, cur_o);
  for (int t = blockIdx.x * kFbankWaves + wv; t < T; t += t_step) {
    fetch(t + t_step, next_e, next_o);
    float xe[4], xo[4];
    bool exact = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xe[r] = static_cast<float>(cur_e[r]);
      xo[r] = static_cast<float>(cur_o[r]);
      // integer-valued samples of at most 16 bits: any summation order is exact (int16 input: by type)
      if (!std::is_same<SampleT, int16_t>::value)
        exact = exact && (fabsf(xe[r]) <= 32768.0f) && (xe[r] == truncf(xe[r]))
                      && (fabsf(xo[r]) <= 32768.0f) && (xo[r] == truncf(xo[r]));
    }

    // ---- fbank.cc:48-52: DC offset = sequential float sum / 400.
    float sum;
    if (__all(exact)) {
      // 400 integers of magnitude <= 2^15: every partial sum is an integer below
      // 2^24, so the tree sum equals the reference's sequential sum bit for bit.
      const float p = ((xe[0] + xo[0]) + (xe[1] + xo[1])) + ((xe[2] + xo[2]) + (xe[3] + xo[3]));
      sum = TwWaveSum(p);
    } else {
      // general float input: keep the reference's order (one lane, 400 adds)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pidx = lane + kWave * r;
        if (2 * pidx + 1 < kFrameLength) { s_x[2 * pidx] = xe[r]; s_x[2 * pidx + 1] = xo[r]; }
      }
      WaveSync();
      float s = 0;
      if (lane == 0) {
#pragma unroll 8        // (a rare path: its temporaries must not set the kernel's register count)
        for (int i = 0; i < kFrameLength; ++i) s += s_x[i];
      }
      sum = __shfl(s, 0);
      WaveSync();
    }
    const float mean = sum / kFrameLength;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                     // fbank.cc:53-55
      xe[r] -= mean;
      xo[r] -= mean;
    }

    // ---- fbank.cc:58-68: pre-emphasis in double (0.97 is a double literal), one rounding to
    // float, then the Hamming window; zero padding 400..511.  x[i] -= 0.97 x[i - 1] runs from the
    // top down, so the predecessor is the not-yet-emphasised sample; x[0] takes itself.
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pidx = lane + kWave * r;
      // sample 2p - 1 = the odd sample of pair p - 1: lane - 1 of the same r, or lane 63 of r - 1
      float prev_e = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xo[r]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
      if (lane == 0) prev_e = r == 0 ? xe[0] : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xo[r > 0 ? r - 1 : 0]), 63));
      // (the Hamming window at samples 2p, 2p + 1: one 8-byte LDS read a frame instead of eight registers all along)
      const f32x2 win = *reinterpret_cast<const f32x2 *>(tab.window + 2 * (pidx < kFrameLength / 2 ? pidx : 0));
      float ye = 0.0f, yo = 0.0f;
      if (2 * pidx + 1 < kFrameLength) {
        ye = static_cast<float>(static_cast<double>(xe[r]) - 0.97 * static_cast<double>(prev_e));
        ye *= win[0];
        yo = static_cast<float>(static_cast<double>(xo[r]) - 0.97 * static_cast<double>(xe[r]));
        yo *= win[1];
      }
      *((r & 1) ? PointHi(fc.stage[r >> 1]) : PointLo(fc.stage[r >> 1])) = f32x2{ye, yo};
    }
    WaveSync();

    // ---- srfft.cc:95-237 as passes over the block schedule
    ComplexFft256(lane, fc);

    // ---- bit-reversed read (srfft.cc:239-265), real post-pass (srfft.cc:389-436)
    // and power spectrum (fbank.cc:193-211) fused: bin k and its partner 256-k.
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = 1 + lane + kWave * r;            // 1..128
      const int kd = kFftCplx - k;
      const RealBins rb = RealPostPass(fc, r);
      s_pow[k] = rb.a_re * rb.a_re + rb.a_im * rb.a_im;
      if (kd != k) s_pow[kd] = rb.o_re * rb.o_re + rb.o_im * rb.o_im;
    }
    if (lane == 0) {                                 // srfft.cc:444-447, fbank.cc:201-210
      const f32x2 z00 = s_z[0];
      const float zeroth = z00[0] + z00[1], n2th = z00[0] - z00[1];
      s_pow[0] = zeroth * zeroth;
      s_pow[kFftCplx] = n2th * n2th;
    }
    WaveSync();

    // ---- fbank.cc:165-184: E[b] = sum_j w_b[j] * P[off_b + j], a sequential float dot per bin (vector.cc:252-262).
    // The products first, nine per lane over all 64 lanes (a product is one rounding whichever lane forms it) ...
    {
      const f32x4 w0 = *reinterpret_cast<const f32x4 *>(tab.mel_wprod + 4 * lane);
      const f32x4 w1 = *reinterpret_cast<const f32x4 *>(tab.mel_wprod + 256 + 4 * lane);
      const float w2 = tab.mel_wprod[512 + lane];
      f32x4 q0, q1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q0[e] = MelProduct(w0[e], mel_p0[e]);
        q1[e] = MelProduct(w1[e], mel_p1[e]);
      }
      *reinterpret_cast<f32x4 *>(s_prod + 4 * lane) = q0;
      *reinterpret_cast<f32x4 *>(s_prod + 256 + 4 * lane) = q1;
      s_prod[512 + lane] = MelProduct(w2, mel_p2[0]);
    }
    WaveSync();
    // ... then the additions in the reference's order, one lane per bin (the run is padded with +0 products to a
    // multiple of four); floor FLT_EPSILON and log (fbank.cc:244-245; the C library's logf, pk_logf.h)
    if (lane < kNumBins) {
      float e = 0.0f;
      // (straight-line: which lanes take part in group g does not change from frame to frame -- the comparisons are
      // loop-invariant execution masks, the groups cost their four additions and nothing else)
#pragma unroll
      for (int g = 0; g < kMelMaxPadded / 4; ++g) {
        if (4 * g < mel_plen) {
          const f32x4 q = *reinterpret_cast<const f32x4 *>(mel_run + 4 * g);
          e += q[0];
          e += q[1];
          e += q[2];
          e += q[3];
        }
      }
      if (e < 1.1920928955078125e-07f) e = 1.1920928955078125e-07f;
      out0[(int64_t)t * kNumBins + lane] = LogfRestated(e, tab.logf_tab);
    }
    WaveSync();
#pragma unroll
    for (int r = 0; r < 4; ++r) { cur_e[r] = next_e[r]; cur_o[r] = next_o[r]; }
  }
}

// One workgroup of four wavefronts per utterance.  The running window sum is rounded to float
// every frame (cmvn.cc:66-70), so ONE wavefront walks the frames in order, one lane per
// feature, and does nothing but that chain -- one f32 add while the window fills (for floats,
// float(double(s) + double(x)) IS s + x: the double sum is exact or differs from it by less than
// a quarter of a float ulp), widen / two fp64 adds / narrow once it slides -- leaving S_t in LDS.
// Everything else is parallel over frames and belongs to the other waves: one moves frames
// HBM -> LDS by LDS-DMA, kCmvnAhead tiles (64 frames, 10 KiB each) ahead of the chain with a
// counted wait, so that the chain never waits for memory (with one tile of lead the kernel ran at
// one DMA round trip per tile: 2.9 us per 64 frames); two turn (x_t, S_t) of the previous tile
// into y_t (lane = frame, 16-byte LDS reads of four features) and store rows of 64 consecutive
// frames (256 bytes) into the feature-major operand of the first affine layer.
// The window count is min(t + 1, 600), so the smoothing weight and the 1/count scale come
// from the host-built CmvnTables.  One raw s_barrier per tile.
constexpr int kCmvnTile = 64;                                   // frames per tile
constexpr int kCmvnTileFloats = kCmvnTile * kNumBins;           // 2560 floats = 10 KiB
constexpr int kCmvnWaves = 4;                                   // chain, two writers, loader
constexpr int kCmvnAhead = 3;                                   // tiles requested beyond the chain's
constexpr int kCmvnInSlots = kCmvnAhead + 2;                    // writers' tile, chain's tile, kCmvnAhead ahead
constexpr int kCmvnOldSlots = kCmvnAhead + 1;
constexpr int kCmvnPieces = kCmvnTileFloats / 256;              // 1 KiB DMA instructions per tile: 10
static_assert(kCmvnRawSlack >= kCmvnTileFloats, "the loader reads whole tiles");
static_assert(kCmvnRawLead >= kCmvnWindow * kNumBins - (kCmvnWindow / kCmvnTile) * kCmvnTileFloats,
              "the first sliding tile starts this far before frame 0");

// s_waitcnt vmcnt(n) for the run-time multiples of kCmvnPieces the loader needs (the count is an
// immediate; the loader wave branches on a wave-uniform value)
__device__ __forceinline__ void CmvnWaitOutstanding(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
  }
}
static_assert(kCmvnPieces == 10 && kCmvnAhead <= 3, "CmvnWaitOutstanding covers at most 2 tiles x 20 pieces");

__global__ __launch_bounds__(kWave * kCmvnWaves) void CmvnKernel(
    const float *__restrict__ raw, UttLayout utts, const float *__restrict__ g,
    const CmvnTables *__restrict__ tab, int left, int right, float *__restrict__ yt, int64_t ldy) {
  // ONE LDS array, carved by hand: with several __shared__ objects hipcc waits for the
  // in-flight LDS-DMA before every LDS read
  constexpr int kLdsFloats = (kCmvnInSlots + kCmvnOldSlots + 2) * kCmvnTileFloats + 2 * kCmvnWindow + 64;
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float *s_in = lds;                                            // [kCmvnInSlots][tile]: x[t][d]
  float *s_old = s_in + kCmvnInSlots * kCmvnTileFloats;         // [kCmvnOldSlots][tile]: x[t - 600][d]
  float *s_sum = s_old + kCmvnOldSlots * kCmvnTileFloats;       // [2][tile]: S_t[d], the window sums after frame t
  float *s_alpha = s_sum + 2 * kCmvnTileFloats, *s_nscale = s_alpha + kCmvnWindow;
  float *s_g = s_nscale + kCmvnWindow;                          // the 40 global sums

  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 chain, 1-2 writers, 3 loader
  const int utt = blockIdx.x;
  const int T = utts.num_frames[utt];
  if (T <= 0) return;
  const float *x0 = raw + utts.raw_base[utt] * kNumBins;
  float *y0 = yt + utts.pad_base[utt];

  const int ntiles = (T + kCmvnTile - 1) / kCmvnTile;
  auto slides = [](int i) { return (i + 1) * kCmvnTile > kCmvnWindow; };   // some frame of tile i has t >= 600
  auto pieces = [&](int i) { return i < ntiles ? (slides(i) ? 2 * kCmvnPieces : kCmvnPieces) : 0; };
  // tile i of the utterance -> s_in[i % kCmvnInSlots], the frames leaving the window ->
  // s_old[i % kCmvnOldSlots].  Scalar-base LDS-DMA (pk_dma.h): a piece is 1 KiB of consecutive
  // floats, so the source is a wave-uniform base + lane * 16 bytes and the loader issues no
  // vector arithmetic at all.  The last tile reads past the utterance (into the next utterance,
  // or into the kCmvnRawSlack floats every raw buffer carries behind its last frame) and the
  // first sliding tile starts 24 frames before it (the previous utterance, or the kCmvnRawLead
  // floats in front of the first frame): values nobody uses, from memory that exists.
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto fetch = [&](int i) {
    float *din = s_in + (i % kCmvnInSlots) * kCmvnTileFloats, *dold = s_old + (i % kCmvnOldSlots) * kCmvnTileFloats;
    const bool sl = slides(i);
    const int e0 = i * kCmvnTileFloats;                      // float index of the tile in the utterance
#pragma unroll
    for (int p = 0; p < kCmvnPieces; ++p) {
      DmaScalarBase(din + p * 256, reinterpret_cast<const char *>(x0 + e0 + p * 256), lane_off);
      if (sl) {
        const int o = e0 + p * 256 - kCmvnWindow * kNumBins;      // >= -kCmvnRawLead
        DmaScalarBase(dold + p * 256, reinterpret_cast<const char *>(x0 + o), lane_off);
      }
    }
  };
  auto tile_sync = [&]() {
    __builtin_amdgcn_s_waitcnt(0xC07F);          // this wave's LDS traffic is done (lgkmcnt(0))
    __builtin_amdgcn_s_barrier();
  };

  if (role == 3) {
#pragma unroll
    for (int i = 0; i < kCmvnAhead; ++i)
      if (i < ntiles) fetch(i);
    int later = 0;
#pragma unroll
    for (int i = 1; i < kCmvnAhead; ++i) later += pieces(i);
    CmvnWaitOutstanding(later);                  // tile 0 is in
  } else {
    // the per-frame scalars and the global sums, while the first tile is on its way
    for (int i = threadIdx.x; i < kCmvnWindow; i += kWave * (kCmvnWaves - 1)) { s_alpha[i] = tab->alpha[i]; s_nscale[i] = tab->neg_scale[i]; }
    if (threadIdx.x < kNumBins) s_g[threadIdx.x] = g[threadIdx.x];
  }
  tile_sync();

  float s = 0.0f;                                  // chain wave: cached window sum of feature `lane`
  const int d = lane < kNumBins ? lane : 0;       // lanes 40..63 of the chain wave shadow feature 0
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 g4[kNumBins / 8];                          // writers: the global sums of this wave's five feature quads
  if (role == 1 || role == 2) {
#pragma unroll
    for (int qi = 0; qi < kNumBins / 8; ++qi) g4[qi] = *reinterpret_cast<const f32x4 *>(s_g + 4 * (role - 1 + 2 * qi));
  }
  for (int i = 0; i <= ntiles; ++i) {
    if (role == 3) {
      // request tile i + kCmvnAhead (its slot held tile i - 2, whose last readers passed the
      // previous barrier); tile i + 1 must have landed before this iteration's barrier
      if (i + kCmvnAhead < ntiles) fetch(i + kCmvnAhead);
      int later = 0;
#pragma unroll
      for (int a = 2; a <= kCmvnAhead; ++a) later += pieces(i + a);
      CmvnWaitOutstanding(later);
    } else if (role == 0) {
      if (i < ntiles) {
        const float *xin = s_in + (i % kCmvnInSlots) * kCmvnTileFloats, *xold = s_old + (i % kCmvnOldSlots) * kCmvnTileFloats;
        float *sum = s_sum + (i & 1) * kCmvnTileFloats;
        const int t0 = i * kCmvnTile;
        // The LDS operands of the WHOLE tile -- 64 values of x, and of x[t - 600] once the window
        // slides -- are requested up front, 128 registers; then the serial chain is arithmetic plus
        // one LDS store per frame.  (In blocks of 8 with the next block requested early, hipcc still
        // put a full lgkmcnt(0) in front of every block -- the counter has 4 bits, 16 younger
        // operations were in flight -- and the tile cost twice the chain's own time.)  Whole tiles
        // always: frames past the end of the utterance compute values nobody reads (their slots of
        // the tile exist), and lanes 40..63 store feature 0's value on top of lane 0's.
        // MODE 0: no frame of the tile has left the window yet; 2: all have; 1: mixed.
        auto walk = [&](auto mode) {
          constexpr int MODE = decltype(mode)::value;
          float xs[kCmvnTile], xo[MODE != 0 ? kCmvnTile : 1];
#pragma unroll
          for (int v = 0; v < kCmvnTile; ++v) {
            xs[v] = xin[v * kNumBins + d];
            if (MODE != 0) xo[v] = xold[v * kNumBins + d];
          }
#pragma unroll
          for (int v = 0; v < kCmvnTile; ++v) {
            double acc = s;                                 // cmvn.cc:44-52
            acc += xs[v];
            if (MODE == 2 || (MODE == 1 && t0 + v >= kCmvnWindow))
              acc += -1.0 * static_cast<double>(xo[MODE != 0 ? v : 0]);     // cmvn.cc:58-64
            s = static_cast<float>(acc);                    // cmvn.cc:66-70
            sum[v * kNumBins + d] = s;
          }
        };
        if (t0 + kCmvnTile <= kCmvnWindow) walk(std::integral_constant<int, 0>());
        else if (t0 >= kCmvnWindow) walk(std::integral_constant<int, 2>());
        else walk(std::integral_constant<int, 1>());
      }
    } else if (i >= 1) {
      // writers: tile i - 1, lane = frame, the ten feature quads split between the two waves
      const int j = i - 1;
      const float *xin = s_in + (j % kCmvnInSlots) * kCmvnTileFloats, *sum = s_sum + (j & 1) * kCmvnTileFloats;
      const int t = j * kCmvnTile + lane;
      const int tt = t < kCmvnWindow ? t : kCmvnWindow - 1;
      const float al = s_alpha[tt], ns = s_nscale[tt];
      const bool smooth = t + 1 < kCmvnWindow;
      float *col = y0 + left + t;                               // frame t of feature 0; feature d is d * ldy further
#pragma unroll
      for (int qi = 0; qi < kNumBins / 8; ++qi) {
        const int q = role - 1 + 2 * qi;
        const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xin + lane * kNumBins + 4 * q);
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sum + lane * kNumBins + 4 * q);
        float *dst = col + (int64_t)(4 * q) * ldy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float st = s4[e];
          if (smooth) st += al * g4[qi][e];                     // cmvn.cc:73-92 (count < window)
          float y = x4[e];
          y += ns * st;                                         // cmvn.cc:94-101
          if (t < T) dst[(int64_t)e * ldy] = y;
        }
      }
      // am.cc:73-75, the edge clamp, done once at write time: the first / last frame replicated
      // into the left / right pad.  Only the first and the last tile come here; lane = feature.
      if ((j == 0 && left > 0) || (j == ntiles - 1 && right > 0)) {
        const int dd = 4 * (role - 1 + 2 * (lane >> 2)) + (lane & 3);     // this wave's 20 features
        if (lane < kNumBins / 2) {
          for (int side = 0; side < 2; ++side) {
            if (side == 0 ? !(j == 0 && left > 0) : !(j == ntiles - 1 && right > 0)) continue;
            const int te = side == 0 ? 0 : T - 1;                         // the frame to replicate
            const int lt = te - j * kCmvnTile;
            const int te_c = te < kCmvnWindow ? te : kCmvnWindow - 1;
            float st = sum[lt * kNumBins + dd];
            if (te + 1 < kCmvnWindow) st += s_alpha[te_c] * s_g[dd];
            float y = xin[lt * kNumBins + dd];
            y += s_nscale[te_c] * st;
            float *row = y0 + (int64_t)dd * ldy;
            if (side == 0) for (int p = 0; p < left; ++p) row[p] = y;
            else for (int p = 0; p < right; ++p) row[left + T + p] = y;
          }
        }
      }
    }
    tile_sync();
  }
}

__global__ void PadTransposeKernel(const float *__restrict__ feats, int T, int dim, int left,
                                   int right, float *__restrict__ yt, int64_t ldy,
                                   int64_t col0) {
  const int total = T + left + right;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)total * dim;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int c = idx % total, d = idx / total;
    int t = c - left;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    yt[(int64_t)d * ldy + col0 + c] = feats[(int64_t)t * dim + d];
  }
}

}  // namespace

// Parity-test hook: LogfRestated on arbitrary inputs.
__global__ void LogfTestKernel(const float *__restrict__ x, int n, const FrontendTables *__restrict__ gtab,
                               float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = LogfRestated(x[i], gtab->logf_tab);
}

// Parity-test hook: pk_srfft_compute (srfft.cc:371-461) alone -- the FFT passes and the real
// post-pass of FbankKernel on caller-supplied 512-sample frames, packed output
// [Re0 + Im0, Re0 - Im0, Re1, Im1, ..., Re255, Im255] (srfft.cc:444-447).  One wave per frame.
__global__ __launch_bounds__(kWave) void Srfft512TestKernel(const float *__restrict__ frames, int n,
                                                            const FrontendTables *__restrict__ gtab,
                                                            float *__restrict__ out) {
  __shared__ LdsTables tab;
  __shared__ f32x2 s_z[kFftCplx];
  CopyTablesToLds(tab, gtab);
  __syncthreads();
  const int lane = threadIdx.x;
  FftLane fc;
  MakeFftLane(fc, lane, tab, s_z);
  for (int f = blockIdx.x; f < n; f += gridDim.x) {
    const float *x = frames + (int64_t)f * kFftSize;
    float *y = out + (int64_t)f * kFftSize;
    for (int i = lane; i < kFftCplx; i += kWave) s_z[FftSlot(i)] = f32x2{x[2 * i], x[2 * i + 1]};
    WaveSync();
    ComplexFft256(lane, fc);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = 1 + lane + kWave * r;
      const int kd = kFftCplx - k;
      const RealBins rb = RealPostPass(fc, r);
      y[2 * k] = rb.a_re;
      y[2 * k + 1] = rb.a_im;
      if (kd != k) { y[2 * kd] = rb.o_re; y[2 * kd + 1] = rb.o_im; }
    }
    if (lane == 0) {
      const f32x2 z00 = s_z[0];
      y[0] = z00[0] + z00[1];
      y[1] = z00[0] - z00[1];
    }
    WaveSync();
  }
}

void LaunchSrfft512Test(const float *frames, int n, const FrontendTables *d_tables, float *out, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(Srfft512TestKernel, dim3(n < 1024 ? n : 1024), dim3(kWave), 0, stream, frames, n, d_tables, out);
}

void LaunchLogfTest(const float *x, int n, const FrontendTables *d_tables, float *out, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(LogfTestKernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, d_tables, out);
}

void LaunchFbank(const float *wave_f32, const int16_t *wave_i16, const UttLayout &utts,
                 int num_utts, int max_frames, const FrontendTables *d_tables, float *raw,
                 hipStream_t stream) {
  if (num_utts <= 0 || max_frames <= 0) return;
  // ~12 workgroups per CU over the launch; every wave then walks several frames, which
  // amortises the table copy
  int gx = 3072 / num_utts;     // measured on 256 x 10 s: 1536 -> 0.57 ms, 3072 -> 0.51 ms, more: no change
  if (gx < 1) gx = 1;
  const int need = (max_frames + kFbankWaves - 1) / kFbankWaves;
  if (gx > need) gx = need;
  dim3 grid(gx, num_utts), block(kWave * kFbankWaves);
  if (wave_i16)
    hipLaunchKernelGGL(FbankKernel<int16_t>, grid, block, 0, stream, wave_i16, utts, d_tables, raw);
  else
    hipLaunchKernelGGL(FbankKernel<float>, grid, block, 0, stream, wave_f32, utts, d_tables, raw);
}

void LaunchCmvn(const float *raw, const UttLayout &utts, int num_utts, const float *d_global41,
                const CmvnTables *d_cmvn_tab, int left, int right, float *yt, int64_t ldy,
                hipStream_t stream) {
  if (num_utts <= 0) return;
  hipLaunchKernelGGL(CmvnKernel, dim3(num_utts), dim3(kWave * kCmvnWaves), 0, stream, raw, utts,
                     d_global41, d_cmvn_tab, left, right, yt, ldy);
}

void LaunchPadTranspose(const float *feats, int T, int dim, int left, int right, float *yt,
                        int64_t ldy, int64_t col0, hipStream_t stream) {
  if (T <= 0) return;
  int64_t n = (int64_t)(T + left + right) * dim;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(PadTransposeKernel, dim3(blocks), dim3(256), 0, stream, feats, T, dim, left,
                     right, yt, ldy, col0);
}

}  // namespace pkmi
