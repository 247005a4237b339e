// pk_tables.h -- constant tables of the front-end, built once on the host.
//
// The reference computes these at construction with float libm calls
// (fbank.cc:103-163 mel triangles, fbank.cc:249-256 Hamming window,
// srfft.cc:45-93 split-radix coefficient tables, srfft.cc:384-394 the iterated
// real-FFT twiddle).  They do not depend on the audio, so we build them on the
// host with the same arithmetic (host file compiled -O2 -ffp-contract=off, no
// -march) and upload them; the kernels only do the per-frame arithmetic.
#ifndef PK_TABLES_H_
#define PK_TABLES_H_

#include <stdint.h>

namespace pkmi {

constexpr int kSampleRate = 16000;
constexpr int kFrameLength = 400;      // fbank.cc:15  (25 ms)
constexpr int kFrameShift = 160;       // fbank.cc:14  (10 ms)
constexpr int kFftSize = 512;          // fbank.cc:259 round-up to power of two
constexpr int kFftCplx = 256;          // complex points of the half-size FFT
constexpr int kLogCplx = 8;
constexpr int kNumBins = 40;           // fbank.h:10
constexpr int kCmvnWindow = 600;       // cmvn.h:10
constexpr int kCmvnGlobalFrames = 200; // cmvn.h:11

// Split-radix schedule: the sub-transforms of size 2^logm, logm = 8..1, in pass
// order.  Pass p handles logm = 8 - p.  blk_off lists the block offsets of each
// pass back to back; pass_start[p] indexes into it.
constexpr int kNumPasses = 8;
constexpr int kMaxBlocks = 171;        // 1+1+3+5+11+21+43+85 (+1 spare)

// Twiddle tables: for logm = 4..8 six arrays of m/4 entries indexed by n
// (entry 0 and entry m/8 unused).  Offsets into tw[] per logm.
constexpr int kTwFloats = 6 * (4 + 8 + 16 + 32 + 64);

constexpr int kMelMaxLen = 64;         // longest triangle (checked at build time)
constexpr int kMelPacked = 640;        // all triangles back to back (sum of lengths, checked)
// The mel stage as the kernel runs it: every tap's product w[j] * P[off + j] (fbank.cc:165-184 -> vector.cc:252-262)
// is formed once, by any lane -- a product is one rounding wherever it is computed -- and only the additions keep the
// reference's order, one lane per bin.  Products are laid out bin after bin, each bin's run padded to a multiple of
// four with zero-weight taps (e + (+0) = e), so that a group of four products never straddles two bins: 9 per lane.
constexpr int kMelProducts = 576;      // >= sum of the padded lengths (checked); 64 lanes x (4 + 4 + 1)
constexpr int kMelMaxPadded = 32;      // longest padded run (checked): the kernel adds it in eight straight-line groups

// What the front-end kernels keep in LDS, in the layout they keep it in: built once on the host
// and copied flat, 16 bytes per lane (ten separate table copies, each a round trip to L2, were a
// third of the fbank kernel's time on a single utterance).
struct alignas(16) FrontendLdsImage {
  double logf_tab[32];
  float window[kFrameLength];
  float tw[kTwFloats];
  float post_re[kFftCplx / 2 + 1];
  float post_im[kFftCplx / 2 + 1];
  float mel_wprod[kMelProducts];       // tap weights in product order (padding taps: 0)
  short mel_pgrp[kMelProducts / 4];    // FFT bin of the first tap of product group q (products 4 q .. 4 q + 3: consecutive bins)
  short mel_pbase[kNumBins];           // first product of bin b (a multiple of four)
  short mel_plen[kNumBins];            // its padded length
  short tw_off[kLogCplx + 1];
  short pass_start[kNumPasses + 1];
  unsigned char blk_off[kMaxBlocks + 1];
  unsigned char bitrev[kFftCplx];
};
static_assert(sizeof(FrontendLdsImage) % 16 == 0, "copied in 16-byte pieces");

struct FrontendTables {
  double logf_tab[32];                 // pk_logf.h: 16 x (1/c, log c) of the C library's logf
  float window[kFrameLength];
  // FFT
  int32_t pass_start[kNumPasses + 1];
  int32_t blk_off[kMaxBlocks + 1];
  int32_t tw_off[kLogCplx + 1];        // tw_off[logm] -> start of the 6*(m/4) block
  float tw[kTwFloats];
  int32_t bitrev[kFftCplx];
  float post_re[kFftCplx / 2 + 1];
  float post_im[kFftCplx / 2 + 1];
  // mel: per bin (first FFT bin, length) and the weights of all bins back to back
  int32_t mel_off[kNumBins];
  int32_t mel_len[kNumBins];
  int32_t mel_maxlen;
  int32_t mel_base[kNumBins];          // start of bin b's weights in mel_packed
  float mel_packed[kMelPacked];
  FrontendLdsImage lds;                // the same tables, narrowed and packed for the kernels
};

// Returns 0 on success.  Pure host code, no HIP.
int BuildFrontendTables(FrontendTables *t);

// CMVN per-frame scalars.  The window count after frame t is min(t + 1, 600) exactly, so
// the smoothing weight (cmvn.cc:73-92) and the 1/count scale (cmvn.cc:94-101) -- both
// formed in double and narrowed to float in the reference -- depend on t only.  They are
// tabulated on the host with the reference's arithmetic: alpha[t] (0 once the window is
// full) and neg_scale[t] = -float(1 / double(count_t + alpha[t] * global_count)).
struct CmvnTables {
  float alpha[kCmvnWindow];
  float neg_scale[kCmvnWindow];
};
void BuildCmvnTables(float global_count, CmvnTables *t);

}  // namespace pkmi

#endif  // PK_TABLES_H_
