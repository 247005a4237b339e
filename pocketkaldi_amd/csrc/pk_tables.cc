// pk_tables.cc -- host construction of the front-end constant tables.
// Compile: g++ -O2 -ffp-contract=off (no -march): float expressions must round
// exactly where the reference's do.  See pk_tables.h for what each table is.
#include "pk_tables.h"
#include "pk_logf.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace pkmi {

namespace {

// srfft.cc:37-39 full-precision 2*pi; fbank.cc:18-20 truncated 2*pi.
const double kTwoPiFull = 6.283185307179586476925286766559005;
const double kTwoPiTrunc = 6.28318530718;

struct Block { int off, logm; };

// srfft.cc:224-237: a transform of 2^logm points at `off` recurses into
// (off, logm-1), (off + m/2, logm-2), (off + 3m/4, logm-2).
void Enumerate(int off, int logm, std::vector<Block> *out) {
  if (logm <= 0) return;
  out->push_back({off, logm});
  if (logm >= 2) {
    int m = 1 << logm;
    Enumerate(off, logm - 1, out);
    Enumerate(off + m / 2, logm - 2, out);
    Enumerate(off + 3 * (m / 4), logm - 2, out);
  }
}

float MelScale(float freq) {  // fbank.h:30-32
  return 1127.0f * logf(1.0f + freq / 700.0f);
}

}  // namespace

int BuildFrontendTables(FrontendTables *t) {
  memset(t, 0, sizeof(*t));
  static const double logf_tab[kLogfTableDoubles] = PK_LOGF_TABLE_INIT;
  memcpy(t->logf_tab, logf_tab, sizeof(logf_tab));

  // ---- Hamming window, fbank.cc:249-256 (float angle step, float cos, the
  // 0.54 - 0.46 * c expression in double, one rounding to float)
  float a = kTwoPiTrunc / (kFrameLength - 1);
  for (int i = 0; i < kFrameLength; ++i) {
    float i_fl = static_cast<float>(i);
    t->window[i] = 0.54 - 0.46 * cosf(a * i_fl);
  }

  // ---- split-radix block schedule
  std::vector<Block> blocks;
  Enumerate(0, kLogCplx, &blocks);
  std::stable_sort(blocks.begin(), blocks.end(), [](const Block &x, const Block &y) {
    return x.logm != y.logm ? x.logm > y.logm : x.off < y.off;
  });
  if (static_cast<int>(blocks.size()) > kMaxBlocks) return -1;
  int pos = 0;
  for (int p = 0; p < kNumPasses; ++p) {
    int logm = kLogCplx - p;
    t->pass_start[p] = pos;
    for (const Block &b : blocks)
      if (b.logm == logm) t->blk_off[pos++] = b.off;
  }
  t->pass_start[kNumPasses] = pos;

  // ---- butterfly coefficient tables, srfft.cc:64-91
  int woff = 0;
  for (int logm = 4; logm <= kLogCplx; ++logm) {
    int m = 1 << logm, m4 = m / 4, m8 = m / 8;
    t->tw_off[logm] = woff;
    float *base = t->tw + woff;
    for (int n = 1; n < m4; ++n) {
      if (n == m8) continue;
      float ang = n * kTwoPiFull / m;
      float c = cosf(ang), s = sinf(ang);
      base[0 * m4 + n] = c;
      base[1 * m4 + n] = -(s + c);
      base[2 * m4 + n] = s - c;
      ang = 3 * n * kTwoPiFull / m;
      c = cosf(ang);
      s = sinf(ang);
      base[3 * m4 + n] = c;
      base[4 * m4 + n] = -(s + c);
      base[5 * m4 + n] = s - c;
    }
    woff += 6 * m4;
  }
  if (woff != kTwFloats) return -2;

  // ---- bit reversal (what srfft.cc:239-265 realises)
  for (int i = 0; i < kFftCplx; ++i) {
    int r = 0, v = i;
    for (int bit = 0; bit < kLogCplx; ++bit) { r = (r << 1) | (v & 1); v >>= 1; }
    t->bitrev[i] = r;
  }

  // ---- real post-pass twiddle, srfft.cc:384-394: iterated float complex product
  float ang1 = static_cast<float>(kTwoPiFull / kFftSize * -1);
  float root_re = cosf(ang1), root_im = sinf(ang1);
  float kre = 1.0f, kim = 0.0f;
  t->post_re[0] = kre;
  t->post_im[0] = kim;
  for (int k = 1; k <= kFftCplx / 2; ++k) {
    float tmp = (kre * root_re) - (kim * root_im);
    kim = kre * root_im + kim * root_re;
    kre = tmp;
    t->post_re[k] = kre;
    t->post_im[k] = kim;
  }

  // ---- mel triangles, fbank.cc:103-163
  float sample_freq = kSampleRate;
  int num_fft_bins = kFftSize / 2;
  float fft_bin_width = sample_freq / kFftSize;
  float mel_low = MelScale(20);
  float mel_high = MelScale(kSampleRate / 2);
  float mel_delta = (mel_high - mel_low) / (kNumBins + 1);
  t->mel_maxlen = 0;
  int packed = 0;
  for (int bin = 0; bin < kNumBins; ++bin) {
    float left = mel_low + bin * mel_delta;
    float center = mel_low + (bin + 1) * mel_delta;
    float right = mel_low + (bin + 2) * mel_delta;
    int first = -1, last = -1;
    std::vector<float> w(num_fft_bins, 0.0f);
    for (int i = 0; i < num_fft_bins; ++i) {
      float freq = fft_bin_width * i;
      float mel = MelScale(freq);
      if (mel > left && mel < right) {
        w[i] = (mel <= center) ? (mel - left) / (center - left)
                               : (right - mel) / (right - center);
        if (first < 0) first = i;
        last = i;
      }
    }
    if (first < 0 || last <= first) return -3;
    int len = last + 1 - first;
    if (len > kMelMaxLen) return -4;
    t->mel_off[bin] = first;
    t->mel_len[bin] = len;
    t->mel_maxlen = std::max(t->mel_maxlen, len);
    if (packed + len > kMelPacked) return -5;
    t->mel_base[bin] = packed;
    for (int j = 0; j < len; ++j) t->mel_packed[packed + j] = w[first + j];
    packed += len;
  }

  // ---- the packed LDS image of the kernels
  FrontendLdsImage &L = t->lds;
  memcpy(L.logf_tab, t->logf_tab, sizeof(L.logf_tab));
  memcpy(L.window, t->window, sizeof(L.window));
  memcpy(L.tw, t->tw, sizeof(L.tw));
  memcpy(L.post_re, t->post_re, sizeof(L.post_re));
  memcpy(L.post_im, t->post_im, sizeof(L.post_im));
  // the mel stage in product order (pk_tables.h: kMelProducts)
  memset(L.mel_wprod, 0, sizeof(L.mel_wprod));
  memset(L.mel_pgrp, 0, sizeof(L.mel_pgrp));
  int prod = 0;
  for (int b = 0; b < kNumBins; ++b) {
    const int len = t->mel_len[b], padded = (len + 3) / 4 * 4;
    // (a padding tap reads a power-spectrum slot behind the triangle: it must exist -- pw[] has kFftCplx + 4 of them)
    if (prod + padded > kMelProducts || padded > kMelMaxPadded || t->mel_off[b] + padded > kFftCplx + 4) return -6;
    L.mel_pbase[b] = (short)prod;
    L.mel_plen[b] = (short)padded;
    for (int j = 0; j < len; ++j) L.mel_wprod[prod + j] = t->mel_packed[t->mel_base[b] + j];
    for (int j = 0; j < padded; j += 4) L.mel_pgrp[(prod + j) / 4] = (short)(t->mel_off[b] + j);
    prod += padded;
  }
  for (int i = 0; i <= kLogCplx; ++i) L.tw_off[i] = (short)t->tw_off[i];
  for (int i = 0; i <= kNumPasses; ++i) L.pass_start[i] = (short)t->pass_start[i];
  for (int i = 0; i <= kMaxBlocks; ++i) {
    if (t->blk_off[i] < 0 || t->blk_off[i] > 255) return -1;
    L.blk_off[i] = (unsigned char)t->blk_off[i];
  }
  for (int i = 0; i < kFftCplx; ++i) L.bitrev[i] = (unsigned char)t->bitrev[i];
  return 0;
}

void BuildCmvnTables(float global_count_f, CmvnTables *t) {
  const double global_count = global_count_f;
  for (int i = 0; i < kCmvnWindow; ++i) {
    float cnt = static_cast<float>(i + 1);
    float alpha = 0.0f;
    const double count = cnt;
    if (count < kCmvnWindow) {                       // cmvn.cc:80-91
      double from_global = kCmvnWindow - count;
      if (from_global > kCmvnGlobalFrames) from_global = kCmvnGlobalFrames;
      alpha = static_cast<float>(from_global / global_count);
      cnt += alpha * global_count_f;                 // the count element of the float axpy
    }
    const float scale = static_cast<float>(1 / static_cast<double>(cnt));   // cmvn.cc:99
    t->alpha[i] = alpha;
    t->neg_scale[i] = -scale;
  }
}

}  // namespace pkmi
