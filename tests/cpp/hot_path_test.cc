// hot_path_test.cc -- the reference's own test programs for this path (test/fbank_test.cc,
// test/cmvn_test.cc, test/nnet_test.cc), re-expressed against include/pocketkaldi_amd.hpp:
// a C++ caller written the way the reference's callers are, linked to libpk_mi355.so.
// Built and run by tests/test_cpp_host.py (needs a GPU); `--link-only` exits before any compute.
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "pocketkaldi_amd.hpp"

using pocketkaldi::AcousticModel;
using pocketkaldi::CMVN;
using pocketkaldi::Fbank;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d) last error: %s\n", #cond, __FILE__, __LINE__, pk_mi355_last_error()); exit(1); } } while (0)

static std::vector<float> ReadArray(const std::string &path) {
  std::ifstream is(path);
  std::vector<float> v;
  double x;
  while (is >> x) v.push_back(static_cast<float>(x));
  return v;
}

// 44-byte-header 16-bit PCM, as pk_16kpcm_read accepts (src/pcm_reader.cc:45-220)
static std::vector<float> ReadWav16(const std::string &path) {
  std::ifstream is(path, std::ios::binary);
  std::vector<char> bytes((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
  CHECK(bytes.size() > 44 && memcmp(bytes.data(), "RIFF", 4) == 0);
  std::vector<float> s((bytes.size() - 44) / 2);
  for (size_t i = 0; i < s.size(); ++i) {
    int16_t v;
    memcpy(&v, &bytes[44 + 2 * i], 2);
    s[i] = v;
  }
  return s;
}

static std::vector<float> ReadVec0(const std::string &path) {
  std::ifstream is(path, std::ios::binary);
  char head[12];
  is.read(head, 12);
  CHECK(memcmp(head, "VEC0", 4) == 0);
  int32_t n;
  memcpy(&n, head + 8, 4);
  std::vector<float> v(n);
  is.read(reinterpret_cast<char *>(v.data()), 4 * n);
  return v;
}

static void TestFbankAndCmvn(const std::string &dir) {      // test/fbank_test.cc:15-56, test/cmvn_test.cc:33-82
  std::vector<float> pcm = ReadWav16(dir + "en-us-hello.wav");
  CHECK(pcm.size() == 7802);
  pk_vector_t pcm_data = {static_cast<int>(pcm.size()), pcm.data()};
  Fbank fbank;
  pk_matrix_t fbank_feat = {0, 0, nullptr};
  fbank.Compute(&pcm_data, &fbank_feat);
  CHECK(fbank.last_status().ok());
  CHECK(fbank_feat.ncol == 47 && fbank_feat.nrow == 40);
  std::vector<float> corr = ReadArray(dir + "fbankmat_en-us-hello.wav.txt");
  CHECK(corr.size() == 47u * 40u);
  for (size_t i = 0; i < corr.size(); ++i) CHECK(fabs(fbank_feat.data[i] - corr[i]) < 3e-5);

  std::vector<float> stats = ReadVec0(dir + "cmvn_stats.bin");
  CHECK(stats.size() == 41);
  pk_vector_t global_stats = {41, stats.data()};
  CMVN cmvn(&global_stats, &fbank_feat);
  std::vector<float> corr2 = ReadArray(dir + "fbankcmvnmat_en-us-hello.wav.txt");
  pk_vector_t feats = {0, nullptr};
  for (int i = 0; i < fbank_feat.ncol; ++i) {
    cmvn.GetFrame(i, &feats);
    CHECK(cmvn.last_status().ok() && feats.dim == 40);
    for (int d = 0; d < feats.dim; ++d) CHECK(fabs(feats.data[d] - corr2[i * feats.dim + d]) < 3e-5);
  }
  free(feats.data);
  free(fbank_feat.data);
}

static void TestLinearLayerThroughDecodable() {              // test/nnet_test.cc:23-55 numbers
  float W[] = {0.1f, 0.8f, 0.9f, 0.4f, 0.2f, 0.7f, 0.2f, 0.1f, 0.1f, 0.4f, 0.3f, 0.2f};
  float b[] = {0.1f, -0.1f, 0.2f, -0.2f};
  AcousticModel am;
  CHECK(am.AddLinear(3, 4, W, b).ok());
  std::vector<float> prior(4, 1.0f);          // log prior = 0
  std::vector<int32_t> tid2pdf = {0, 3, 2, 1, 0};
  CHECK(am.Finalize(prior, 0, 0, tid2pdf).ok());
  CHECK(am.num_pdfs() == 4 && am.TransitionIdToPdfId(1) == 3);
  float x[] = {0.3f, -0.1f, 0.9f};
  pk_matrix_t feats = {1, 3, x};
  pk_decodable_t d;
  pk_decodable_init(&d, am.handle(), 1.0f, &feats);
  CHECK(d.log_prob.ncol == 1 && d.log_prob.nrow == 4);
  const float y[] = {0.86f, 0.63f, 0.34f, 0.07f};           // expected layer outputs
  for (int i = 0; i < 4; ++i) CHECK(fabs(d.log_prob.data[i] - logf(y[i])) < 1e-5);
  CHECK(!pk_decodable_islastframe(&d, -1));                 // first poll of Decoder::Decode
  CHECK(pk_decodable_islastframe(&d, 0));
  CHECK(pk_decodable_loglikelihood(&d, 0, 1) == d.log_prob.data[3]);
  pk_decodable_destroy(&d);
  CHECK(d.log_prob.data == nullptr);
}

// The throughput recipe of INTEGRATION.md section 2: two batches alternating, int16 PCM in
// page-locked memory, results fetched as views of the batch's page-locked arena.
static void TestPipelinedBatches(const std::string &dir) {
  std::vector<float> wav = ReadWav16(dir + "en-us-hello.wav");
  std::vector<float> stats = ReadVec0(dir + "cmvn_stats.bin");
  CHECK(wav.size() == 7802 && stats.size() == 41);
  // 40 -> 8 softmax scorer, no context
  std::vector<float> W(8 * 40), b(8, 0.0f), prior(8, 0.125f);
  for (size_t i = 0; i < W.size(); ++i) W[i] = 0.01f * (float)((int)(i * 7919 % 31) - 15);
  AcousticModel am;
  CHECK(am.AddLinear(40, 8, W.data(), b.data()).ok());
  CHECK(am.AddLayer(PK_NNET_SOFTMAX_LAYER).ok());
  CHECK(am.Finalize(prior, 0, 0, {}).ok());

  const int kUtts = 3;
  const int ns[kUtts] = {7802, 4000, 399};             // the last one is shorter than a frame
  int16_t *pcm = static_cast<int16_t *>(pk_mi355_host_malloc(sizeof(int16_t) * (7802 + 4000 + 399)));
  CHECK(pcm != nullptr);
  int64_t off = 0;
  for (int u = 0; u < kUtts; ++u)
    for (int i = 0; i < ns[u]; ++i) pcm[off++] = (int16_t)wav[i];

  pk_mi355_batch_t *batch[2];
  for (int k = 0; k < 2; ++k) {
    batch[k] = pk_mi355_batch_create(am.handle(), stats.data(), kUtts, off);
    CHECK(batch[k] != nullptr);
  }
  pk_decodable_t views[2][kUtts];
  for (int round = 0; round < 3; ++round) {
    for (int k = 0; k < 2; ++k) {                      // queue both: k = 1 scores while k = 0 copies
      CHECK(pk_mi355_batch_set_waves_i16(batch[k], pcm, ns, kUtts) == 0);
      CHECK(pk_mi355_batch_score(batch[k], 0.1f, /*sync=*/0) == 0);
      CHECK(pk_mi355_batch_fetch_all(batch[k], views[k], kUtts, /*sync=*/0) == 0);
    }
    for (int k = 0; k < 2; ++k) {
      CHECK(pk_mi355_batch_synchronize(batch[k]) == 0);
      for (int u = 0; u < kUtts; ++u) {
        pk_decodable_t owned;                          // the malloc'd copy of the same utterance
        CHECK(pk_mi355_batch_fetch(batch[k], u, &owned) == 0);
        const int T = pk_mi355_batch_num_frames(batch[k], u);
        CHECK(views[k][u].log_prob.ncol == T && owned.log_prob.ncol == T);
        if (T > 0) {
          CHECK(views[k][u].log_prob.nrow == 8);
          CHECK(memcmp(views[k][u].log_prob.data, owned.log_prob.data, sizeof(float) * 8 * T) == 0);
          CHECK(pk_decodable_islastframe(&views[k][u], T - 1));
          CHECK(pk_decodable_loglikelihood(&views[k][u], 0, 3) == views[k][u].log_prob.data[3]);
        }
        pk_decodable_destroy(&owned);
        pk_decodable_destroy(&views[k][u]);            // a view: frees nothing
      }
    }
  }
  CHECK(pk_mi355_batch_num_frames(batch[0], 0) == 47 && pk_mi355_batch_num_frames(batch[0], 2) == 0);
  for (int k = 0; k < 2; ++k) pk_mi355_batch_destroy(batch[k]);
  pk_mi355_host_free(pcm);
}

// Host logic that needs no device: frame counts (fbank.cc:35-42) and the length-balanced sharding of a ragged list.
static void TestHostLogic() {
  CHECK(pk_mi355_num_frames(399) == 0 && pk_mi355_num_frames(400) == 1 && pk_mi355_num_frames(160000) == 998);
  std::vector<int> frames;
  unsigned x = 12345;
  long long total = 0;
  for (int u = 0; u < 2048; ++u) {                   // 2 s .. 20 s
    x = x * 1664525u + 1013904223u;
    frames.push_back(pk_mi355_num_frames(32000 + (int)(x % 288001u)));
    total += frames.back();
  }
  const std::vector<std::vector<int> > shards = pocketkaldi::PartitionByFrames(frames, 8);
  CHECK(shards.size() == 8);
  std::vector<char> seen(frames.size(), 0);
  long long worst = 0;
  for (size_t r = 0; r < shards.size(); ++r) {
    long long load = 0;
    for (size_t i = 0; i < shards[r].size(); ++i) {
      CHECK(!seen[shards[r][i]]);
      CHECK(i == 0 || shards[r][i - 1] < shards[r][i]);
      seen[shards[r][i]] = 1;
      load += frames[shards[r][i]];
    }
    if (load > worst) worst = load;
  }
  for (size_t u = 0; u < seen.size(); ++u) CHECK(seen[u]);
  CHECK(worst * 8.0 / total - 1.0 < 0.02);             // slowest rank within 2 % of the mean (measured: 1e-5)
  CHECK(pocketkaldi::PartitionByFrames(std::vector<int>(), 4).size() == 4);
  CHECK(pocketkaldi::PartitionByFrames(std::vector<int>(1, 5), 2)[0] == std::vector<int>(1, 0));
}

int main(int argc, char **argv) {
  if (argc > 1 && strcmp(argv[1], "--link-only") == 0) {
    printf("%s\n", pk_mi355_version());
    TestHostLogic();
    printf("host logic ok\n");
    return 0;
  }
  std::string dir = argc > 1 ? argv[1] : "tests/golden/";
  if (dir.back() != '/') dir += '/';
  TestFbankAndCmvn(dir);
  TestLinearLayerThroughDecodable();
  TestPipelinedBatches(dir);
  printf("hot_path_test ok\n");
  return 0;
}
