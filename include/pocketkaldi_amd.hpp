// pocketkaldi_amd.hpp -- header-only C++ host mirror of the reference's classes for the
// acoustic-scoring path, over the C ABI of libpk_mi355.so (include/pk_mi355.h).
//
// Same names, argument meaning and ownership rules as the reference, so code written against
// pocketkaldi's headers reads the same:
//
//   pocketkaldi::Fbank::Compute(const pk_vector_t*, pk_matrix_t*)        src/fbank.h:47-53
//   pocketkaldi::CMVN(global_stats, raw).GetFrame(t, pk_vector_t*)       src/cmvn.h:17-26
//   pocketkaldi::AcousticModel::Read / num_pdfs / TransitionIdToPdfId    src/am.h:23-52
//   pocketkaldi::AcousticModel::Compute(frames, loglikelihood)           src/am.h:35
//   pk_decodable_init/_destroy/_loglikelihood/_islastframe               src/decodable.h:20-41
//
// Error behaviour: the reference reports load errors through Status and treats misuse as
// assert(); here load / device errors surface as pocketkaldi::Status (ok() / what()), and the
// void compute members throw nothing -- they leave the output empty and set Status, readable
// through last_status().  No exception crosses the C ABI.
#ifndef POCKETKALDI_AMD_HPP_
#define POCKETKALDI_AMD_HPP_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "pk_mi355.h"

namespace pocketkaldi {

// src/status.h:37-100, reduced to what this path needs.
class Status {
 public:
  Status() : code_(0) {}
  Status(int code, const std::string &msg) : code_(code), msg_(msg) {}
  static Status OK() { return Status(); }
  static Status FromLast(int code) { return code == 0 ? Status() : Status(code, pk_mi355_last_error()); }
  bool ok() const { return code_ == 0; }
  int code() const { return code_; }
  const std::string &what() const { return msg_; }

 private:
  int code_;
  std::string msg_;
};

// src/fbank.h:47-53
class Fbank {
 public:
  // wave: 16 kHz mono sample values as pk_16kpcm_read produces them; fbank_feature is
  // resized (realloc) to {nrow = 40, ncol = T} like src/fbank.cc:267-276.
  void Compute(const pk_vector_t *wave, pk_matrix_t *fbank_feature) {
    status_ = Status::FromLast(pk_mi355_fbank_compute(wave, fbank_feature));
  }
  const Status &last_status() const { return status_; }

 private:
  Status status_;
};

// src/cmvn.h:17-26.  The reference computes frame by frame and asserts sequential access
// (src/cmvn.cc:38); here all frames are produced on the first GetFrame() and served from a
// host copy, so any access order works.
class CMVN {
 public:
  CMVN(const pk_vector_t *global_stats, const pk_matrix_t *raw_feats)
      : global_stats_(global_stats), raw_(raw_feats), done_(false) {
    out_.ncol = out_.nrow = 0;
    out_.data = nullptr;
  }
  ~CMVN() { free(out_.data); }
  CMVN(const CMVN &) = delete;
  CMVN &operator=(const CMVN &) = delete;

  void GetFrame(int frame, pk_vector_t *feats) {
    if (!done_) {
      status_ = Status::FromLast(pk_mi355_cmvn_apply(global_stats_, raw_, &out_));
      done_ = true;
    }
    if (!status_.ok() || frame < 0 || frame >= out_.ncol) return;
    if (feats->dim != out_.nrow) {   // pk_vector_copy resizes (src/vector.cc)
      feats->data = static_cast<float *>(realloc(feats->data, sizeof(float) * out_.nrow));
      feats->dim = out_.nrow;
    }
    memcpy(feats->data, out_.data + static_cast<size_t>(frame) * out_.nrow, sizeof(float) * out_.nrow);
  }
  // all frames at once: {nrow = 40, ncol = T}, owned by this object
  const pk_matrix_t *AllFrames() {
    pk_vector_t dummy = {0, nullptr};
    GetFrame(-1, &dummy);
    return &out_;
  }
  const Status &last_status() const { return status_; }

 private:
  const pk_vector_t *global_stats_;
  const pk_matrix_t *raw_;
  pk_matrix_t out_;
  bool done_;
  Status status_;
};

// src/am.h:23-52 (+ the Nnet it owns, src/nnet.h:88-104)
class AcousticModel {
 public:
  AcousticModel() : am_(pk_mi355_am_create()) {}
  ~AcousticModel() { pk_mi355_am_destroy(am_); }
  AcousticModel(const AcousticModel &) = delete;
  AcousticModel &operator=(const AcousticModel &) = delete;

  // AcousticModel::Read (src/am.cc:23-63) with the Configuration already resolved to paths
  // and integers (keys nnet, prior, left_context, right_context, num_pdfs, tid2pdf).
  Status Read(const std::string &nnet, const std::string &prior, const std::string &tid2pdf,
              int left_context, int right_context, int num_pdfs) {
    return Status::FromLast(pk_mi355_am_read(am_, nnet.c_str(), prior.c_str(),
                                             tid2pdf.empty() ? nullptr : tid2pdf.c_str(),
                                             left_context, right_context, num_pdfs));
  }
  // In-memory construction (what Nnet::ReadLayer does per layer, src/nnet.cc:80-130)
  Status AddLinear(int in_dim, int out_dim, const float *W, const float *b) {
    return Status::FromLast(pk_mi355_am_add_linear(am_, in_dim, out_dim, W, b));
  }
  Status AddLayer(int layer_type) { return Status::FromLast(pk_mi355_am_add_layer(am_, layer_type)); }
  Status Finalize(const std::vector<float> &prior, int left_context, int right_context,
                  const std::vector<int32_t> &tid2pdf) {
    return Status::FromLast(pk_mi355_am_finalize(am_, prior.data(), static_cast<int>(prior.size()),
                                                 left_context, right_context,
                                                 tid2pdf.empty() ? nullptr : tid2pdf.data(),
                                                 static_cast<int>(tid2pdf.size())));
  }

  int TransitionIdToPdfId(int transition_id) const { return pk_mi355_am_transition_to_pdf(am_, transition_id); }
  int num_pdfs() const { return pk_mi355_am_num_pdfs(am_); }

  // src/am.cc:90-115: frames {nrow = feat_dim, ncol = T} -> loglikelihood {nrow = num_pdfs,
  // ncol = T} = log(max(p, 1e-20)) - log prior  (no acoustic scale; pk_decodable_init applies it)
  void Compute(const pk_matrix_t *frames, pk_matrix_t *loglikelihood) {
    pk_decodable_t d;
    pk_decodable_init(&d, am_, 1.0f, frames);
    free(loglikelihood->data);
    *loglikelihood = d.log_prob;   // ownership moves to the caller, as with pk_matrix_t in the reference
  }

  pk_mi355_am_t *handle() const { return am_; }

 private:
  pk_mi355_am_t *am_;
};

// Utterance sharding for N GPUs (one process per GPU, a full weight replica each, no data-path collective): which rank
// scores which utterance of a list.  The reference scores a ragged list one WAV after another (src/main.cc:34-46, every
// length from src/fbank.cc:35-42); balancing by FRAMES -- longest first, each utterance to the rank with the fewest
// frames so far, ties to the lower rank -- keeps the slowest rank within a fraction of a percent of the mean where
// `u mod N` leaves several percent.  A pure function of the frame counts: every rank computes the same map without
// talking.  (The Python twin is pocketkaldi_amd.dist.partition_by_frames.)
inline std::vector<std::vector<int> > PartitionByFrames(const std::vector<int> &frames, int world) {
  std::vector<int> order(frames.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frames[a] > frames[b]; });
  std::vector<long long> load(world > 0 ? world : 0, 0);
  std::vector<std::vector<int> > shards(load.size());
  for (size_t i = 0; i < order.size() && !load.empty(); ++i) {
    const int r = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());   // (first minimum = lower rank)
    shards[r].push_back(order[i]);
    load[r] += frames[order[i]];
  }
  for (size_t r = 0; r < shards.size(); ++r) std::sort(shards[r].begin(), shards[r].end());
  return shards;
}

}  // namespace pocketkaldi

#endif  // POCKETKALDI_AMD_HPP_
