// broadcast_test.cc -- the multi-GPU form of pk_load (src/pocketkaldi.cc:72-144) as a C++ host
// writes it: every rank builds / reads the model, then ONE
// pk_mi355_am_broadcast over its RCCL communicator replaces its weights by rank `root`'s.
// Linked against the system RCCL and libpk_mi355.so; run by tests/test_cpp_host.py (-m gpu).
// One GPU here, so nranks = 1: the call path (run-time RCCL binding, in-place ncclBroadcast of
// the device blob, caller stream / own stream) is the one N ranks take; the out-of-place form
// (pk_mi355_am_broadcast_from) shows RCCL writing ANOTHER model's blob.  RCCL has not run this
// entry with more than one rank yet (no multi-GPU box in the build loop).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "pk_mi355.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d) last error: %s\n", #cond, __FILE__, __LINE__, pk_mi355_last_error()); exit(1); } } while (0)

static pk_mi355_am_t *BuildModel(bool real_values) {
  const int D = 40, H = 32, N = 24;
  std::vector<float> W1(H * D * 3), b1(H), W2(N * H), b2(N), prior(N, 1.0f / N);
  for (size_t i = 0; i < W1.size(); ++i) W1[i] = real_values ? 0.02f * (float)((int)(i * 7919 % 41) - 20) : 0.0f;
  for (size_t i = 0; i < W2.size(); ++i) W2[i] = real_values ? 0.05f * (float)((int)(i * 104729 % 23) - 11) : 0.0f;
  for (int i = 0; i < H; ++i) b1[i] = real_values ? 0.01f * i : 0.0f;
  for (int i = 0; i < N; ++i) b2[i] = real_values ? -0.02f * i : 0.0f;
  pk_mi355_am_t *am = pk_mi355_am_create();
  CHECK(am != nullptr);
  CHECK(pk_mi355_am_add_linear(am, D * 3, H, W1.data(), b1.data()) == 0);
  CHECK(pk_mi355_am_add_layer(am, PK_NNET_RELU_LAYER) == 0);
  CHECK(pk_mi355_am_add_linear(am, H, N, W2.data(), b2.data()) == 0);
  CHECK(pk_mi355_am_add_layer(am, PK_NNET_SOFTMAX_LAYER) == 0);
  CHECK(pk_mi355_am_finalize(am, prior.data(), N, 1, 1, nullptr, 0) == 0);
  return am;
}

static std::vector<float> Score(pk_mi355_am_t *am) {
  const int T = 50, D = 40;
  std::vector<float> x(T * D);
  for (size_t i = 0; i < x.size(); ++i) x[i] = 0.1f * (float)((int)(i * 31 % 17) - 8);
  pk_matrix_t feats = {T, D, x.data()};
  pk_decodable_t d;
  pk_decodable_init(&d, am, 0.1f, &feats);
  CHECK(d.log_prob.ncol == T);
  std::vector<float> out(d.log_prob.data, d.log_prob.data + (size_t)T * d.log_prob.nrow);
  pk_decodable_destroy(&d);
  return out;
}

int main() {
  CHECK(pk_mi355_set_device(0) == 0);
  CHECK(hipSetDevice(0) == hipSuccess);
  pk_mi355_am_t *am = BuildModel(true);
  const std::vector<float> want = Score(am);

  ncclUniqueId id;
  CHECK(ncclGetUniqueId(&id) == ncclSuccess);        // rank 0 makes it; a real job hands it to the other ranks
  ncclComm_t comm;
  CHECK(ncclCommInitRank(&comm, /*nranks=*/1, id, /*rank=*/0) == ncclSuccess);

  // synchronous form: NULL stream
  CHECK(pk_mi355_am_broadcast(am, comm, /*root=*/0, nullptr) == 0);
  CHECK(Score(am) == want);

  // asynchronous form on the caller's stream
  hipStream_t s;
  CHECK(hipStreamCreate(&s) == hipSuccess);
  CHECK(pk_mi355_am_broadcast(am, comm, 0, s) == 0);
  CHECK(hipStreamSynchronize(s) == hipSuccess);
  CHECK(Score(am) == want);

  // the blob really is what travels: overwrite it with another model's, broadcast is the identity
  // at one rank, so the scores must now be the OTHER model's (zeros -> uniform posteriors)
  pk_mi355_am_t *zero = BuildModel(false);
  CHECK(pk_mi355_am_blob_bytes(zero) == pk_mi355_am_blob_bytes(am));
  CHECK(hipMemcpy(pk_mi355_am_blob_device_ptr(am), pk_mi355_am_blob_device_ptr(zero), pk_mi355_am_blob_bytes(am),
                  hipMemcpyDeviceToDevice) == hipSuccess);
  CHECK(pk_mi355_am_broadcast(am, comm, 0, nullptr) == 0);
  CHECK(Score(am) == Score(zero) && Score(am) != want);

  // RCCL itself writes the destination model's weights: `fresh` shares nothing with `real` and
  // starts from a third pattern (all zeros); the out-of-place form sends real's blob and receives
  // into fresh's, which at one rank is RCCL copying blob to blob on its own stream
  pk_mi355_am_t *real = BuildModel(true);
  pk_mi355_am_t *fresh = BuildModel(false);
  CHECK(Score(fresh) != want);
  CHECK(pk_mi355_am_broadcast_from(fresh, real, comm, 0, nullptr) == 0);
  CHECK(Score(fresh) == want);
  CHECK(Score(real) == want);
  // ... and on a caller stream, back to zeros
  CHECK(pk_mi355_am_broadcast_from(fresh, zero, comm, 0, s) == 0);
  CHECK(hipStreamSynchronize(s) == hipSuccess);
  CHECK(Score(fresh) == Score(zero));
  pk_mi355_am_destroy(fresh);
  pk_mi355_am_destroy(real);

  // misuse is reported, not fatal
  CHECK(pk_mi355_am_broadcast(am, nullptr, 0, nullptr) != 0);
  CHECK(pk_mi355_am_broadcast(am, comm, -1, nullptr) != 0);

  CHECK(hipStreamDestroy(s) == hipSuccess);
  CHECK(ncclCommDestroy(comm) == ncclSuccess);
  pk_mi355_am_destroy(zero);
  pk_mi355_am_destroy(am);
  printf("broadcast_test ok\n");
  return 0;
}
