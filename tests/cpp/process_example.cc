// process_example.cc -- SURVEY 8(f-1): pk_process (pocketkaldi.cc:176-248) on top of libpk_mi355.so,
// with the reference's REAL decoder consuming the decodable.
//
//   process_example <model.conf> <utterance.wav> <graph.fst> [--reference-softmax]
//
// Stage for stage what pk_process does:
//   pk_read_audio       -> pk_mi355_16kpcm_read                         (pocketkaldi.cc:166-174)
//   empty utterance     -> empty hyp                                    (:180-184)
//   Fbank / CMVN / NNET -> pk_mi355_process_acoustic(verbose = 1): one call, device-resident, prints
//                          the reference's "Fbank:" / "CMVN:" / "NNET:" stage lines (:189-218)
//   decoder.Decode(&decodable); decoder.BestPath()                      (:220-222)
//                       -> pkref_decode(): the reference's decoder.cc / fst.cc compiled UNMODIFIED
//                          against include/reference_binding/decodable.h (oracle/_ref/libpkref_decoder.so,
//                          test infrastructure built by oracle/Makefile; no decoder is written here)
//   hyp words in spoken order, loglikelihood_per_frame = weight / T     (:225-239)
//   pk_decodable_destroy                                                (:247)
// Before the decode the example polls the decodable the way Decoder::Decode does (decoder.cc:49,
// 252-255): islastframe(-1) first, then one loglikelihood per frame.
// Built and run by tests/test_gpu_decoder.py; `--link-only` exits before touching the GPU.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pk_mi355.h"

extern "C" int pkref_decode(const char *fst_path, pk_decodable_t *decodable, int *words, int max_words,
                            float *weight, int *decode_ok);

struct Recognizer {                 // the fields of pk_t this path uses (pocketkaldi.h:36-42)
  pk_mi355_am_t *am = nullptr;
  float cmvn41[41];
  pk_vector_t cmvn_global_stats;
  std::string fst;
};

struct Utterance {                  // pk_utterance_t (pocketkaldi.h:53-57)
  pk_vector_t raw_wave = {0, nullptr};
  std::string hyp;
  float loglikelihood_per_frame = 0.0f;
  int num_frames = 0;
  float weight = 0.0f;
};

static int Process(Recognizer *recognizer, Utterance *utt) {
  if (utt->raw_wave.dim == 0) {     // pocketkaldi.cc:180-184
    utt->hyp = "";
    return 0;
  }
  pk_decodable_t decodable;
  if (pk_mi355_process_acoustic(recognizer->am, &recognizer->cmvn_global_stats, &utt->raw_wave, 0.1f,
                                &decodable, /*verbose=*/1) != 0) {
    fprintf(stderr, "process_acoustic: %s\n", pk_mi355_last_error());
    return 1;
  }
  const int T = decodable.log_prob.ncol;
  utt->num_frames = T;

  // the decoder's access pattern (decoder.cc:49,252-255): never the last frame before frame 0
  if (pk_decodable_islastframe(&decodable, -1)) return 2;
  double poll = 0.0;
  for (int frame = 0; !pk_decodable_islastframe(&decodable, frame - 1); ++frame)
    poll += pk_decodable_loglikelihood(&decodable, frame, 1 + frame % 36);
  if (!isfinite(poll)) return 3;

  int words[256], ok = 0;
  float weight = 0.0f;
  const int n = pkref_decode(recognizer->fst.c_str(), &decodable, words, 256, &weight, &ok);
  if (n < 0) {
    fprintf(stderr, "cannot read %s\n", recognizer->fst.c_str());
    return 4;
  }
  for (int i = 0; i < n && i < 256; ++i) utt->hyp += "w" + std::to_string(words[i]) + " ";
  utt->weight = weight;
  if (n > 0) utt->loglikelihood_per_frame = weight / T;      // pocketkaldi.cc:239
  pk_decodable_destroy(&decodable);                          // pocketkaldi.cc:247
  return ok ? 0 : 5;
}

int main(int argc, char **argv) {
  if (argc >= 2 && strcmp(argv[1], "--link-only") == 0) {
    printf("%s\n", pk_mi355_version());
    return 0;
  }
  if (argc < 4) {
    fprintf(stderr, "usage: %s model.conf utterance.wav graph.fst [--reference-softmax]\n", argv[0]);
    return 64;
  }
  Recognizer rec;
  if (pk_mi355_load(argv[1], PK_MI355_PRECISION_F32, &rec.am, rec.cmvn41) != 0) {   // pk_load, :72-144
    fprintf(stderr, "pk_mi355_load: %s\n", pk_mi355_last_error());
    return 1;
  }
  if (argc >= 5 && strcmp(argv[4], "--reference-softmax") == 0)
    pk_mi355_am_set_softmax(rec.am, PK_MI355_SOFTMAX_REFERENCE);
  rec.cmvn_global_stats.dim = 41;
  rec.cmvn_global_stats.data = rec.cmvn41;
  rec.fst = argv[3];

  Utterance utt;
  if (pk_mi355_16kpcm_read(argv[2], &utt.raw_wave) != 0) {                          // pk_read_audio
    fprintf(stderr, "pk_mi355_16kpcm_read: %s\n", pk_mi355_last_error());
    return 1;
  }
  const int rc = Process(&rec, &utt);
  printf("frames: %d\nhyp: %s\nweight: %.9g\nloglikelihood_per_frame: %.9g\n", utt.num_frames, utt.hyp.c_str(),
         utt.weight, utt.loglikelihood_per_frame);
  free(utt.raw_wave.data);
  pk_mi355_am_destroy(rec.am);
  if (rc == 0) printf("process_example ok\n");
  return rc;
}
