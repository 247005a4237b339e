// ref_decoder_shim.cc -- extern "C" entry over the reference's REAL decoder, compiled against
// the drop-in binding instead of the reference's own decodable.h.
//
// TEST INFRASTRUCTURE ONLY (integration proof for SURVEY 8(b) / 8(f-1)).  oracle/Makefile compiles
//   /root/reference/src/decoder.cc fst.cc util.cc hashtable.cc strlcpy.cc
// from where they lie, UNMODIFIED, in a scratch directory of symbolic links in which
// include/reference_binding/decodable.h stands where src/decodable.h stood --
// that substitution is the documented binding, not a stand-in for anything the image lacks.
// Output: oracle/_ref/libpkref_decoder.so, whose only undefined pk_* symbols are
// pk_decodable_islastframe / pk_decodable_loglikelihood (decoder.cc:49,252,276): at run time they
// bind to libpk_mi355.so.  Nothing of the reference is copied into this repository.
//
// What it gives the tests: Decoder::Decode (decoder.cc:39-66) + BestPath consuming a
// pk_decodable_t made on the GPU, i.e. the second half of pk_process (pocketkaldi.cc:208-240).
#include "decoder.h"
#include "fst.h"
#include "util.h"

#include <algorithm>
#include <vector>

using pocketkaldi::Decoder;
using pocketkaldi::Fst;

extern "C" {

// pocketkaldi.cc:208-240 without the symbol table: decode, best path, words in spoken order
// (pk_process reverses hyp.words(), :226-227), weight = hyp.weight() (the numerator of
// loglikelihood_per_frame, :239).  Returns the number of words (<= max_words written), -1 when
// the FST cannot be read.
int pkref_decode(const char *fst_path, pk_decodable_t *decodable, int *words, int max_words,
                 float *weight, int *decode_ok) {
  pocketkaldi::util::ReadableFile fd;
  pocketkaldi::Status status = fd.Open(fst_path);
  if (!status.ok()) return -1;
  Fst fst;
  status = fst.Read(&fd);
  if (!status.ok()) return -1;

  Decoder decoder(&fst);
  // (Decoder::ReachedFinal is declared, decoder.h:64, but defined nowhere in the reference)
  const bool ok = decoder.Decode(decodable);
  if (decode_ok) *decode_ok = ok ? 1 : 0;
  Decoder::Hypothesis hyp = decoder.BestPath();
  std::vector<int> w = hyp.words();
  std::reverse(w.begin(), w.end());
  for (int i = 0; i < (int)w.size() && i < max_words; ++i) words[i] = w[i];
  if (weight) *weight = hyp.weight();
  return (int)w.size();
}

// sizeof / offsets as the reference-side translation units see them (ABI check from the tests)
int pkref_sizeof_decodable(void) { return (int)sizeof(pk_decodable_t); }
int pkref_offsetof_decodable_am(void) { return (int)offsetof(pk_decodable_t, am); }

}  // extern "C"
