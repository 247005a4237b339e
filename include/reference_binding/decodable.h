/*
 * decodable.h -- the reference-side binding of libpk_mi355.so.
 *
 * This file TAKES THE PLACE of pocketkaldi's src/decodable.h (decodable.h:1-43): put it there
 * (with pk_mi355.h next to it or on the include path) and drop src/decodable.cc from the build.
 * decoder.h:31 includes "decodable.h" and decoder.cc:49,252,276 call
 * pk_decodable_islastframe / pk_decodable_loglikelihood: both compile UNCHANGED against this
 * file; pocketkaldi.cc compiles with the two edits of INTEGRATION.md section 2(b) (pk_load,
 * pk_destroy) -- pk_process (pocketkaldi.cc:176-248) and pocketkaldi.h are untouched.
 *
 * The reference's own matrix.h / vector.h come first, so pk_matrix_t / pk_vector_t are the
 * reference's declarations (pk_mi355.h skips its identical ones under their include guards).
 * The model handle keeps the type NAME the reference uses for it: decodable.h:15-27 says
 * `AcousticModel *am` in pk_decodable_t and in pk_decodable_init, and pk_t::am is a
 * `pocketkaldi::AcousticModel *` (pocketkaldi.h:38).  Here that pointer is the opaque handle
 * pk_mi355_am_create() / pk_mi355_load() return (weights in HBM); it must only be handed to
 * pk_mi355_* / pk_decodable_* functions -- never dereferenced, never `delete`d (that is edit 2(b)).
 *
 * tests/test_reference_binding.py compiles the reference's decoder.cc and (edited as documented)
 * pocketkaldi.cc against exactly this file; oracle/Makefile builds the reference's real decoder
 * against it (oracle/_ref/libpkref_decoder.so) and tests/test_gpu_decoder.py runs it on decodables
 * made on the GPU.
 */
#ifndef POCKETKALDI_DECODABLE_H_
#define POCKETKALDI_DECODABLE_H_

#include "matrix.h"
#include "vector.h"

namespace pocketkaldi { class AcousticModel; }
using pocketkaldi::AcousticModel;            /* as decodable.h:12 */
#define PK_MI355_AM_T pocketkaldi::AcousticModel
#include "pk_mi355.h"

#endif  /* POCKETKALDI_DECODABLE_H_ */
