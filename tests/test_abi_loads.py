"""CPU-only: the C-ABI library builds/loads and exports every symbol that
include/pk_mi355.h declares; compute entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import pocketkaldi_amd as pk

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(REPO, "include", "pk_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(pk_(?:mi355|decodable)_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(pk.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(pk.lib_path()) if os.path.exists(pk.lib_path()) else pk.lib()
    for name in declared_functions():
        assert hasattr(L, name), name


def test_struct_layout_matches_reference_abi():
    # decodable.h:15-18 on LP64: {int ncol; int nrow; float *data;} + pointer = 16 + 8
    assert ctypes.sizeof(pk.pk_matrix_t) == 16
    assert ctypes.sizeof(pk.pk_vector_t) == 16
    assert ctypes.sizeof(pk.pk_decodable_t) == 24
    assert pk.pk_matrix_t.ncol.offset == 0 and pk.pk_matrix_t.nrow.offset == 4
    assert pk.pk_matrix_t.data.offset == 8
    assert pk.pk_decodable_t.am.offset == 16


def test_num_frames_host_logic():        # fbank.cc:35-42
    assert [pk.num_frames(n) for n in (0, 399, 400, 559, 560, 7802, 160000)] == [0, 0, 1, 1, 2, 47, 998]


def test_model_structure_errors_without_device():
    L = pk.lib()
    am = L.pk_mi355_am_create()
    try:
        assert L.pk_mi355_am_add_layer(am, 4) != 0          # nnet.cc:106-127: kinds 0..3 only
        assert b"unexpected layer type" in L.pk_mi355_last_error()
    finally:
        L.pk_mi355_am_destroy(am)


@pytest.mark.skipif(pk.lib().pk_mi355_device_count() > 0, reason="a GPU is present")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(pk.PkError):
        pk.Fbank().compute(np.zeros(1600, dtype=np.float32))
    with pytest.raises(pk.PkError):
        pk.AcousticModel([("linear", np.eye(4, dtype=np.float32), np.zeros(4, np.float32))],
                         prior=np.full(4, 0.25, np.float32))
