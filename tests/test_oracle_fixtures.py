"""Pin the CPU oracle against every fixture the reference's own tests hold for
the hot path: srfft_test's 128-point pair, the Kaldi fbank /
online-CMVN dumps used by fbank_test / cmvn_test, nnet_test's layer answers.
CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KNOWN = json.load(open(os.path.join(G, "ref_known_answers.json")))


def test_srfft128_known_answer():
    k = KNOWN["srfft128"]            # test/srfft_test.cc:284 asserts |diff| < 1e-6
    y = O.Srfft(128).forward(np.array(k["input"], dtype=np.float32))
    assert np.max(np.abs(y - np.array(k["output"], dtype=np.float32))) < k["tol_abs"]


def test_wav_reader_matches_reference_facts():
    w = O.wav_read(os.path.join(G, "en-us-hello.wav"))
    k = KNOWN["wav_hello"]
    assert w.shape[0] == k["num_samples"]
    assert list(w[:3]) == k["first"]
    assert O.num_frames(w.shape[0]) == k["num_frames"]


def test_num_frames_edges():          # fbank.cc:35-42
    assert O.num_frames(0) == 0
    assert O.num_frames(399) == 0
    assert O.num_frames(400) == 1
    assert O.num_frames(559) == 1
    assert O.num_frames(560) == 2
    assert O.num_frames(160000) == 998


# The Kaldi dumps carry 7 significant digits at magnitude 10..19, i.e. they are
# only good to ~1e-5 absolute themselves; the reference's own build is within
# 2.2e-5 of them.  3e-5 is therefore the tightest honest bound.
KALDI_DUMP_TOL = 3e-5


def test_fbank_vs_kaldi_dump():       # test/fbank_test.cc:15-56
    w = O.wav_read(os.path.join(G, "en-us-hello.wav"))
    feats = O.Fbank().compute(w)
    ref = np.loadtxt(os.path.join(G, "fbankmat_en-us-hello.wav.txt"), dtype=np.float64)
    assert feats.shape == (47, 40) and ref.shape == (47 * 40,)
    assert np.max(np.abs(feats.reshape(-1) - ref)) < KALDI_DUMP_TOL


def test_cmvn_vs_kaldi_dump():        # test/cmvn_test.cc:33-82
    w = O.wav_read(os.path.join(G, "en-us-hello.wav"))
    feats = O.Fbank().compute(w)
    stats = O.read_vec(os.path.join(G, "cmvn_stats.bin"))
    assert stats.shape == (41,)
    out = O.cmvn(stats, feats)
    ref = np.loadtxt(os.path.join(G, "fbankcmvnmat_en-us-hello.wav.txt"), dtype=np.float64)
    assert np.max(np.abs(out.reshape(-1) - ref)) < KALDI_DUMP_TOL


def test_linear_known_answer():       # test/nnet_test.cc:23-55
    k = KNOWN["linear"]
    nn = O.Nnet([("linear", np.array(k["W"]), np.array(k["b"]))])
    y = nn.propagate(np.array([k["x"]], dtype=np.float32))
    assert np.max(np.abs(y[0] - np.array(k["y"]))) < k["tol_abs"]


def test_softmax_known_answer():      # test/nnet_test.cc:57-73
    k = KNOWN["softmax"]
    y = O.Nnet([("softmax",)]).propagate(np.array([k["x"]], dtype=np.float32))
    assert np.max(np.abs(y[0] - np.array(k["y"]))) < k["tol_abs"]


def test_relu_known_answer():         # test/nnet_test.cc:75-92
    k = KNOWN["relu"]
    y = O.Nnet([("relu",)]).propagate(np.array([k["x"]], dtype=np.float32))
    assert np.max(np.abs(y[0] - np.array(k["y"]))) < k["tol_abs"]


def test_normalize_known_answer():    # test/nnet_test.cc:94-110
    k = KNOWN["normalize"]
    y = O.Nnet([("normalize",)]).propagate(np.array([k["x"]], dtype=np.float32))
    assert abs(float(np.sum(y.astype(np.float64) ** 2)) - k["sum_sq"]) < k["tol_abs"]


def test_gemm_property_like_gemm_test():   # test/gemm_test.cc:32-62
    rng = np.random.default_rng(7)
    for (m, n, k) in [(512, 512, 512), (100, 100, 1), (1, 1, 1), (121, 233, 17)]:
        A = rng.random((m, k), dtype=np.float32)
        B = rng.random((k, n), dtype=np.float32)
        got = O.sgemm(A, B)
        want = A.astype(np.float64) @ B.astype(np.float64)
        assert np.max(np.abs(got - want)) < 0.01


def test_splice_edges():              # am.cc:65-88
    f = np.arange(3 * 2, dtype=np.float32).reshape(3, 2)
    s = O.splice(f, 2, 1)
    assert s.shape == (3, 8)
    assert list(s[0]) == [0, 1, 0, 1, 0, 1, 2, 3]
    assert list(s[2]) == [0, 1, 2, 3, 4, 5, 4, 5]
