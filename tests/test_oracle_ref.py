"""The oracle against the REAL reference code that builds in this image
(oracle/_ref = srfft.cc + gemm.cc + gemm_haswell.cc compiled from /root/reference)
and against the committed outputs of that build (tests/golden/ref_*.npz).
Bit-exact.  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_srfft512_matches_committed_reference_output():
    z = np.load(os.path.join(G, "ref_srfft512.npz"))
    f = O.Srfft(512)
    for fr, sp in zip(z["frames"], z["spectra"]):
        assert np.array_equal(_bits(f.forward(fr)), _bits(sp))


def test_sgemm_matches_committed_reference_output():
    z = np.load(os.path.join(G, "ref_sgemm.npz"))
    for i in range(4):
        A, B, Cref = z["A%d" % i], z["B%d" % i], z["C%d" % i]
        assert np.array_equal(_bits(O.sgemm(A, B)), _bits(Cref))
        assert np.array_equal(_bits(O.sgemm_naive(A, B)), _bits(Cref))


needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 128, 512, 1024])
def test_srfft_bitwise_vs_live_reference(n):
    rng = np.random.default_rng(n)
    f = O.Srfft(n)
    for scale in (1.0, 3e4, 1e-3):
        x = (rng.standard_normal(n) * scale).astype(np.float32)
        assert np.array_equal(_bits(f.forward(x)), _bits(O.ref_srfft(x)))


@needs_ref
@pytest.mark.parametrize("shape", [(7, 16, 440), (37, 50, 1024), (13, 33, 2048),
                                   (300, 20, 513), (6, 16, 512), (5, 4100, 30), (1, 1, 1)])
def test_sgemm_bitwise_vs_live_reference(shape):
    m, n, k = shape
    rng = np.random.default_rng(m * 131 + k)
    A = rng.standard_normal((m, k)).astype(np.float32)
    B = rng.standard_normal((k, n)).astype(np.float32)
    r = O.ref_sgemm(A, B)
    assert np.array_equal(_bits(O.sgemm(A, B)), _bits(r))
    assert np.array_equal(_bits(O.sgemm_naive(A, B)), _bits(r))
