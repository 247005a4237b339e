"""Range safety of the fp16 matrix-core modes (include/pk_mi355.h, PK_MI355_PRECISION_F16X3 / _F16; VERDICT round 3,
next #1).  fp16 holds 2^-24 .. 65504 and the lo half of a split operand sits 2^-12 below its value, so the
"1e-6 on log-likelihoods" of f16x3 holds only while operands sit inside that window:
  * weights are placed there by an exact power-of-two prescale at finalize, whatever scale the model has;
  * activations carry a per-operand exponent (calibration), and every call checks what it wrote --
    saturation at 65504 or an operand whose lo halves are all subnormal FAILS the call, loudly.
Oracle: nnet.cc:22-36 restated (oracle/pk_oracle.c); tolerance: north_star's 1e-4, asserted at 2e-5.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import pocketkaldi_amd as pk
from pocketkaldi_amd import synth
from oracle import oracle as O


def _rel_err(got, ref):
    return np.max(np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0))


def _rescaled_S(down=-10, up=10):
    """Model S with the second affine layer's weights AND bias times 2^down and the third layer's weights times
    2^up: ReLU is positively homogeneous and powers of two are exact, so the network computes bit-identical values
    in fp32 -- but its second hidden activation lives 2^down lower."""
    layers, prior, L, R = synth.model("S")
    lin = [i for i, l in enumerate(layers) if l[0] == "linear"]
    out = list(layers)
    out[lin[1]] = ("linear", layers[lin[1]][1] * np.float32(2.0 ** down), layers[lin[1]][2] * np.float32(2.0 ** down))
    out[lin[2]] = ("linear", layers[lin[2]][1] * np.float32(2.0 ** up), layers[lin[2]][2])
    return layers, out, prior, L, R


def _feats(seconds=2.0, utt=7):
    g = synth.global_cmvn_stats()
    return O.cmvn(g, O.Fbank().compute(synth.utterance(utt, seconds)))


def test_rescaled_layers_fail_loudly_until_calibrated_then_match_the_oracle():
    orig, layers, prior, L, R = _rescaled_S()
    feats = _feats()
    ref = O.Nnet(layers).am_compute(feats, prior, L, R, 0.1)
    # the construction is exact: the oracle cannot tell the two networks apart
    assert np.array_equal(ref, O.Nnet(orig).am_compute(feats, prior, L, R, 0.1))

    am = pk.AcousticModel(layers, prior, L, R, precision="f16x3")
    w_exp, x_exp = am.exponents()
    w0, _ = pk.AcousticModel(orig, prior, L, R, precision="f16x3").exponents()
    # the weight prescale is automatic and exact: the rescaled layers end up on the SAME fp16 grid
    assert list(w_exp - w0) == [0, 10, -10, 0, 0] and list(x_exp) == [0] * 5
    for e, l in zip(w_exp, [l for l in layers if l[0] == "linear"]):
        assert 2.0 ** 13 <= np.abs(l[1]).max() * 2.0 ** int(e) < 2.0 ** 14

    # uncalibrated: the input of affine layer 2 peaks near 2^-8 -- every lo half subnormal.  Loud, not quiet.
    with pytest.raises(pk.PkError, match="affine layer 2 is too small"):
        pk.Decodable(am, 0.1, feats)
    bs = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, 32000)
    bs.set_waves([synth.utterance(7, 2.0)])
    with pytest.raises(pk.PkError, match="too small"):
        bs.score(0.1)
    with pytest.raises(pk.PkError, match="too small"):          # sticky: nothing is delivered afterwards either
        bs.fetch(0)
    bs.score(0.1, sync=False)
    with pytest.raises(pk.PkError, match="too small"):
        bs.synchronize()

    am.calibrate(feats)
    _, x_exp = am.exponents()
    am0 = pk.AcousticModel(orig, prior, L, R, precision="f16x3").calibrate(feats)
    assert list(x_exp - am0.exponents()[1]) == [0, 0, 10, 0, 0], x_exp          # the one activation that sits 2^10 lower
    got = pk.Decodable(am, 0.1, feats).log_prob()
    err = _rel_err(got, ref)
    print("\n  rescaled model S after calibration: max err %.2e, exponents x %s w %s" % (err, list(x_exp), list(w_exp)))
    assert err < 2e-5
    # the batch path reads the same exponent words
    bs.score(0.1)
    assert np.array_equal(bs.fetch(0).log_prob(), got)
    # calibrating the unscaled model lands on the same grid: identical bits from the rescaled network
    assert np.array_equal(pk.Decodable(am0, 0.1, feats).log_prob(), got)


def test_batch_calibrate_from_waves():
    _, layers, prior, L, R = _rescaled_S()
    g = synth.global_cmvn_stats()
    waves = [synth.utterance(20, 1.0), synth.utterance(21, 2.5)]
    am = pk.AcousticModel(layers, prior, L, R, precision="f16x3")
    bs = pk.BatchScorer(am, g, 2, sum(len(w) for w in waves))
    bs.set_waves(waves)
    bs.calibrate()
    with pytest.raises(pk.PkError, match="not scored"):          # calibration passes are not results
        bs.fetch(0)
    bs.score(0.1)
    nn = O.Nnet(layers)
    for u, w in enumerate(waves):
        ref = nn.am_compute(O.cmvn(g, O.Fbank().compute(w)), prior, L, R, 0.1)
        assert _rel_err(bs.fetch(u).log_prob(), ref) < 2e-5


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
def test_saturation_is_reported_not_clamped_quietly(prec):
    """First affine layer (weights and bias) times 2^15, second layer's weights times 2^-15: the same network in
    fp32 -- the oracle's log-likelihoods do not move -- with a first hidden activation near 2^17."""
    layers, prior, L, R = synth.model("tiny")
    lin = [i for i, l in enumerate(layers) if l[0] == "linear"]
    big = list(layers)
    big[lin[0]] = ("linear", layers[lin[0]][1] * np.float32(2.0 ** 15), layers[lin[0]][2] * np.float32(2.0 ** 15))
    big[lin[1]] = ("linear", layers[lin[1]][1] * np.float32(2.0 ** -15), layers[lin[1]][2])
    feats = _feats(1.0, 3)
    ref = O.Nnet(big).am_compute(feats, prior, L, R, 0.1)
    assert np.array_equal(ref, O.Nnet(layers).am_compute(feats, prior, L, R, 0.1)) and np.all(np.isfinite(ref))
    am = pk.AcousticModel(big, prior, L, R, precision=prec)
    with pytest.raises(pk.PkError, match="affine layer 1 saturated"):
        pk.Decodable(am, 0.1, feats)
    bs = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, 16000)
    bs.set_waves([synth.utterance(3, 1.0)])
    with pytest.raises(pk.PkError, match="saturated"):
        bs.score(0.1)
    with pytest.raises(pk.PkError, match="saturated"):
        bs.fetch_all()
    # views handed out BEFORE the verdict (sync = 0) are voided by it: loglikelihood answers NaN
    bs.score(0.1, sync=False)
    early = bs.fetch_all(sync=False)
    with pytest.raises(pk.PkError, match="saturated"):
        bs.synchronize()
    assert np.isnan(early[0].loglikelihood(0, 1))
    for v in early:
        v.destroy()
    # an explicit exponent (or calibration) brings the operand back into range
    am.set_input_exponents([0, -14, 0])
    tol = 2e-5 if prec == "f16x3" else 2e-2
    assert _rel_err(pk.Decodable(am, 0.1, feats).log_prob(), ref) < tol
    bs.score(0.1)                                   # the verdict of the earlier call is not sticky across a new score
    assert bs.fetch_all()[0].log_prob().shape == (98, 50)
    am.set_input_exponents([0, 0, 0])
    am.calibrate(feats)
    assert am.exponents()[1][1] < -4
    assert _rel_err(pk.Decodable(am, 0.1, feats).log_prob(), ref) < tol
    with pytest.raises(pk.PkError):
        am.set_input_exponents([0, 31, 0])
    with pytest.raises(pk.PkError):
        am.set_input_exponents([0, 0])


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_range_verdict_is_this_calls_own(lanes, monkeypatch):
    """the verdict of a score call must not depend on what an EARLIER call left in the scorer's buffers.
    A batch that saturates operand 0 (two utterances with long silences: CMVN output near -28; two layer-stack chunks,
    so with PK_MI355_LANES=2 both lanes take part), then a SMALLER healthy batch on the same scorer (one chunk, one
    lane): the columns the first batch filled behind the second one's last column, and lane 2's range words, used to
    count towards the second call's verdict.  Also: an empty batch after a failed one reports nothing."""
    monkeypatch.setenv("PK_MI355_CHUNK", "256")
    monkeypatch.setenv("PK_MI355_LANES", lanes)
    layers, prior, L, R = synth.model("tiny")
    g = synth.global_cmvn_stats()
    loud = [np.concatenate([np.zeros(24000, np.float32), synth.utterance(40 + i, 1.5)]) for i in range(2)]
    calm = synth.utterance(50, 1.0)
    fb = O.Fbank()
    m_loud = max(np.abs(O.cmvn(g, fb.compute(w))).max() for w in loud)
    m_calm = np.abs(O.cmvn(g, fb.compute(calm))).max()
    assert m_loud > 2.2 * m_calm, (m_loud, m_calm)            # room for a power of two between the two maxima
    e0 = int(np.floor(np.log2(65504.0 / m_calm))) - 0         # the largest exponent the calm batch stays under the clamp with
    while m_calm * 2.0 ** e0 >= 65000.0:
        e0 -= 1
    assert m_loud * 2.0 ** e0 > 65504.0
    am = pk.AcousticModel(layers, prior, L, R, precision="f16x3")
    am.set_input_exponents([e0, 0, 0])
    bs = pk.BatchScorer(am, g, 2, sum(len(w) for w in loud))
    bs.set_waves(loud)
    with pytest.raises(pk.PkError, match="affine layer 0 saturated"):
        bs.score(0.1)
    bs.set_waves([calm])                                      # 98 frames: one chunk, far fewer columns than before
    bs.score(0.1)                                             # must pass: nothing of the loud batch counts
    ref = O.Nnet(layers).am_compute(O.cmvn(g, fb.compute(calm)), prior, L, R, 0.1)
    assert _rel_err(bs.fetch(0).log_prob(), ref) < 2e-5
    bs.set_waves(loud)
    with pytest.raises(pk.PkError, match="saturated"):
        bs.score(0.1)
    bs.set_waves([])                                          # an empty batch: no results, and no verdict of its own or anyone's
    bs.score(0.1)
    bs.synchronize()
    assert bs.fetch_all() == []
    bs.close()


@pytest.mark.parametrize("log2_scale", [-20, -12, -8, -4, 0, 6, 12])
def test_weight_scale_is_irrelevant(log2_scale):
    """One affine layer whose weights live at He-normal x 2^s: the error budget of the unscaled case (2^-20 of
    sum |x||w|) must hold at every s.  Round 3's arithmetic (no prescale) is at 1.3e-4 for s = -8 and at plain-fp16
    accuracy, 1.4e-3, for s = -12."""
    T, K, N = 64, 1024, 512
    rng = np.random.default_rng(5)
    W = (rng.standard_normal((N, K)) * np.sqrt(2.0 / K)).astype(np.float32) * np.float32(2.0 ** log2_scale)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32) * np.float32(2.0 ** log2_scale)
    x = rng.standard_normal((T, K)).astype(np.float32)
    layers = [("linear", W, b)]
    am = pk.AcousticModel(layers, num_pdfs=N, precision="f16x3")
    gpu = am.propagate(x)
    ref = O.Nnet(layers).propagate(x)
    budget = (np.abs(x) @ np.abs(W.T)) + np.abs(b)
    assert np.max(np.abs(gpu - ref) / budget) < 2.0 ** -20
    w_exp, _ = am.exponents()
    assert 2.0 ** 13 <= np.abs(W).max() * 2.0 ** int(w_exp[0]) < 2.0 ** 14


def test_non_finite_weights_are_refused_in_f16_modes():
    W = np.eye(8, dtype=np.float32)
    W[3, 3] = np.inf
    with pytest.raises(pk.PkError, match="non-finite"):
        pk.AcousticModel([("linear", W, np.zeros(8, np.float32))], num_pdfs=8, precision="f16x3")
    pk.AcousticModel([("linear", W, np.zeros(8, np.float32))], num_pdfs=8, precision="f32")      # F32 carries it


def test_f32_models_have_no_exponents_and_calibrate_is_a_no_op():
    layers, prior, L, R = synth.model("tiny")
    am = pk.AcousticModel(layers, prior, L, R)
    w, x = am.exponents()
    assert list(w) == [0, 0, 0] and list(x) == [0, 0, 0]
    am.calibrate(_feats(0.5, 1))
    with pytest.raises(pk.PkError):
        am.set_input_exponents([0, 0, 0])


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_the_broadcast_carries_the_calibration():
    """The exponent words are part of the weight blob: after the one collective (RCCL at one rank here, out-of-place
    form: root sends `am`'s blob, `other` receives) a model that was built from zeros and never calibrated holds the
    root's exponents and scores like it."""
    try:
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    pk.set_device(0)
    _, layers, prior, L, R = _rescaled_S()
    feats = _feats(1.0, 9)
    am = pk.AcousticModel(layers, prior, L, R, precision="f16x3").calibrate(feats)
    want = pk.Decodable(am, 0.1, feats).log_prob()
    zl = [(l[0], np.zeros_like(l[1]), np.zeros_like(l[2])) if l[0] == "linear" else l for l in layers]
    other = pk.AcousticModel(zl, np.full_like(prior, 1.0), L, R, precision="f16x3")
    assert list(other.exponents()[0]) == [0] * 5
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        other.broadcast(comm.value, root=0, src=am)
    finally:
        rccl.ncclCommDestroy(comm)
    for a, b in zip(other.exponents(), am.exponents()):
        assert list(a) == list(b)
    assert np.array_equal(pk.Decodable(other, 0.1, feats).log_prob(), want)


def _fuzz_seeds(n):
    import os
    base = int(os.environ.get("PK_FUZZ_BASE", "0"))
    return range(base, base + int(os.environ.get("PK_FUZZ_SEEDS", n)))


@pytest.mark.parametrize("seed", _fuzz_seeds(8))
def test_fuzz_rescaled_networks_calibrate_onto_the_same_grid(seed):
    """Random ReLU networks, every hidden activation moved by a random power of two (h_i' = h_i 2^k_i: W_i' = W_i
    2^(k_i - k_(i-1)), b_i' = b_i 2^k_i; the logits stay): the same network in fp32, bit for bit in the oracle.  After
    calibration the split-fp16 operands of the rescaled network ARE those of the original (w_exp absorbs the weight
    scale, x_exp the activation scale, both exact), so f16x3 returns the same bits for both -- and both sit inside
    2e-5 of the oracle.  Uncalibrated, a network pushed far enough either way must fail loudly, never quietly."""
    rng = np.random.default_rng(0xCA1B + seed)
    nh = int(rng.integers(1, 4))
    dims = [440] + [int(rng.integers(48, 400)) for _ in range(nh)] + [int(rng.integers(30, 700))]
    layers = []
    for i in range(len(dims) - 1):
        W = (rng.standard_normal((dims[i + 1], dims[i])) * np.sqrt(2.0 / dims[i])).astype(np.float32)
        b = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
        layers.append(("linear", W, b))
        layers.append(("relu",) if i < len(dims) - 2 else ("softmax",))
    prior = rng.uniform(0.5, 1.5, dims[-1])
    prior = (prior / prior.sum()).astype(np.float32)
    k = [0] + [int(rng.integers(-12, 13)) for _ in range(nh)] + [0]
    moved, li = [], 0
    for l in layers:
        if l[0] == "linear":
            moved.append(("linear", l[1] * np.float32(2.0 ** (k[li + 1] - k[li])), l[2] * np.float32(2.0 ** k[li + 1])))
            li += 1
        else:
            moved.append(l)
    feats = _feats(0.6, 40 + seed)
    ref = O.Nnet(layers).am_compute(feats, prior, 5, 5, 0.1)
    assert np.array_equal(ref, O.Nnet(moved).am_compute(feats, prior, 5, 5, 0.1))
    am0 = pk.AcousticModel(layers, prior, 5, 5, precision="f16x3").calibrate(feats)
    am1 = pk.AcousticModel(moved, prior, 5, 5, precision="f16x3")
    try:
        quiet = pk.Decodable(am1, 0.1, feats).log_prob()          # in range by luck of the draw: then it must be accurate
        assert _rel_err(quiet, ref) < 1e-4
    except pk.PkError as e:
        assert "range" in str(e)
    am1.calibrate(feats)
    (w0, x0), (w1, x1) = am0.exponents(), am1.exponents()
    assert list(x1 - x0) == [-kk for kk in k[:-1]], (k, x0, x1)
    assert list(w1 - w0) == [k[i] - k[i + 1] for i in range(len(k) - 1)]
    got0, got1 = pk.Decodable(am0, 0.1, feats).log_prob(), pk.Decodable(am1, 0.1, feats).log_prob()
    assert np.array_equal(got0, got1)
    assert _rel_err(got1, ref) < 2e-5
