"""the model-file formats, pinned with files written by the REFERENCE's
own converters (tool/convert_am.py, convert_trans.py, convert_cmvn_stats.py, run in the build
container by tests/golden/make_model_fixtures.py on our seeded text inputs; outputs committed
under tests/golden/refmodel/).  The in-memory model is rebuilt from the TEXT by an independent
parser (tests/refmodel_text.py); readers must agree with it bit for bit."""
import os
import shutil
import struct

import numpy as np
import pytest

import pocketkaldi_amd as pk
from oracle import oracle as O
from refmodel_text import DIR, load_text_model

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def test_reference_written_files_have_the_documented_layout():
    """nnet.cc:80-147 / matrix.cc:288-319 / vector.cc:393-425 as byte facts of the reference's output."""
    raw = open(os.path.join(DIR, "refmodel.nnet"), "rb").read()
    assert raw[:4] == b"NNT0" and struct.unpack("<ii", raw[4:12]) == (4, 7)
    assert raw[12:16] == b"LAY0" and struct.unpack("<ii", raw[16:24]) == (4, 0)
    assert raw[24:28] == b"MAT0" and struct.unpack("<iii", raw[28:40]) == (8, 24, 160)
    assert raw[40:44] == b"VEC0" and struct.unpack("<ii", raw[44:52]) == (160 * 4 + 4, 160)
    tid = open(os.path.join(DIR, "refmodel_tid2pdf.bin"), "rb").read()
    assert tid[:4] == b"VEC0" and struct.unpack("<ii", tid[4:12]) == (58 * 4 + 4, 58)
    assert len(open(os.path.join(DIR, "refmodel_cmvn.bin"), "rb").read()) == 12 + 41 * 4


def test_oracle_readers_agree_with_the_text_model():
    layers, prior, L, R, tid2pdf, cmvn41 = load_text_model()
    assert (L, R) == (2, 1) and [l[0] for l in layers] == ["linear", "relu", "normalize", "linear", "relu", "linear", "softmax"]
    nn = O.Nnet.read(os.path.join(DIR, "refmodel.nnet"))
    x = np.random.default_rng(0).standard_normal((9, 160)).astype(np.float32)
    assert bits_equal(nn.propagate(x), O.Nnet(layers).propagate(x))
    assert bits_equal(O.read_vec(os.path.join(DIR, "refmodel.prior")), prior)
    assert bits_equal(O.read_vec(os.path.join(DIR, "refmodel_cmvn.bin")), cmvn41)
    assert np.array_equal(O.read_vec(os.path.join(DIR, "refmodel_tid2pdf.bin")).view(np.int32), tid2pdf)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_pk_load_reads_the_reference_written_model(precision):
    """pk_mi355_load (pk_load's share, pocketkaldi.cc:72-144) on the reference-written files ==
    the model built in memory from the text, bit for bit; and == the oracle on the same features."""
    layers, prior, L, R, tid2pdf, cmvn41 = load_text_model()
    # (the model has a Normalize layer: f16x3 covers (Linear [ReLU] [Normalize])+ [Softmax] since round 3)
    am_file, stats = pk.AcousticModel.load(os.path.join(DIR, "refmodel.conf"), precision=precision)
    am_mem = pk.AcousticModel(layers, prior, L, R, tid2pdf, precision=precision)
    assert bits_equal(stats, cmvn41)
    assert am_file.num_pdfs() == 18 and am_file.input_dim() == 160
    assert [am_file.transition_id_to_pdf_id(t) for t in range(58)] == list(tid2pdf)
    feats = np.random.default_rng(4).standard_normal((33, 40)).astype(np.float32)
    lp_file = pk.Decodable(am_file, 0.1, feats).log_prob()
    assert bits_equal(lp_file, pk.Decodable(am_mem, 0.1, feats).log_prob())
    ref = O.Nnet.read(os.path.join(DIR, "refmodel.nnet")).am_compute(feats, prior, L, R, 0.1)
    assert np.all(np.abs(lp_file.astype(np.float64) - ref) <= 1e-4 * np.maximum(np.abs(ref), 1.0))
    if precision == "f32":          # with the reference's softmax arithmetic: the reference's bits
        am_file.set_softmax("reference")
        assert bits_equal(pk.Decodable(am_file, 0.1, feats).log_prob(), ref)


@pytest.mark.gpu
def test_whole_path_with_reference_written_model_and_reference_cmvn_stats(tmp_path):
    """The reference's own test/data/cmvn_stats.bin through the PRODUCT's reader (pk_mi355_load),
    then WAV -> fbank -> CMVN -> nnet with the reference-written model: bit-identical to the oracle."""
    for name in ("refmodel.nnet", "refmodel.prior", "refmodel_tid2pdf.bin"):
        shutil.copyfile(os.path.join(DIR, name), tmp_path / name)
    shutil.copyfile(os.path.join(G, "cmvn_stats.bin"), tmp_path / "cmvn_stats.bin")
    (tmp_path / "m.conf").write_text("cmvn_stats = cmvn_stats.bin\nnnet = refmodel.nnet\nprior = refmodel.prior\n"
                                     "left_context = 2\nright_context = 1\nnum_pdfs = 18\ntid2pdf = refmodel_tid2pdf.bin\n")
    am, stats = pk.AcousticModel.load(str(tmp_path / "m.conf"))
    want = O.read_vec(os.path.join(G, "cmvn_stats.bin"))
    assert bits_equal(stats, want) and stats.shape == (41,) and abs(stats[40] - 3.616e7) < 1e4
    am.set_softmax("reference")
    wave = pk.read_wav(os.path.join(G, "en-us-hello.wav"))
    d = pk.process_acoustic(am, stats, wave, 0.1)
    layers, prior, L, R, tid2pdf, _ = load_text_model()
    ref = O.Nnet(layers).am_compute(O.cmvn(want, O.Fbank().compute(wave)), prior, L, R, 0.1)
    assert bits_equal(d.log_prob(), ref)
    assert d.loglikelihood(46, 57) == ref[46, tid2pdf[57]] and d.is_last_frame(46)
