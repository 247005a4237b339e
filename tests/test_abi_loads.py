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


def test_library_exports_nothing_but_the_declared_abi():
    """Round 5: the link goes through csrc/libpk_mi355.map -- the dynamic symbol table IS the header (launchers, kernel
    stubs and host helpers of the five host units stay local), so splitting capi.hip changed no exported name."""
    import subprocess
    pk.lib()
    out = subprocess.check_output(["nm", "-D", "--defined-only", pk.lib_path()], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_functions()


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


# ---- WAV ingestion is host-only (pcm_reader.cc:45-220): checked here without a GPU

GOLDEN = os.path.join(REPO, "tests", "golden")


def test_wav_reader_reference_facts():
    w = pk.read_wav(os.path.join(GOLDEN, "en-us-hello.wav"))
    assert w.dtype == np.float32 and w.shape == (7802,)
    assert list(w[:3]) == [12, 38, -8]
    from oracle import oracle as O
    assert np.array_equal(w, O.wav_read(os.path.join(GOLDEN, "en-us-hello.wav")))
    assert np.array_equal(pk.read_wav(os.path.join(GOLDEN, "en-us-cat.wav")),
                          O.wav_read(os.path.join(GOLDEN, "en-us-cat.wav")))


def test_wav_reader_rejects_what_the_reference_rejects(tmp_path):
    good = open(os.path.join(GOLDEN, "en-us-hello.wav"), "rb").read()

    def write(name, data):
        p = tmp_path / name
        p.write_bytes(data)
        return str(p)

    import struct
    cases = {
        "riff": b"RIFX" + good[4:],
        "size": good[:4] + struct.pack("<i", 1) + good[8:],
        "stereo": good[:22] + struct.pack("<h", 2) + good[24:],
        "rate": good[:24] + struct.pack("<i", 8000) + good[28:],
        "data": good[:40] + struct.pack("<i", 10) + good[44:],
        "short": good[:20],
    }
    for name, data in cases.items():
        with pytest.raises(pk.PkError):
            pk.read_wav(write(name + ".wav", data))
    with pytest.raises(pk.PkError):
        pk.read_wav(str(tmp_path / "missing.wav"))
    # 8-bit and 32-bit PCM are accepted, unscaled
    n = 100
    for bits, fmt, vals in ((8, "b", [-5, 7]), (32, "i", [-70000, 123456])):
        payload = b"".join(struct.pack("<" + fmt, vals[i % 2]) for i in range(n))
        hdr = (b"RIFF" + struct.pack("<i", 36 + len(payload)) + b"WAVEfmt " +
               struct.pack("<ihhiihh", 16, 1, 1, 16000, 16000 * bits // 8, bits // 8, bits) +
               b"data" + struct.pack("<i", len(payload)))
        w = pk.read_wav(write("pcm%d.wav" % bits, hdr + payload))
        assert list(w[:2]) == vals and w.shape == (n,)


def test_model_config_errors_without_gpu(tmp_path):
    """pk_mi355_load parses the reference's key = value model file (configuration.cc:16-72) before
    anything touches the device: the reference's error cases are reported the same way."""
    import struct
    import pocketkaldi_amd as pk

    def load(text):
        p = tmp_path / "model.conf"
        p.write_text(text)
        return pk.AcousticModel.load(str(p))

    with pytest.raises(pk.PkError, match="cannot open"):
        pk.AcousticModel.load(str(tmp_path / "absent.conf"))
    with pytest.raises(pk.PkError, match="Unexpected line"):
        load("# comment\nnnet am.nnet\n")
    with pytest.raises(pk.PkError, match="Unexpected line"):
        load("nnet = a = b\n")
    with pytest.raises(pk.PkError, match="empty"):
        load("nnet =   \n")
    with pytest.raises(pk.PkError, match="Unable to find key 'cmvn_stats'"):
        load("\n  # only a comment\nNNET = am.nnet\n")
    with pytest.raises(pk.PkError, match="cannot open .*/stats.bin"):
        load("cmvn_stats = stats.bin\n")            # relative to the config's directory
    with open(tmp_path / "stats.bin", "wb") as f:   # 3 entries instead of 41
        f.write(b"VEC0" + struct.pack("<ii", 16, 3) + struct.pack("<3f", 1, 2, 3))
    with pytest.raises(pk.PkError, match="41 expected"):
        load("cmvn_stats = stats.bin\n")
    with open(tmp_path / "stats.bin", "wb") as f:
        f.write(b"VEC0" + struct.pack("<ii", 41 * 4 + 4, 41) + struct.pack("<41f", *range(41)))
    with pytest.raises(pk.PkError, match="Unable to find key 'nnet'"):
        load("Cmvn_Stats = stats.bin\n")            # keys are case-insensitive
    with pytest.raises(pk.PkError, match="Unable to find key 'right_context'"):
        load("cmvn_stats = stats.bin\nnnet = a\nprior = b\nleft_context = 5\n")


def test_broadcast_entry_reports_misuse_without_gpu():
    """pk_mi355_am_broadcast (the C-ABI form of the one collective): call-order and argument errors are
    reported through the status channel before anything touches RCCL or the device."""
    L = pk.lib()
    am = L.pk_mi355_am_create()
    try:
        assert L.pk_mi355_am_broadcast(am, None, 0, None) != 0
        assert b"not finalized" in L.pk_mi355_last_error()
        assert L.pk_mi355_am_broadcast(None, None, 0, None) != 0
    finally:
        L.pk_mi355_am_destroy(am)


def test_readers_survive_mutated_files(tmp_path):
    """The host-side readers parse files a caller hands them (pcm_reader.cc:45-220, nnet.cc:80-147, vector.cc:393-425):
    truncated, bit-flipped, length-field-poisoned and over-long variants of the reference-written model files and of the
    reference's WAV must end in an error code (or a valid parse), never in a crash or an exception across the C ABI.
    (No GPU here: a model that parses stops at PK_MI355_E_DEVICE when it would be uploaded.)"""
    import ctypes as C
    import random
    L = pk.lib()
    D = os.path.join(GOLDEN, "refmodel")
    names = ["refmodel.nnet", "refmodel.prior", "refmodel_tid2pdf.bin", "refmodel_cmvn.bin"]
    orig = {n: open(os.path.join(D, n), "rb").read() for n in names}
    conf = ("cmvn_stats = refmodel_cmvn.bin\nnnet = refmodel.nnet\nprior = refmodel.prior\nleft_context = 2\n"
            "right_context = 1\nnum_pdfs = 18\ntid2pdf = refmodel_tid2pdf.bin\n")
    (tmp_path / "m.conf").write_text(conf)
    rng = random.Random(7)

    def mutate(data):
        data = bytearray(data)
        mode = rng.randrange(4)
        if mode == 0:
            return bytes(data[:rng.randrange(len(data))])
        if mode == 1:
            for _ in range(rng.randrange(1, 6)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        elif mode == 2:
            pos = rng.randrange(0, max(1, len(data) - 4))
            data[pos:pos + 4] = rng.choice([0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 0, 1 << 30]).to_bytes(4, "little")
        else:
            data += bytes(rng.randrange(1, 64))
        return bytes(data)

    codes = set()
    for _ in range(400):
        for n in names:
            (tmp_path / n).write_bytes(orig[n])
        victim = rng.choice(names)
        (tmp_path / victim).write_bytes(mutate(orig[victim]))
        h, stats = C.c_void_p(), np.zeros(41, np.float32)
        rc = L.pk_mi355_load(str(tmp_path / "m.conf").encode(), 0, C.byref(h), stats.ctypes.data_as(C.POINTER(C.c_float)))
        codes.add(rc)
        if rc == 0:
            L.pk_mi355_am_destroy(h)
    assert codes <= {0, -1, -2, -3} and -3 in codes

    wav = open(os.path.join(GOLDEN, "en-us-hello.wav"), "rb").read()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    codes = set()
    for _ in range(400):
        data = bytearray(wav)
        mode = rng.randrange(3)
        if mode == 0:
            data = data[:rng.randrange(len(data))]
        elif mode == 1:
            for _ in range(rng.randrange(1, 6)):
                data[rng.randrange(0, 64)] = rng.randrange(256)
        else:
            pos = rng.randrange(0, 40)
            data[pos:pos + 4] = rng.choice([0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 0]).to_bytes(4, "little")
        (tmp_path / "x.wav").write_bytes(bytes(data))
        v = pk.pk_vector_t(0, None)
        rc = L.pk_mi355_16kpcm_read(str(tmp_path / "x.wav").encode(), C.byref(v))
        codes.add(rc)
        if v.data:
            libc.free(C.cast(v.data, C.c_void_p))
    assert codes <= {0, -3} and -3 in codes
