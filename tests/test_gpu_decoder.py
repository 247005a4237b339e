"""the reference's REAL decoder consumes what libpk_mi355.so produces.

oracle/_ref/libpkref_decoder.so is the reference's decoder.cc + fst.cc + util.cc + hashtable.cc +
strlcpy.cc compiled UNMODIFIED against include/reference_binding/decodable.h (oracle/Makefile:
ref_decoder; built in the container that has /root/reference, shipped as a binary -- test
infrastructure, like oracle/_ref/libpkref.so).  Its pk_decodable_islastframe /
pk_decodable_loglikelihood calls (decoder.cc:49,252,276) bind to the product library.

  * CPU: Decoder::Decode + BestPath over a hand-made decodable and the reference's own
    test/data/testinput.fst, against an exhaustive Viterbi in Python.
  * GPU: tests/cpp/process_example.cc -- pk_process (pocketkaldi.cc:176-248) stage for stage: WAV ->
    pk_mi355_process_acoustic -> Decoder::Decode -> hyp, loglikelihood_per_frame -- against the same
    decoder run over a decodable filled with the ORACLE's log-likelihoods.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import pocketkaldi_amd as pk

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")
DECLIB = os.path.join(REPO, "oracle", "_ref", "libpkref_decoder.so")

pytestmark = pytest.mark.skipif(not os.path.exists(DECLIB),
                                reason="oracle/_ref/libpkref_decoder.so not built (needs the reference tree once)")


def read_fst(path):
    """fst.cc:30-90 -> (start, final[], arcs_by_state[]) with arcs (next, ilabel, olabel, weight)."""
    raw = open(path, "rb").read()
    assert raw[:9] == b"pk::fst_0"
    size, ns, na, start = struct.unpack("<iiii", raw[32:48])
    assert size == 12 + ns * 8 + na * 16 == len(raw) - 36
    final = np.frombuffer(raw, np.float32, ns, 48)
    first = np.frombuffer(raw, np.int32, ns, 48 + 4 * ns)
    arcs = [struct.unpack("<iiif", raw[48 + 8 * ns + 16 * i: 64 + 8 * ns + 16 * i]) for i in range(na)]
    out = []
    for s in range(ns):
        if first[s] < 0:
            out.append([])
            continue
        nxt = [first[t] for t in range(s + 1, ns) if first[t] > 0]      # Fst::CountArcs, fst.cc:93-109
        out.append(arcs[first[s]:(nxt[0] if nxt else na)])
    return start, final, out


def viterbi(fst, loglik, tid2pdf):
    """Exhaustive token passing (no beam): cost = sum(arc weight - loglik[t][pdf(ilabel)]) + final."""
    start, final, arcs = fst

    def close(tok):                                   # epsilon arcs, ProcessNonemitting (decoder.cc:186-222)
        queue = list(tok)
        while queue:
            s = queue.pop()
            for nxt, il, ol, w in arcs[s]:
                if il != 0:
                    continue
                c = tok[s][0] + w
                if nxt not in tok or c < tok[nxt][0]:
                    tok[nxt] = (c, tok[s][1] + ([ol] if ol else []))
                    queue.append(nxt)
        return tok

    tok = close({start: (0.0, [])})
    for t in range(loglik.shape[0]):
        new = {}
        for s, (c, words) in tok.items():
            for nxt, il, ol, w in arcs[s]:
                if il == 0:
                    continue
                cc = c + w - float(loglik[t, tid2pdf(il)])
                if nxt not in new or cc < new[nxt][0]:
                    new[nxt] = (cc, words + ([ol] if ol else []))
        tok = close(new)
    best = min(((c + float(final[s]), words, s) for s, (c, words) in tok.items() if np.isfinite(final[s])),
               default=None)
    return best


def decoder_lib():
    pk.lib()
    L = C.CDLL(DECLIB)
    L.pkref_decode.argtypes = [C.c_char_p, C.POINTER(pk.pk_decodable_t), C.POINTER(C.c_int), C.c_int,
                               C.POINTER(C.c_float), C.POINTER(C.c_int)]
    return L


def decode(decodable_ptr, fst_path):
    words = (C.c_int * 512)()
    weight, ok = C.c_float(0), C.c_int(0)
    n = decoder_lib().pkref_decode(fst_path.encode(), decodable_ptr, words, 512, C.byref(weight), C.byref(ok))
    assert n >= 0 and ok.value == 1
    return list(words[:n]), weight.value


def host_decodable(loglik, am_handle):
    """A pk_decodable_t over a numpy [T][num_pdfs] array (log_prob memory layout, decodable.cc:24-31)."""
    d = pk.pk_decodable_t()
    d.log_prob.ncol, d.log_prob.nrow = loglik.shape
    d.log_prob.data = loglik.ctypes.data_as(C.POINTER(C.c_float))
    d.am = am_handle
    return d


def test_fst_fixture_layout_is_the_references():
    """The reader above on the reference's own test/data/testinput.fst gives test/fst_test.cc:23-62's facts."""
    start, final, arcs = read_fst(os.path.join(G, "testinput.fst"))
    assert start == 0 and np.isinf(final[0]) and np.isinf(final[1]) and final[2] == 3.5
    assert arcs[0] == [(1, 1, 1, 0.5), (1, 2, 2, 1.5)] and arcs[1] == [(2, 3, 3, 2.5)] and arcs[2] == []
    s2, f2, a2 = read_fst(os.path.join(G, "refmodel", "wordloop.fst"))
    assert s2 == 0 and len(f2) == 19 and sum(len(a) for a in a2) == 42


def test_real_decoder_over_the_product_lookup_functions_on_cpu():
    """Decoder::Decode calls pk_decodable_islastframe(-1) first (decoder.cc:49) and
    pk_decodable_loglikelihood per arc (decoder.cc:252,276): both are host functions of libpk_mi355.so
    and need no GPU.  The reference's 3-state test graph, T = 2 frames, every assignment of which
    first arc wins."""
    L = pk.lib()
    am = L.pk_mi355_am_create()          # no tid2pdf: transition-id == pdf index (include/pk_mi355.h)
    try:
        fst_path = os.path.join(G, "testinput.fst")
        fst = read_fst(fst_path)
        for seed in range(8):
            ll = (np.random.default_rng(seed).standard_normal((2, 4)) * 2).astype(np.float32)
            d = host_decodable(ll, am)
            words, weight = decode(C.byref(d), fst_path)
            cost, want, _ = viterbi(fst, ll, lambda t: t)
            assert words == want and words[1] == 3 and words[0] in (1, 2)
            # BestPath adds final() twice (decoder.cc:322,340-341): weight = best_cost + final
            assert abs(weight - (cost + 3.5)) < 1e-5 * max(1, abs(weight))
    finally:
        L.pk_mi355_am_destroy(am)


def build_example():
    pk.lib()
    libdir = os.path.dirname(pk.lib_path())
    out = os.path.join(REPO, "tests", "cpp", "process_example.bin")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "cpp", "process_example.cc"), "-o", out,
                           "-L", os.path.dirname(DECLIB), "-l:libpkref_decoder.so", "-L", libdir, "-l:libpk_mi355.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.dirname(DECLIB)])
    return out


def test_process_example_compiles_and_links():
    assert "pk_mi355" in subprocess.check_output([build_example(), "--link-only"], text=True)


def _parse(out):
    kv = dict(l.split(": ", 1) for l in out.splitlines() if ": " in l)
    return int(kv["frames"]), kv["hyp"].split(), float(kv["weight"]), float(kv["loglikelihood_per_frame"])


@pytest.mark.gpu
@pytest.mark.parametrize("wav", ["en-us-hello.wav", "en-us-cat.wav"])
@pytest.mark.parametrize("softmax", ["reference", "stable"])
def test_pk_process_shape_with_the_real_decoder(wav, softmax):
    """pk_process end to end (tests/cpp/process_example.cc): reference-written model files, the reference's
    WAVs, our word-loop graph; scoring on the GPU, search by the reference's decoder.  Expected: the same
    decoder over the ORACLE's log-likelihoods -- identical words; identical weight in reference-softmax
    mode (the log-likelihoods are then bit-identical), within 1e-4 * T otherwise."""
    from oracle import oracle as O
    from refmodel_text import DIR, load_text_model
    conf, fst_path, wav_path = os.path.join(DIR, "refmodel.conf"), os.path.join(DIR, "wordloop.fst"), os.path.join(G, wav)
    args = [build_example(), conf, wav_path, fst_path] + (["--reference-softmax"] if softmax == "reference" else [])
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0 and "process_example ok" in r.stdout, r.stdout + r.stderr
    for stage in ("Fbank:", "CMVN:", "NNET:", "decode:"):          # pocketkaldi.cc:194,206,218, decoder.cc:70
        assert stage in r.stderr
    T, hyp, weight, per_frame = _parse(r.stdout)

    layers, prior, Lc, Rc, tid2pdf, cmvn41 = load_text_model()
    wave = O.wav_read(wav_path)
    ref = O.Nnet(layers).am_compute(O.cmvn(cmvn41, O.Fbank().compute(wave)), prior, Lc, Rc, 0.1)
    ref = np.ascontiguousarray(ref, np.float32)
    assert T == ref.shape[0] == pk.num_frames(len(wave))
    am = pk.AcousticModel(layers, prior, Lc, Rc, tid2pdf)
    d = host_decodable(ref, am.handle)
    want_words, want_weight = decode(C.byref(d), fst_path)
    assert hyp == ["w%d" % w for w in want_words] and len(hyp) >= 1
    if softmax == "reference":
        assert weight == pytest.approx(want_weight, abs=0, rel=1e-7)
        assert per_frame == pytest.approx(np.float32(want_weight) / np.float32(T), rel=1e-6)
    else:
        assert abs(weight - want_weight) <= 1e-4 * T
    # and the search itself against an exhaustive Viterbi (the beam of 16 prunes nothing that matters here)
    cost, vit_words, s = viterbi(read_fst(fst_path), ref, lambda t: int(tid2pdf[t]))
    assert want_words == vit_words
