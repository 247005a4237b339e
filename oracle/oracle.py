"""ctypes front for the CPU oracle (oracle/liboracle.so) and the partial
real-reference build (oracle/_ref/libpkref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libpkref.so")

NUM_BINS = 40
FRAME_LENGTH = 400
FRAME_SHIFT = 160
FFT_SIZE = 512

LINEAR, RELU, NORMALIZE, SOFTMAX = 0, 1, 2, 3

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB) or (
            os.path.isdir("/root/reference/src") and not os.path.exists(_REF)):
        subprocess.check_call(["make", "-s", "-C", _HERE])


class _Nnet(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("layers", C.c_void_p)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.pko_num_frames.restype = C.c_int
        L.pko_num_frames.argtypes = [C.c_int]
        L.pko_srfft_init.argtypes = [C.c_void_p, C.c_int]
        L.pko_srfft_forward.argtypes = [C.c_void_p, _f32p]
        L.pko_fbank_init.argtypes = [C.c_void_p]
        L.pko_fbank_compute.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.pko_fbank_frame.argtypes = [C.c_void_p, _f32p, _f32p, C.c_void_p]
        L.pko_cmvn.argtypes = [_f32p, _f32p, C.c_int, _f32p]
        L.pko_splice.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.pko_logf_array.argtypes = [_f32p, C.c_int, _f32p]
        for name in ("pko_sgemm", "pko_sgemm_naive"):
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p,
                                         C.c_int, _f32p, C.c_int]
        L.pko_nnet_init.argtypes = [C.POINTER(_Nnet)]
        L.pko_nnet_free.argtypes = [C.POINTER(_Nnet)]
        L.pko_nnet_add_linear.argtypes = [C.POINTER(_Nnet), C.c_int, C.c_int, _f32p, _f32p]
        L.pko_nnet_add_simple.argtypes = [C.POINTER(_Nnet), C.c_int]
        L.pko_nnet_propagate.argtypes = [C.POINTER(_Nnet), _f32p, C.c_int, C.c_int, _f32p,
                                         C.c_int]
        L.pko_nnet_output_dim.argtypes = [C.POINTER(_Nnet), C.c_int]
        L.pko_am_compute.argtypes = [C.POINTER(_Nnet), _f32p, C.c_int, C.c_int, C.c_int,
                                     _f32p, C.c_int, C.c_int, C.c_float, _f32p]
        L.pko_relu.argtypes = [_f32p, C.c_int64]
        L.pko_normalize_rows.argtypes = [_f32p, C.c_int, C.c_int]
        L.pko_softmax_rows.argtypes = [_f32p, C.c_int, C.c_int]
        L.pko_wav_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float))]
        L.pko_read_vec_f32.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)),
                                       C.POINTER(C.c_int)]
        L.pko_nnet_read.argtypes = [C.POINTER(_Nnet), C.c_char_p]
        _lib = L
    return _lib


def have_ref():
    build()
    return os.path.exists(_REF)


def ref():
    """The partial REAL reference (srfft.cc + gemm.cc + gemm_haswell.cc)."""
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(_REF)
        R.pkref_sgemm.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int,
                                  _f32p, C.c_int]
        R.pkref_srfft_forward.argtypes = [_f32p, C.c_int]
        _ref = R
    return _ref


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


# --------------------------------------------------------------------------- FFT

class Srfft:
    def __init__(self, n_real=FFT_SIZE):
        self._buf = C.create_string_buffer(1024)
        if lib().pko_srfft_init(self._buf, n_real) != 0:
            raise ValueError("bad FFT size")
        self.n_real = n_real

    def forward(self, x):
        y = np.ascontiguousarray(x, dtype=np.float32).copy()
        assert y.shape == (self.n_real,)
        lib().pko_srfft_forward(self._buf, y)
        return y


def ref_srfft(x):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    ref().pkref_srfft_forward(y, y.shape[0])
    return y


# --------------------------------------------------------------------------- fbank / cmvn

class Fbank:
    def __init__(self):
        self._buf = C.create_string_buffer(64 * 1024)
        if lib().pko_fbank_init(self._buf) != 0:
            raise RuntimeError("pko_fbank_init failed")

    def compute(self, wave):
        wave = np.ascontiguousarray(wave, dtype=np.float32)
        T = lib().pko_num_frames(wave.shape[0])
        out = np.zeros((T, NUM_BINS), dtype=np.float32)
        if T:
            lib().pko_fbank_compute(self._buf, wave, wave.shape[0], out)
        return out

    def frame(self, samples400):
        s = np.ascontiguousarray(samples400, dtype=np.float32)
        out = np.zeros(NUM_BINS, dtype=np.float32)
        spec = np.zeros(FFT_SIZE, dtype=np.float32)
        lib().pko_fbank_frame(self._buf, s, out, spec.ctypes.data_as(C.c_void_p))
        return out, spec


def num_frames(n):
    return lib().pko_num_frames(int(n))


def cmvn(global_stats, raw):
    g = np.ascontiguousarray(global_stats, dtype=np.float32)
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    assert g.shape == (NUM_BINS + 1,) and raw.shape[1] == NUM_BINS
    out = np.zeros_like(raw)
    if raw.shape[0]:
        lib().pko_cmvn(g, raw, raw.shape[0], out)
    return out


def logf(x):
    """libm logf per element (vector.cc:334-339)."""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    out = np.empty_like(x)
    lib().pko_logf_array(x, x.shape[0], out)
    return out


def splice(feats, left, right):
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T, D = feats.shape
    out = np.zeros((T, (left + right + 1) * D), dtype=np.float32)
    lib().pko_splice(feats, T, D, left, right, out)
    return out


# --------------------------------------------------------------------------- GEMM

def _gemm(fn, A, B):
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    m, k = A.shape
    k2, n = B.shape
    assert k == k2
    Cm = np.zeros((m, n), dtype=np.float32)
    fn(m, n, k, A, max(k, 1), B, max(n, 1), Cm, max(n, 1))
    return Cm


def sgemm(A, B):
    return _gemm(lib().pko_sgemm, A, B)


def sgemm_naive(A, B):
    return _gemm(lib().pko_sgemm_naive, A, B)


def ref_sgemm(A, B):
    return _gemm(ref().pkref_sgemm, A, B)


# --------------------------------------------------------------------------- nnet / am

class Nnet:
    """layers: list of ("linear", W[out][in], b[out]) | ("relu",) | ("normalize",) | ("softmax",)"""

    _KIND = {"relu": RELU, "normalize": NORMALIZE, "softmax": SOFTMAX}

    def __init__(self, layers=()):
        self._nn = _Nnet()
        lib().pko_nnet_init(C.byref(self._nn))
        for l in layers:
            if l[0] == "linear":
                W = np.ascontiguousarray(l[1], dtype=np.float32)
                b = np.ascontiguousarray(l[2], dtype=np.float32)
                lib().pko_nnet_add_linear(C.byref(self._nn), W.shape[1], W.shape[0], W, b)
            else:
                lib().pko_nnet_add_simple(C.byref(self._nn), self._KIND[l[0]])

    @classmethod
    def read(cls, path):
        self = cls()
        if lib().pko_nnet_read(C.byref(self._nn), path.encode()) != 0:
            raise IOError("pko_nnet_read failed: %s" % path)
        return self

    def __del__(self):
        try:
            lib().pko_nnet_free(C.byref(self._nn))
        except Exception:
            pass

    def output_dim(self, in_dim):
        return lib().pko_nnet_output_dim(C.byref(self._nn), in_dim)

    def propagate(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        T, D = x.shape
        od = self.output_dim(D)
        if od < 0:
            raise ValueError("dimension mismatch")
        out = np.zeros((T, od), dtype=np.float32)
        rc = lib().pko_nnet_propagate(C.byref(self._nn), x, T, D, out, od)
        if rc != od:
            raise RuntimeError("pko_nnet_propagate rc=%d" % rc)
        return out

    def am_compute(self, feats, prior, left, right, prob_scale):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        prior = np.ascontiguousarray(prior, dtype=np.float32)
        T, D = feats.shape
        out = np.zeros((T, prior.shape[0]), dtype=np.float32)
        rc = lib().pko_am_compute(C.byref(self._nn), prior, prior.shape[0], left, right,
                                  feats, T, D, float(prob_scale), out)
        if rc != 0:
            raise RuntimeError("pko_am_compute rc=%d" % rc)
        return out


def relu(x):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().pko_relu(y, y.size)
    return y


def normalize_rows(x):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().pko_normalize_rows(y, y.shape[0], y.shape[1])
    return y


def softmax_rows(x):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().pko_softmax_rows(y, y.shape[0], y.shape[1])
    return y


# --------------------------------------------------------------------------- files

def wav_read(path):
    p = C.POINTER(C.c_float)()
    n = lib().pko_wav_read(path.encode(), C.byref(p))
    if n < 0:
        raise IOError("pko_wav_read(%s) = %d" % (path, n))
    arr = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    _libc.free(p)
    return arr


def read_vec(path):
    p = C.POINTER(C.c_float)()
    d = C.c_int()
    if lib().pko_read_vec_f32(path.encode(), C.byref(p), C.byref(d)) != 0:
        raise IOError("pko_read_vec_f32(%s)" % path)
    arr = np.ctypeslib.as_array(p, shape=(max(d.value, 1),))[:d.value].copy()
    _libc.free(p)
    return arr
