"""pocketkaldi_amd -- MI355X-native acoustic scoring behind pocketkaldi's own interfaces.

The product is the C-ABI shared library ``libpk_mi355.so`` (HIP kernels for gfx950 +
the reference-compatible ``pk_decodable_*`` functions; see ``include/pk_mi355.h``).
This package is the thin Python host side over that ABI used by the tests and the
benchmark; it mirrors the reference's class names and argument meaning:

    Fbank().compute(wave)                    fbank.h:47-53   Fbank::Compute
    CMVN(global_stats, raw).get_frames()     cmvn.h:17-26    CMVN::GetFrame, all frames
    AcousticModel(layers, prior, ...)        am.h:23-52      AcousticModel
    AcousticModel.propagate(x)               nnet.h:96       Nnet::Propagate
    Decodable(am, prob_scale, feats)         decodable.h:22-41
    BatchScorer(am, global_stats, ...)       pocketkaldi.cc:176-218 stages, batched

There is no CPU fallback: if the library is missing or no gfx950 device is usable,
every compute call raises ``PkError``.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

__all__ = ["PkError", "lib", "lib_path", "read_wav", "process_acoustic", "Fbank", "CMVN", "AcousticModel", "Decodable",
           "BatchScorer", "num_frames", "LINEAR", "RELU", "NORMALIZE", "SOFTMAX", "KINDS"]

LINEAR, RELU, NORMALIZE, SOFTMAX = 0, 1, 2, 3
KINDS = ("fbank", "cmvn", "gemm", "tail", "other")

_HERE = os.path.dirname(os.path.abspath(__file__))


class PkError(RuntimeError):
    pass


class pk_matrix_t(C.Structure):          # matrix.h:20-24
    _fields_ = [("ncol", C.c_int), ("nrow", C.c_int), ("data", C.POINTER(C.c_float))]


class pk_vector_t(C.Structure):          # vector.h:39-42
    _fields_ = [("dim", C.c_int), ("data", C.POINTER(C.c_float))]


class pk_decodable_t(C.Structure):       # decodable.h:15-18
    _fields_ = [("log_prob", pk_matrix_t), ("am", C.c_void_p)]


_lib = None

# every symbol include/pk_mi355.h declares (tests check the library exports all of them)
EXPORTS = [
    "pk_mi355_last_error", "pk_mi355_set_device", "pk_decodable_init", "pk_decodable_destroy",
    "pk_decodable_loglikelihood", "pk_decodable_islastframe", "pk_mi355_am_create",
    "pk_mi355_am_destroy", "pk_mi355_am_add_linear", "pk_mi355_am_add_layer",
    "pk_mi355_am_finalize", "pk_mi355_am_set_precision", "pk_mi355_am_precision", "pk_mi355_am_set_softmax", "pk_mi355_am_softmax", "pk_mi355_am_read", "pk_mi355_load", "pk_mi355_am_num_pdfs", "pk_mi355_am_input_dim",
    "pk_mi355_am_transition_to_pdf", "pk_mi355_am_blob_device_ptr", "pk_mi355_am_blob_bytes",
    "pk_mi355_nnet_propagate", "pk_mi355_num_frames", "pk_mi355_fbank_compute",
    "pk_mi355_cmvn_apply", "pk_mi355_batch_create", "pk_mi355_batch_destroy",
    "pk_mi355_batch_set_waves", "pk_mi355_batch_set_waves_i16", "pk_mi355_batch_set_waves_device",
    "pk_mi355_batch_score", "pk_mi355_batch_synchronize", "pk_mi355_batch_num_utts",
    "pk_mi355_batch_num_frames", "pk_mi355_batch_total_frames", "pk_mi355_batch_loglik_device",
    "pk_mi355_batch_fetch", "pk_mi355_batch_fetch_all", "pk_mi355_batch_fetch_fbank", "pk_mi355_batch_fetch_cmvn", "pk_mi355_test_logf",
    "pk_mi355_test_srfft512", "pk_mi355_am_broadcast", "pk_mi355_am_broadcast_from",
    "pk_mi355_batch_gather_loglik", "pk_mi355_device_malloc", "pk_mi355_device_free", "pk_mi355_memcpy",
    "pk_mi355_host_malloc", "pk_mi355_host_free",
    "pk_mi355_batch_stream", "pk_mi355_batch_enable_timing", "pk_mi355_batch_get_timing",
    "pk_mi355_am_flops_per_frame", "pk_mi355_16kpcm_read", "pk_mi355_process_acoustic",
    "pk_mi355_device_count", "pk_mi355_version",
    "pk_mi355_am_get_exponents", "pk_mi355_am_set_input_exponents", "pk_mi355_am_calibrate", "pk_mi355_batch_calibrate",
]


def lib_path():
    """The product library: always the in-tree build."""
    return os.path.join(_HERE, "libpk_mi355.so")


def lib():
    """Load libpk_mi355.so, (re)building it in-tree first whenever the sources it was built from
    differ from the sources in the tree (content hash, build.py).  A stale library is never
    loaded silently: where no compiler exists the load fails instead."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if _build._stale():
        if not _build.have_compiler():
            raise PkError("libpk_mi355.so is %s and there is no hipcc here to build it"
                          % ("stale (sources changed since it was built)" if os.path.exists(path) else "missing"))
        try:
            _build.build()
        except Exception as e:
            raise PkError("libpk_mi355.so cannot be built: %s" % e)
    L = C.CDLL(path)
    f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.pk_mi355_last_error.restype = C.c_char_p
    L.pk_mi355_version.restype = C.c_char_p
    L.pk_mi355_set_device.argtypes = [C.c_int]
    L.pk_decodable_init.restype = None
    L.pk_decodable_init.argtypes = [C.POINTER(pk_decodable_t), C.c_void_p, C.c_float,
                                    C.POINTER(pk_matrix_t)]
    L.pk_decodable_destroy.restype = None
    L.pk_decodable_destroy.argtypes = [C.POINTER(pk_decodable_t)]
    L.pk_decodable_loglikelihood.restype = C.c_float
    L.pk_decodable_loglikelihood.argtypes = [C.POINTER(pk_decodable_t), C.c_int, C.c_int]
    L.pk_decodable_islastframe.restype = C.c_bool
    L.pk_decodable_islastframe.argtypes = [C.POINTER(pk_decodable_t), C.c_int]
    L.pk_mi355_am_create.restype = C.c_void_p
    L.pk_mi355_am_destroy.restype = None
    L.pk_mi355_am_destroy.argtypes = [C.c_void_p]
    L.pk_mi355_am_add_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p]
    L.pk_mi355_am_add_layer.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_am_set_precision.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_am_precision.argtypes = [C.c_void_p]
    L.pk_mi355_am_finalize.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int]
    L.pk_mi355_am_read.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                   C.c_int]
    L.pk_mi355_am_set_softmax.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_am_softmax.argtypes = [C.c_void_p]
    L.pk_mi355_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), f32p]
    L.pk_mi355_am_num_pdfs.argtypes = [C.c_void_p]
    L.pk_mi355_am_input_dim.argtypes = [C.c_void_p]
    L.pk_mi355_am_transition_to_pdf.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_am_blob_device_ptr.restype = C.c_void_p
    L.pk_mi355_am_blob_device_ptr.argtypes = [C.c_void_p]
    L.pk_mi355_am_blob_bytes.restype = C.c_size_t
    L.pk_mi355_am_blob_bytes.argtypes = [C.c_void_p]
    L.pk_mi355_am_flops_per_frame.restype = C.c_double
    L.pk_mi355_am_flops_per_frame.argtypes = [C.c_void_p]
    L.pk_mi355_nnet_propagate.argtypes = [C.c_void_p, C.POINTER(pk_matrix_t), C.POINTER(pk_matrix_t)]
    L.pk_mi355_num_frames.argtypes = [C.c_int]
    L.pk_mi355_fbank_compute.argtypes = [C.POINTER(pk_vector_t), C.POINTER(pk_matrix_t)]
    L.pk_mi355_cmvn_apply.argtypes = [C.POINTER(pk_vector_t), C.POINTER(pk_matrix_t),
                                      C.POINTER(pk_matrix_t)]
    L.pk_mi355_batch_create.restype = C.c_void_p
    L.pk_mi355_batch_create.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int64]
    L.pk_mi355_batch_destroy.restype = None
    L.pk_mi355_batch_destroy.argtypes = [C.c_void_p]
    L.pk_mi355_batch_set_waves.argtypes = [C.c_void_p, C.POINTER(pk_vector_t), C.c_int]
    L.pk_mi355_batch_set_waves_i16.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.POINTER(C.c_int),
                                               C.c_int]
    L.pk_mi355_batch_set_waves_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.pk_mi355_batch_score.argtypes = [C.c_void_p, C.c_float, C.c_int]
    L.pk_mi355_batch_synchronize.argtypes = [C.c_void_p]
    L.pk_mi355_batch_num_utts.argtypes = [C.c_void_p]
    L.pk_mi355_batch_num_frames.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_batch_total_frames.restype = C.c_int64
    L.pk_mi355_batch_total_frames.argtypes = [C.c_void_p]
    L.pk_mi355_batch_loglik_device.restype = C.c_void_p
    L.pk_mi355_batch_loglik_device.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_batch_fetch.argtypes = [C.c_void_p, C.c_int, C.POINTER(pk_decodable_t)]
    L.pk_mi355_batch_fetch_all.argtypes = [C.c_void_p, C.POINTER(pk_decodable_t), C.c_int, C.c_int]
    L.pk_mi355_batch_fetch_fbank.argtypes = [C.c_void_p, C.c_int, f32p]
    L.pk_mi355_batch_fetch_cmvn.argtypes = [C.c_void_p, C.c_int, f32p]
    L.pk_mi355_test_logf.argtypes = [f32p, C.c_int, f32p]
    L.pk_mi355_test_srfft512.argtypes = [f32p, C.c_int, f32p]
    L.pk_mi355_am_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.pk_mi355_am_broadcast_from.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.pk_mi355_batch_gather_loglik.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.pk_mi355_device_malloc.restype = C.c_void_p
    L.pk_mi355_device_malloc.argtypes = [C.c_size_t]
    L.pk_mi355_device_free.restype = None
    L.pk_mi355_device_free.argtypes = [C.c_void_p]
    L.pk_mi355_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.pk_mi355_host_malloc.restype = C.c_void_p
    L.pk_mi355_host_malloc.argtypes = [C.c_size_t]
    L.pk_mi355_host_free.restype = None
    L.pk_mi355_host_free.argtypes = [C.c_void_p]
    L.pk_mi355_batch_stream.restype = C.c_void_p
    L.pk_mi355_batch_stream.argtypes = [C.c_void_p]
    L.pk_mi355_batch_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.pk_mi355_batch_get_timing.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_int)]
    L.pk_mi355_16kpcm_read.argtypes = [C.c_char_p, C.POINTER(pk_vector_t)]
    L.pk_mi355_process_acoustic.argtypes = [C.c_void_p, C.POINTER(pk_vector_t), C.POINTER(pk_vector_t), C.c_float,
                                            C.POINTER(pk_decodable_t), C.c_int]
    L.pk_mi355_am_get_exponents.argtypes = [C.c_void_p, i32p, i32p, C.c_int]
    L.pk_mi355_am_set_input_exponents.argtypes = [C.c_void_p, i32p, C.c_int]
    L.pk_mi355_am_calibrate.argtypes = [C.c_void_p, C.POINTER(pk_matrix_t)]
    L.pk_mi355_batch_calibrate.argtypes = [C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise PkError(lib().pk_mi355_last_error().decode() or "pk_mi355 error %d" % rc)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _as_matrix(a):
    """[T][D] frame-major numpy array -> pk_matrix_t{ncol=T, nrow=D} (borrowed)."""
    m = pk_matrix_t()
    m.ncol, m.nrow = a.shape
    m.data = _fp(a)
    return m


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _take_matrix(m):
    """Copy a library-allocated pk_matrix_t into numpy [ncol][nrow] and free it."""
    if m.ncol == 0 or m.nrow == 0 or not m.data:
        out = np.zeros((m.ncol, m.nrow), dtype=np.float32)
    else:
        out = np.ctypeslib.as_array(m.data, shape=(m.ncol, m.nrow)).copy()
    _libc.free(m.data)
    return out


def set_device(device):
    _check(lib().pk_mi355_set_device(int(device)))


def num_frames(num_samples):
    """Fbank::CalcNumFrames (fbank.cc:35-42)."""
    return lib().pk_mi355_num_frames(int(num_samples))


def read_wav(path):
    """pk_16kpcm_read (pcm_reader.cc:45-220) -> float32 samples (unscaled)."""
    v = pk_vector_t(0, None)
    _check(lib().pk_mi355_16kpcm_read(path.encode(), C.byref(v)))
    out = np.ctypeslib.as_array(v.data, shape=(max(v.dim, 1),))[:v.dim].copy()
    _libc.free(v.data)
    return out


def process_acoustic(am, global_stats, wave, prob_scale=0.1, verbose=False):
    """Stages 1-3 of pk_process (pocketkaldi.cc:186-218) fused on the device -> Decodable."""
    g, w = _f32(global_stats), _f32(wave)
    gv = pk_vector_t(g.shape[0], _fp(g))
    wv = pk_vector_t(w.shape[0], _fp(w) if w.size else None)
    d = pk_decodable_t()
    _check(lib().pk_mi355_process_acoustic(am.handle, C.byref(gv), C.byref(wv), float(prob_scale),
                                           C.byref(d), 1 if verbose else 0))
    return Decodable._from_struct(d, am)


class Fbank:
    """fbank.h:47-53.  compute(wave) -> [T][40] log-mel features."""

    def compute(self, wave):
        wave = _f32(wave)
        v = pk_vector_t(wave.shape[0], _fp(wave))
        m = pk_matrix_t(0, 0, None)
        _check(lib().pk_mi355_fbank_compute(C.byref(v), C.byref(m)))
        return _take_matrix(m)


class CMVN:
    """cmvn.h:17-26.  Sliding-window mean normalisation with a global prior."""

    def __init__(self, global_stats, raw_feats):
        self.global_stats = _f32(global_stats)
        self.raw = _f32(raw_feats)

    def get_frames(self):
        g = pk_vector_t(self.global_stats.shape[0], _fp(self.global_stats))
        raw = _as_matrix(self.raw) if self.raw.size else pk_matrix_t(0, 40, None)
        out = pk_matrix_t(0, 0, None)
        _check(lib().pk_mi355_cmvn_apply(C.byref(g), C.byref(raw), C.byref(out)))
        return _take_matrix(out)


class AcousticModel:
    """am.h:23-52 + nnet.h:88-104.

    layers: list of ("linear", W[out][in], b[out]) | ("relu",) | ("normalize",) | ("softmax",)
    prior:  pdf prior probabilities (the log is taken at load, am.cc:43)
    """
    _KIND = {"relu": RELU, "normalize": NORMALIZE, "softmax": SOFTMAX}

    PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}

    def __init__(self, layers=None, prior=None, left_context=0, right_context=0, tid2pdf=None,
                 num_pdfs=None, precision="f32"):
        L = lib()
        self._h = L.pk_mi355_am_create()
        _check(L.pk_mi355_am_set_precision(self._h, self.PRECISIONS[precision]))
        if layers is not None:
            for l in layers:
                if l[0] == "linear":
                    W, b = _f32(l[1]), _f32(l[2])
                    _check(L.pk_mi355_am_add_linear(self._h, W.shape[1], W.shape[0], _fp(W), _fp(b)))
                else:
                    _check(L.pk_mi355_am_add_layer(self._h, self._KIND[l[0]]))
            pr = None if prior is None else _f32(prior)
            n = int(num_pdfs) if num_pdfs is not None else (0 if pr is None else pr.shape[0])
            tid = None if tid2pdf is None else np.ascontiguousarray(tid2pdf, dtype=np.int32)
            _check(L.pk_mi355_am_finalize(
                self._h, None if pr is None else _fp(pr), n, left_context, right_context,
                None if tid is None else tid.ctypes.data_as(C.POINTER(C.c_int32)),
                0 if tid is None else tid.shape[0]))

    @classmethod
    def read(cls, nnet_path, prior_path, tid2pdf_path, left_context, right_context, num_pdfs,
             precision="f32"):
        """AcousticModel::Read (am.cc:23-63) from the converted model files."""
        self = cls(precision=precision)
        _check(lib().pk_mi355_am_read(self._h, nnet_path.encode(), prior_path.encode(),
                                      None if tid2pdf_path is None else tid2pdf_path.encode(),
                                      left_context, right_context, num_pdfs))
        return self

    @classmethod
    def load(cls, config_path, precision="f32"):
        """pk_load's model part (pocketkaldi.cc:72-144): the reference's key = value model file.
        Returns (model, cmvn_global_stats[41])."""
        self = cls.__new__(cls)
        h = C.c_void_p()
        stats = np.zeros(41, dtype=np.float32)
        self._h = None
        _check(lib().pk_mi355_load(config_path.encode(), cls.PRECISIONS[precision], C.byref(h), _fp(stats)))
        self._h = h.value
        return self, stats

    def set_softmax(self, mode):
        """"stable" (default): overflow-safe log-softmax; "reference": the reference's operations one by one."""
        _check(lib().pk_mi355_am_set_softmax(self._h, {"stable": 0, "reference": 1}[mode]))
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().pk_mi355_am_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def num_pdfs(self):
        return lib().pk_mi355_am_num_pdfs(self._h)

    def input_dim(self):
        return lib().pk_mi355_am_input_dim(self._h)

    def transition_id_to_pdf_id(self, tid):
        return lib().pk_mi355_am_transition_to_pdf(self._h, int(tid))

    def flops_per_frame(self):
        return lib().pk_mi355_am_flops_per_frame(self._h)

    def blob(self):
        """(device pointer, bytes) of the packed weights, for the RCCL broadcast."""
        return lib().pk_mi355_am_blob_device_ptr(self._h), lib().pk_mi355_am_blob_bytes(self._h)

    def broadcast(self, rccl_comm, root=0, stream=None, src=None):
        """pk_mi355_am_broadcast[_from]: one ncclBroadcast into this model's blob over the caller's ncclComm_t
        (an integer handle); with src, the root sends that model's blob instead of its own."""
        _check(lib().pk_mi355_am_broadcast_from(self._h, None if src is None else src.handle, C.c_void_p(rccl_comm),
                                                int(root), None if stream is None else C.c_void_p(stream)))

    def exponents(self):
        """f16x3 / f16: (w_exp, x_exp) per affine layer -- the exact power-of-two operand scalings."""
        w = np.zeros(64, dtype=np.int32)
        x = np.zeros(64, dtype=np.int32)
        n = lib().pk_mi355_am_get_exponents(self._h, w.ctypes.data_as(C.POINTER(C.c_int32)),
                                            x.ctypes.data_as(C.POINTER(C.c_int32)), 64)
        if n < 0:
            _check(n)
        return w[:n].copy(), x[:n].copy()

    def set_input_exponents(self, x_exp):
        x = np.ascontiguousarray(x_exp, dtype=np.int32)
        _check(lib().pk_mi355_am_set_input_exponents(self._h, x.ctypes.data_as(C.POINTER(C.c_int32)), x.shape[0]))

    def calibrate(self, feats):
        """f16x3 / f16: set every operand's exponent from CMVN'd features [T][feat_dim] (pk_mi355_am_calibrate)."""
        feats = _f32(feats)
        m = _as_matrix(feats)
        _check(lib().pk_mi355_am_calibrate(self._h, C.byref(m)))
        return self

    def propagate(self, x):
        """Nnet::Propagate (nnet.cc:149-163): x [T][in_dim] -> [T][out_dim]."""
        x = _f32(x)
        m_in = _as_matrix(x) if x.size else pk_matrix_t(0, x.shape[1], None)
        m_out = pk_matrix_t(0, 0, None)
        _check(lib().pk_mi355_nnet_propagate(self._h, C.byref(m_in), C.byref(m_out)))
        return _take_matrix(m_out)


def device_logf(x):
    """The front-end kernel's logf on an array (parity-test hook)."""
    x = _f32(x).ravel()
    out = np.empty_like(x)
    _check(lib().pk_mi355_test_logf(_fp(x), x.shape[0], _fp(out)))
    return out


def device_srfft512(frames):
    """The front-end kernel's 512-point real FFT on [n][512] frames (parity-test hook)."""
    x = _f32(frames).reshape(-1, 512)
    out = np.empty_like(x)
    _check(lib().pk_mi355_test_srfft512(_fp(x), x.shape[0], _fp(out)))
    return out


class _Pinned:
    def __init__(self, nbytes):
        self.ptr = lib().pk_mi355_host_malloc(max(int(nbytes), 1))
        if not self.ptr:
            raise PkError(lib().pk_mi355_last_error().decode())

    def __del__(self):
        try:
            lib().pk_mi355_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_i16(n):
    """An int16 numpy array of n samples in page-locked host memory (pk_mi355_host_malloc)."""
    owner = _Pinned(2 * int(n))
    buf = (C.c_int16 * int(n)).from_address(owner.ptr)
    arr = np.frombuffer(buf, dtype=np.int16)
    arr = arr.view(_OwnedArray)
    arr._owner = owner
    return arr


class _OwnedArray(np.ndarray):
    _owner = None

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class Decodable:
    """decodable.h:15-41 over the C ABI (what decoder.cc consumes)."""

    def __init__(self, am, prob_scale, feats):
        feats = _f32(feats)
        self._d = pk_decodable_t()
        self._am = am
        m = _as_matrix(feats)
        lib().pk_decodable_init(C.byref(self._d), am.handle, float(prob_scale), C.byref(m))
        if feats.shape[0] > 0 and self._d.log_prob.ncol != feats.shape[0]:
            raise PkError(lib().pk_mi355_last_error().decode() or "pk_decodable_init failed")

    @classmethod
    def _from_struct(cls, d, am):
        self = cls.__new__(cls)
        self._d, self._am = d, am
        return self

    def loglikelihood(self, frame, trans_id):
        return lib().pk_decodable_loglikelihood(C.byref(self._d), int(frame), int(trans_id))

    def is_last_frame(self, frame):
        return bool(lib().pk_decodable_islastframe(C.byref(self._d), int(frame)))

    def log_prob(self):
        """The host matrix as numpy [T][num_pdfs] (a copy)."""
        lp = self._d.log_prob
        if lp.ncol == 0:
            return np.zeros((0, lp.nrow), dtype=np.float32)
        return np.ctypeslib.as_array(lp.data, shape=(lp.ncol, lp.nrow)).copy()

    def destroy(self):
        if self._d is not None:
            lib().pk_decodable_destroy(C.byref(self._d))
            self._d = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class BatchScorer:
    """Many utterances in flight: PCM -> fbank -> CMVN -> nnet -> log-likelihoods, all in HBM."""

    def __init__(self, am, global_stats, max_utts, max_total_samples):
        g = _f32(global_stats)
        if g.shape != (41,):
            raise PkError("global_stats must have 41 entries")
        self._am = am
        self._h = lib().pk_mi355_batch_create(am.handle, _fp(g), int(max_utts), int(max_total_samples))
        if not self._h:
            raise PkError(lib().pk_mi355_last_error().decode())
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            lib().pk_mi355_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_waves(self, waves):
        waves = [_f32(w) for w in waves]
        arr = (pk_vector_t * max(len(waves), 1))()
        for i, w in enumerate(waves):
            arr[i].dim, arr[i].data = w.shape[0], _fp(w)
        _check(lib().pk_mi355_batch_set_waves(self._h, arr, len(waves)))

    def set_waves_i16(self, waves):
        waves = [np.ascontiguousarray(w, dtype=np.int16) for w in waves]
        cat = np.concatenate(waves) if waves else np.zeros(0, np.int16)
        ns = (C.c_int * max(len(waves), 1))(*[w.shape[0] for w in waves])
        _check(lib().pk_mi355_batch_set_waves_i16(self._h, cat.ctypes.data_as(C.POINTER(C.c_int16)), ns,
                                                  len(waves)))

    def set_waves_i16_raw(self, samples, num_samples):
        """Concatenated int16 samples (e.g. a pinned_i16() array) + per-utterance counts; no host copy."""
        ns = (C.c_int * max(len(num_samples), 1))(*[int(n) for n in num_samples])
        self._keep = samples
        _check(lib().pk_mi355_batch_set_waves_i16(self._h, samples.ctypes.data_as(C.POINTER(C.c_int16)), ns,
                                                  len(num_samples)))

    def set_waves_device(self, device_ptr, num_samples, keep_alive=None):
        """PCM already in HBM: device_ptr -> concatenated float samples."""
        ns = (C.c_int * max(len(num_samples), 1))(*[int(n) for n in num_samples])
        self._keep = keep_alive
        _check(lib().pk_mi355_batch_set_waves_device(self._h, C.c_void_p(device_ptr), ns, len(num_samples)))

    def score(self, prob_scale=0.1, sync=True):
        _check(lib().pk_mi355_batch_score(self._h, float(prob_scale), 1 if sync else 0))

    def synchronize(self):
        _check(lib().pk_mi355_batch_synchronize(self._h))

    def calibrate(self):
        """f16x3 / f16: calibrate the model's operand exponents on the utterances currently set."""
        _check(lib().pk_mi355_batch_calibrate(self._h))

    def num_utts(self):
        return lib().pk_mi355_batch_num_utts(self._h)

    def num_frames(self, utt):
        return lib().pk_mi355_batch_num_frames(self._h, utt)

    def total_frames(self):
        return lib().pk_mi355_batch_total_frames(self._h)

    def stream(self):
        return lib().pk_mi355_batch_stream(self._h)

    def loglik_device(self, utt):
        return lib().pk_mi355_batch_loglik_device(self._h, utt)

    def fetch(self, utt):
        d = pk_decodable_t()
        _check(lib().pk_mi355_batch_fetch(self._h, utt, C.byref(d)))
        return Decodable._from_struct(d, self._am)

    def fetch_all(self, sync=True):
        """Every utterance's decodable from one transfer into the batch's page-locked arena (views)."""
        n = self.num_utts()
        arr = (pk_decodable_t * max(n, 1))()
        _check(lib().pk_mi355_batch_fetch_all(self._h, arr, n, 1 if sync else 0))
        out = [Decodable._from_struct(arr[u], self._am) for u in range(n)]
        for d in out:
            d._owner = self          # a view must not outlive the batch whose arena it points into
        return out

    def fetch_fbank(self, utt):
        out = np.zeros((self.num_frames(utt), 40), dtype=np.float32)
        _check(lib().pk_mi355_batch_fetch_fbank(self._h, utt, _fp(out)))
        return out

    def fetch_cmvn(self, utt):
        out = np.zeros((self.num_frames(utt), 40), dtype=np.float32)
        _check(lib().pk_mi355_batch_fetch_cmvn(self._h, utt, _fp(out)))
        return out

    def gather_loglik(self, utt, d_frames, d_trans_ids, n, d_out):
        """Device-side loglikelihood(frame, trans_id) for n pairs (all pointers are device pointers)."""
        _check(lib().pk_mi355_batch_gather_loglik(self._h, utt, C.c_void_p(d_frames), C.c_void_p(d_trans_ids),
                                                  int(n), C.c_void_p(d_out)))

    def enable_timing(self, on=True):
        _check(lib().pk_mi355_batch_enable_timing(self._h, 1 if on else 0))

    def timing(self):
        ms = (C.c_float * 5)()
        n = (C.c_int * 5)()
        _check(lib().pk_mi355_batch_get_timing(self._h, ms, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KINDS)}
