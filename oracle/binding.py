"""ctypes binding of liborc.so (the C oracle)."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_size_t, c_uint, c_void_p

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_here, "liborc.so")


def build(force=False):
    srcs = [os.path.join(_here, f) for f in os.listdir(_here) if f.endswith(".c")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return
    subprocess.check_call(["make", "-s", "-C", _here])


build()
_lib = ctypes.CDLL(LIB_PATH)
_fp = POINTER(c_float)


def _f(a):
    return a.ctypes.data_as(_fp)


_lib.orc_biquad_cascade.argtypes = [_fp, _fp, c_size_t, _fp, _fp, c_size_t]
_lib.orc_biquad_bank.argtypes = [_fp, _fp, c_size_t, c_size_t, c_size_t, c_size_t, _fp, _fp,
                                 POINTER(c_uint), c_size_t]
_lib.orc_biquad_impulse_response.argtypes = [_fp, c_size_t, _fp, _fp, c_size_t]


def biquad_cascade(x, coef, state=None):
    """One channel through len(coef) sections; returns (y, new_state)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1, 5)
    ns = coef.shape[0]
    st = np.zeros((max(ns, 1), 2), np.float32) if state is None else np.array(state, dtype=np.float32, copy=True)
    y = np.empty_like(x)
    _lib.orc_biquad_cascade(_f(y), _f(x), x.size, _f(coef) if ns else None, _f(st), ns)
    return y, st


def biquad_bank(x, coef, nsec, state):
    """[channels][n] block through per-channel cascades. coef [C][S][5], state [C][S][2] updated in place."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    C, n = x.shape
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    nsec = np.ascontiguousarray(nsec, dtype=np.uint32)
    assert state.dtype == np.float32 and state.flags.c_contiguous
    y = np.empty_like(x)
    _lib.orc_biquad_bank(_f(y), _f(x), C, n, n, n, _f(coef), _f(state),
                         nsec.ctypes.data_as(POINTER(c_uint)), coef.shape[1])
    return y


def biquad_impulse_response(n, coef, state):
    coef = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1, 5)
    out = np.empty(n, np.float32)
    _lib.orc_biquad_impulse_response(_f(out), n, _f(coef), _f(state), coef.shape[0])
    return out


def biquad_cascade_f64(x, coef):
    """Same recurrence in float64 (zero start state): the round-off yardstick used by the
    parity tests to tell implementation error from the float32 noise floor of the recursion."""
    from scipy.signal import lfilter  # direct evaluation, double precision
    y = np.asarray(x, dtype=np.float64)
    for b0, b1, b2, a1, a2 in np.asarray(coef, dtype=np.float64).reshape(-1, 5):
        y = lfilter([b0, b1, b2], [1.0, -a1, -a2], y)
    return y


_lib.orc_dyn_biquad_cascade.argtypes = [_fp, _fp, c_size_t, _fp, _fp, c_size_t]
_lib.orc_dyn_biquad_cascade_f64.argtypes = [POINTER(ctypes.c_double), _fp, c_size_t, _fp, POINTER(ctypes.c_double), c_size_t]


def dyn_biquad_cascade(x, coef, state=None):
    """Time-varying sections: coef [ns][n][5], one coefficient set per section and sample; returns (y, state)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    ns = coef.shape[0] if coef.size else 0
    assert ns == 0 or coef.shape[1:] == (x.size, 5)
    st = np.zeros((max(ns, 1), 2), np.float32) if state is None else np.array(state, dtype=np.float32, copy=True)
    y = np.empty_like(x)
    _lib.orc_dyn_biquad_cascade(_f(y), _f(x), x.size, _f(coef) if ns else None, _f(st), ns)
    return y, st


def dyn_biquad_cascade_f64(x, coef, state=None):
    """The same recurrence in double arithmetic on the float32 coefficients; returns (y, state)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    ns = coef.shape[0] if coef.size else 0
    st = np.zeros((max(ns, 1), 2), np.float64) if state is None else np.array(state, dtype=np.float64, copy=True)
    y = np.empty(x.size, np.float64)
    dp = POINTER(ctypes.c_double)
    _lib.orc_dyn_biquad_cascade_f64(y.ctypes.data_as(dp), _f(x), x.size, _f(coef) if ns else None, st.ctypes.data_as(dp), ns)
    return y, st


def biquad_cascade_f64_state(x, coef, state=None):
    """The recurrence in float64 with a carried state: returns (y, state[ns][2]).  scipy's lfilter is the same
    transposed direct form II (zi = {d0, d1}), so the state of the reference's sections maps onto it one to one;
    this is what lets a test follow exact arithmetic across re-designs that keep the filter memory."""
    from scipy.signal import lfilter
    coef = np.asarray(coef, dtype=np.float64).reshape(-1, 5)
    st = np.zeros((coef.shape[0], 2)) if state is None else np.array(state, dtype=np.float64, copy=True)
    y = np.asarray(x, dtype=np.float64)
    for j, (b0, b1, b2, a1, a2) in enumerate(coef):
        y, st[j] = lfilter([b0, b1, b2], [1.0, -a1, -a2], y, zi=st[j])
    return y, st


# ---- FFT / fast convolution primitives (fft_oracle.c) ---------------------------------------------------
_lib.orc_packed_direct_fft.argtypes = [_fp, _fp, c_size_t]
_lib.orc_packed_reverse_fft.argtypes = [_fp, _fp, c_size_t]
_lib.orc_fastconv_parse.argtypes = [_fp, _fp, c_size_t]
_lib.orc_fastconv_apply.argtypes = [_fp, _fp, _fp, _fp, c_size_t]
_lib.orc_fastconv_parse_apply.argtypes = [_fp, _fp, _fp, _fp, c_size_t]
_lib.orc_convolve.argtypes = [_fp, _fp, _fp, c_size_t, c_size_t]
_lib.orc_convolve_f64.argtypes = [POINTER(ctypes.c_double), _fp, _fp, c_size_t, c_size_t]


def packed_direct_fft(x, rank):
    """x: 2*2^rank interleaved float32 -> spectrum (unnormalised, e^{-jwn})."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.size == 2 << rank
    y = np.empty_like(x)
    _lib.orc_packed_direct_fft(_f(y), _f(x), rank)
    return y


def packed_reverse_fft(x, rank):
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.size == 2 << rank
    y = np.empty_like(x)
    _lib.orc_packed_reverse_fft(_f(y), _f(x), rank)
    return y


def fastconv_parse(src, rank):
    src = np.ascontiguousarray(src, dtype=np.float32)
    assert src.size >= 1 << (rank - 1)
    img = np.empty(2 << rank, np.float32)
    _lib.orc_fastconv_parse(_f(img), _f(src), rank)
    return img


def fastconv_parse_apply(dst, conv_image, src, rank):
    """dst[0..2^rank) += conv(src[0..2^(rank-1)), ir); dst is modified in place."""
    tmp = np.empty(2 << rank, np.float32)
    src = np.ascontiguousarray(src, dtype=np.float32)
    assert dst.dtype == np.float32 and dst.flags.c_contiguous and dst.size >= 1 << rank
    _lib.orc_fastconv_parse_apply(_f(dst), _f(tmp), _f(conv_image), _f(src), rank)


def convolve(src, conv, count=None):
    """Naive float32 dst[i+j] += src[i]*conv[j] over the first `count` source samples."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    conv = np.ascontiguousarray(conv, dtype=np.float32)
    count = src.size if count is None else count
    dst = np.zeros(count + conv.size, np.float32)
    _lib.orc_convolve(_f(dst), _f(src), _f(conv), conv.size, count)
    return dst


def convolve_f64(src, conv, count=None):
    src = np.ascontiguousarray(src, dtype=np.float32)
    conv = np.ascontiguousarray(conv, dtype=np.float32)
    count = src.size if count is None else count
    dst = np.zeros(count + conv.size, np.float64)
    _lib.orc_convolve_f64(dst.ctypes.data_as(POINTER(ctypes.c_double)), _f(src), _f(conv), conv.size, count)
    return dst


# ---- Convolver (convolver_oracle.c) -------------------------------------------------------------------
_lib.orc_convolver_create.restype = c_void_p
_lib.orc_convolver_create.argtypes = [_fp, c_size_t, c_size_t, c_float]
_lib.orc_convolver_destroy.argtypes = [c_void_p]
_lib.orc_convolver_process.argtypes = [c_void_p, _fp, _fp, c_size_t]
_lib.orc_convolver_data_size.restype = c_size_t
_lib.orc_convolver_data_size.argtypes = [c_void_p]
_lib.orc_convolver_rank.restype = c_size_t
_lib.orc_convolver_rank.argtypes = [c_void_p]


class Convolver:
    """lsp::dspu::Convolver restated (util/Convolver.h:35-114)."""

    def __init__(self, data, rank, phase=0.0):
        data = np.ascontiguousarray(data, dtype=np.float32)
        self._h = _lib.orc_convolver_create(_f(data), data.size, rank, phase)

    def process(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        dst = np.empty_like(src)
        _lib.orc_convolver_process(self._h, _f(dst), _f(src), src.size)
        return dst

    def process_chunked(self, src, step):
        src = np.ascontiguousarray(src, dtype=np.float32)
        out = np.empty_like(src)
        for i in range(0, src.size, step):
            out[i:i + step] = self.process(src[i:i + step])
        return out

    @property
    def data_size(self):
        return _lib.orc_convolver_data_size(self._h)

    @property
    def rank(self):
        return _lib.orc_convolver_rank(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.orc_convolver_destroy(self._h)
            self._h = None
