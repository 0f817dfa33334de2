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
