"""Host side of the spectral envelopes (mi_envelope_*: envelope::noise_lin / reverse_noise_lin of misc/envelope.h) against
the oracle's restatement of basic_noise_lin.  No GPU needed."""
import ctypes
import importlib
import math

import numpy as np
import pytest

from oracle import spectral as osp

FP = ctypes.POINTER(ctypes.c_float)
VIOLET, BLUE, WHITE, PINK, BROWN, MINUS_4_5, PLUS_4_5 = range(7)
P45 = np.float32(4.5 / (20.0 * float(np.float32(math.log10(2.0)))))
SLOPE = {VIOLET: 1.0, BLUE: 0.5, WHITE: 0.0, PINK: -0.5, BROWN: -1.0, MINUS_4_5: -float(P45), PLUS_4_5: float(P45)}


@pytest.fixture(scope="module")
def mi():
    return importlib.import_module("lsp-dsp-units_amd")


@pytest.mark.parametrize("kind", range(7))
@pytest.mark.parametrize("n", [1, 2, 513, 2049])
def test_linear_grid_envelopes(mi, kind, n):
    first, last, center = 0.0, 24000.0, 100.0               # the Analyzer's call (Analyzer.cpp:276-277)
    for name, sign in (("mi_envelope_noise_lin", 1.0), ("mi_envelope_reverse_noise_lin", -1.0)):
        got = np.full(n, -1.0, np.float32)
        mi.check(getattr(mi.lib, name)(got.ctypes.data_as(FP), first, last, center, n, kind))
        if kind == WHITE:
            want = np.ones(n, np.float32)
        else:
            want = osp.reverse_noise_lin(first, last, center, n, sign * SLOPE[kind])
        # powf of the host libm against numpy's float32 power: equal to the last bit or one ulp apart
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
    if n > 2 and kind == PINK:
        # -3 dB per octave in power = amplitude ~ f^-0.5: 400 Hz is half of 100 Hz... relative to the centre
        g = np.empty(n, np.float32)
        mi.check(mi.lib.mi_envelope_noise_lin(g.ctypes.data_as(FP), first, last, center, n, PINK))
        f = np.linspace(first, last, n)
        i = int(np.argmin(np.abs(f - 400.0)))
        assert abs(g[i] - (f[i] / center) ** -0.5) < 1e-5


@pytest.mark.parametrize("kind", range(7))
@pytest.mark.parametrize("n", [1, 2, 300])
def test_logarithmic_grid_and_list_envelopes(mi, kind, n):
    """envelope::noise_log / reverse_noise_log (envelope.cpp:152-268) and noise_list / reverse_noise_list (:272-344)
    against the oracle's restatement, plus the definition itself: (f / center)^k on the grid."""
    first, last, center = 20.0, 20000.0, 1000.0
    freqs = np.geomspace(first, last, n).astype(np.float32) if n > 1 else np.array([440.0], np.float32)
    for reverse, sign in ((0, 1.0), (1, -1.0)):
        got = np.full(n, -1.0, np.float32)
        mi.check(mi.lib.mi_envelope_noise_log(got.ctypes.data_as(FP), first, last, center, n, kind, reverse))
        want = np.ones(n, np.float32) if kind == WHITE else osp.noise_log(first, last, center, n, sign * SLOPE[kind])
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)    # expf and powf of two libms
        if n > 2 and kind != WHITE:
            grid = first * (last / first) ** (np.arange(n) / (n - 1))
            np.testing.assert_allclose(got, (grid / center) ** (sign * SLOPE[kind]), rtol=2e-5)
        got = np.full(n, -1.0, np.float32)
        mi.check(mi.lib.mi_envelope_noise_list(got.ctypes.data_as(FP), freqs.ctypes.data_as(FP), center, n, kind, reverse))
        want = np.ones(n, np.float32) if kind == WHITE else osp.noise_list(freqs, center, sign * SLOPE[kind])
        np.testing.assert_allclose(got, want, rtol=4e-7, atol=0)


def test_bad_arguments(mi):
    g = np.empty(4, np.float32)
    with pytest.raises(mi.MiError):
        mi.check(mi.lib.mi_envelope_noise_lin(g.ctypes.data_as(FP), 0.0, 1.0, 1.0, 4, 9))
    with pytest.raises(mi.MiError):
        mi.check(mi.lib.mi_envelope_reverse_noise_lin(None, 0.0, 1.0, 1.0, 4, PINK))
    with pytest.raises(mi.MiError):
        mi.check(mi.lib.mi_envelope_noise_log(g.ctypes.data_as(FP), 1.0, 2.0, 1.0, 4, 9, 0))
    with pytest.raises(mi.MiError):
        mi.check(mi.lib.mi_envelope_noise_list(g.ctypes.data_as(FP), None, 1.0, 4, PINK, 0))
