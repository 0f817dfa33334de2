"""Oracle spectral units pinned by the reference's tests, and the product's host-side windows against them."""
import numpy as np
import pytest

import oracle
from oracle import spectral as sp


def test_reference_utest_spectral_proc():
    """src/test/utest/util/spectral_proc.cpp:37-67: rank 8 of max 14, no callback, 440 Hz sine, output ==
    input delayed by latency() within 1e-5."""
    n = 8192
    w = np.float32(2 * np.pi * 440.0 / 48000.0)
    src = np.sin((w * np.arange(n, dtype=np.float32)).astype(np.float32)).astype(np.float32)
    p = sp.SpectralProcessor(14)
    p.set_phase(0.0)
    p.set_rank(8)
    dst = p.process(src)
    lat = p.latency()
    assert lat == 256
    assert np.abs(dst[lat:] - src[:n - lat]).max() <= 1e-5


def test_spectral_proc_with_callback_scaling():
    """A x0.5 spectrum callback: exercises packed_direct/reverse_fft scaling."""
    rng = np.random.default_rng(3)
    src = rng.standard_normal(4096).astype(np.float32)
    p = sp.SpectralProcessor(10)
    p.bind(lambda spec, rank: spec * np.float32(0.5))
    dst = p.process(src)
    lat = p.latency()
    assert np.abs(dst[lat:] - 0.5 * src[:-lat]).max() <= 1e-5


def test_analyzer_sine_peak():
    """Analytic pin: a full-scale sine at bin 64 of a 1024-point Hann analysis reads N/4 * amplitude at that bin."""
    sr, rank = 48000, 10
    a = sp.Analyzer(2, rank, sr, 1.0, 0)
    a.configure(sample_rate=sr, rate=20.0, rank=rank, window_name="hann", reactivity=0.0001, shift=1.0)
    n = 1 << rank
    f = 64 * sr / n
    t = np.arange(3 * 2400, dtype=np.float64)
    x = np.stack([np.sin(2 * np.pi * f * t / sr), 0.5 * np.sin(2 * np.pi * f * t / sr)]).astype(np.float32)
    a.process(x)
    assert a.period == 2400 and a.step == 1200
    peak = a.data[:, :a.csize].argmax(axis=1)
    assert list(peak) == [64, 64]
    assert abs(a.data[0, 64] - n / 4) / (n / 4) < 2e-3
    assert abs(a.data[1, 64] - n / 8) / (n / 8) < 2e-3


@pytest.mark.parametrize("name", sorted(sp.WINDOW_IDS))
@pytest.mark.parametrize("n", [1, 2, 7, 64, 4096])
def test_product_windows_match_oracle(mi, name, n):
    """Host logic, no GPU: mi_window (C++) against the Python restatement of windows.cpp."""
    if n == 1 and name in ("hann", "hamming", "blackman", "nuttall", "blackman_nuttall", "blackman_harris"):
        pytest.skip("n == 1 divides by zero in the reference (2 pi / (n - 1)); value is inf/nan by construction")
    got = mi.make_window(n, sp.WINDOW_IDS[name])
    ref = sp.window(n, name)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)


def test_all_window_types_are_finite_and_bounded(mi):
    for t in range(21):
        w = mi.make_window(512, t)
        assert np.all(np.isfinite(w)) and w.max() <= 1.0 + 1e-5 and w.min() >= -0.1, t


def test_general_window_families(mi):
    """windows::*_general (misc/windows.h:71-155): with the reference's constants each family reproduces its named window
    bit for bit (windows.cpp:91-99,152-160,179-181,199-213,232-234,300-302,315-317,332-334,351-353,398-400); with other
    parameters the closed forms of windows.cpp:70-89,141-150,164-177,185-197,217-230,286-298,306-313,321-330,338-349,
    371-396 hold; wrong parameter counts and parameter-free windows are refused."""
    n = 257
    # windows::window_t (misc/windows.h:36-62)
    ids = {k: i for i, k in enumerate(["hann", "hamming", "blackman", "lanczos", "gaussian", "poisson", "parzen", "tukey", "welch",
                                       "nuttall", "blackman_nuttall", "blackman_harris", "hann_poisson", "bartlett_hann",
                                       "bartlett_fejer", "triangular", "rectangular", "flat_top", "cosine", "sqr_cosine", "cubic"])}
    assert all(ids[k] == v for k, v in sp.WINDOW_IDS.items())
    named = {"hann": [0.5, 0.5], "hamming": [0.54, 0.46], "blackman": [0.16],
             "nuttall": [0.355768, 0.487396, 0.144232, 0.012604], "blackman_nuttall": [0.3635819, 0.4891775, 0.1365995, 0.0106411],
             "blackman_harris": [0.35875, 0.48829, 0.14128, 0.01168], "flat_top": [1.0, 1.93, 1.29, 0.388, 0.028],
             "triangular": [0], "bartlett_fejer": [-1], "gaussian": [0.4], "poisson": [n * 0.5],
             "bartlett_hann": [0.62, 0.48, 0.38], "hann_poisson": [2.0], "tukey": [0.5]}
    for name, q in named.items():
        np.testing.assert_array_equal(mi.make_window_general(n, ids[name], q), mi.make_window(n, ids[name]), err_msg=name)
    i = np.arange(n, dtype=np.float64)
    c = (n - 1) * 0.5
    np.testing.assert_allclose(mi.make_window_general(n, ids["hamming"], [0.6, 0.4]), 0.6 - 0.4 * np.cos(2 * np.pi * i / (n - 1)), atol=2e-6)
    np.testing.assert_allclose(mi.make_window_general(n, ids["gaussian"], [0.25]), np.exp(-0.5 * ((i - c) / (c * 0.25)) ** 2), atol=2e-6)
    np.testing.assert_allclose(mi.make_window_general(n, ids["poisson"], [40.0]), np.exp(-np.abs(i - c) / 40.0), atol=2e-6)
    np.testing.assert_allclose(mi.make_window_general(n, ids["triangular"], [1]), 1.0 - np.abs((i - c) * 2.0 / (n + 1)), atol=2e-6)
    np.testing.assert_array_equal(mi.make_window_general(n, ids["tukey"], [0.0]), np.ones(n, np.float32))      # a == 0: rectangular
    assert np.all(mi.make_window_general(n, ids["gaussian"], [0.4]) == mi.make_window(n, ids["gaussian"]))
    for bad in (("hann", [0.5]), ("cosine", [1.0]), ("tukey", [0.5, 0.5])):
        with pytest.raises(mi.MiError):
            mi.make_window_general(n, ids[bad[0]], bad[1])
