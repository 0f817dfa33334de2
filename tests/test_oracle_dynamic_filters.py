"""lsp::dspu::DynamicFilters: pins of the oracle and of the product's host side, no GPU.

The reference holds no test or vector for this unit and its three arithmetic primitives are in the absent lsp-dsp-lib
(oracle/dynamic_filters.py header), so the restatement is pinned by what the unit must do:
  * constant coefficients: the time-varying recurrence IS FilterBank::process's, bit for bit;
  * constant gain, bilinear types: the digital sections realise the analog cascades at the pre-warped frequency, i.e. the
    spectrum of the sections equals DynamicFilters::freq_chart (complex), for every type;
  * what the gain MEANS per type: bells peak at it, shelves reach it, passes scale by it, gain 1 is transparent for the
    equaliser types;
  * the product's C++ builders (mi_dynfilter_sections / _freq_chart) against the numpy restatement for every type."""
import numpy as np
import pytest

import oracle
from oracle import dynamic_filters as df
from oracle import filter_design as fd

SR = 48000
TYPES = [t for t in range(1, len(fd.FILTER_TYPES)) if df.cascade_count(t, 1) > 0]
BILINEAR = [t for t in TYPES if t & 1]


def _filt(t, slope=2, freq=1000.0, freq2=4000.0, q=0.5):
    d = df.DynamicFilters(1)
    d.set_sample_rate(SR)
    d.set_params(0, t, slope, freq, freq2, 1.0, q)
    d.set_filter_active(0, True)
    return d


def test_constant_coefficients_are_the_static_recurrence_bit_for_bit():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(5000).astype(np.float32)
    for t in (fd.FLT_BT_RLC_BELL, fd.FLT_BT_LRX_LOPASS, fd.FLT_MT_BWC_HISHELF):
        d = _filt(t, slope=3)
        coef = d.coefficients(0, np.full(x.size, 1.7, np.float32))
        assert np.all(coef == coef[:, :1, :])                       # the same sections for every sample
        y_dyn, st_dyn = oracle.dyn_biquad_cascade(x, coef)
        y_sta, st_sta = oracle.biquad_cascade(x, coef[:, 0, :])
        np.testing.assert_array_equal(y_dyn, y_sta)
        np.testing.assert_array_equal(st_dyn, st_sta)
        # and process() carries the memory from call to call
        a = d.process(0, x[:2000], np.full(2000, 1.7, np.float32))
        b = d.process(0, x[2000:], np.full(3000, 1.7, np.float32))
        np.testing.assert_array_equal(np.concatenate([a, b]), y_sta)


@pytest.mark.parametrize("t", BILINEAR)
@pytest.mark.parametrize("gain", [0.35, 1.0, 2.5])
def test_sections_realise_the_analog_cascades(t, gain):
    """Bilinear types: H of the digital sections on the unit circle == freq_chart (the analog cascades at the pre-warped
    frequency, DynamicFilters.cpp:1897-1930), complex, for slopes 1..3."""
    f = np.array([30.0, 200.0, 1000.0, 3000.0, 9000.0, 18000.0], np.float32)
    for slope in (1, 2, 3):
        d = _filt(t, slope)
        coef = d.coefficients(0, np.array([gain], np.float32))[:, 0, :]
        h = fd.freq_response(coef, f, SR)
        chart = d.freq_chart(0, f, gain)
        scale = max(np.abs(chart).max(), 1e-3)
        assert np.abs(h - chart).max() <= 2e-4 * scale, (fd.FILTER_TYPES[t], slope)


def _mag(t, gain, f, slope=2, freq=1000.0, freq2=4000.0, q=0.0):
    d = _filt(t, slope, freq, freq2, q)
    coef = d.coefficients(0, np.array([gain], np.float32))[:, 0, :]
    return np.abs(fd.freq_response(coef, np.atleast_1d(np.float64(f)), SR))


# Types whose dynamic builder (DynamicFilters.cpp:625-1738) and static designer (Filter.cpp:722-1487) are the same analog
# design: at a constant gain g the dynamic filter IS the static Filter with fGain = g.  The two are separate pieces of the
# reference, restated separately (oracle/filter_design.py -- pinned bit for bit to the product's designer and through it to
# the README / BS.1770 anchors -- and oracle/dynamic_filters.py), so their agreement pins the builders AND the per-sample
# transforms (bilinear and matched) of the dynamic path.  The remaining types are different designs in the reference
# itself (the dynamic LRX shelves / bells / ladders have their own formulas, RLC_BANDPASS another normalisation).
SAME_AS_STATIC = [t for t in TYPES if not any(k in fd.FILTER_TYPES[t] for k in
                  ("RLC_BANDPASS", "LRX_LOSHELF", "LRX_HISHELF", "LRX_BELL", "LRX_LADDER", "LRX_BANDPASS"))]


@pytest.mark.parametrize("t", SAME_AS_STATIC)
def test_constant_gain_is_the_static_filter(t):
    f = np.array([50.0, 300.0, 1000.0, 2500.0, 8000.0, 16000.0])
    for slope in (1, 2, 3):
        for g in (0.4, 2.5):
            for q in (0.0, 0.7):
                _, _, static = fd.design(fd.Params(t, slope, 1000.0, 4000.0, g, q), SR)
                d = _filt(t, slope, 1000.0, 4000.0, q)
                dyn = d.coefficients(0, np.array([g], np.float32))[:, 0, :]
                hs, hd = fd.freq_response(static, f, SR), fd.freq_response(dyn, f, SR)
                assert np.abs(hs - hd).max() <= 2e-4 * max(np.abs(hs).max(), 1e-9), (fd.FILTER_TYPES[t], slope, g, q)


def test_what_the_gain_means():
    """Every family, incl. the ones with their own dynamic design: gain 1 leaves the equaliser types transparent, the
    passes scale by the gain, the notch keeps its zero."""
    g = 2.5
    lo, mid, hi = 20.0, 1000.0, 20000.0
    for pre in ("BT", "MT"):
        T = lambda name: getattr(fd, "FLT_%s_%s" % (pre, name))
        tol = 0.02 if pre == "BT" else 0.12                      # matched-Z keeps the response only well below Nyquist
        assert abs(_mag(T("AMPLIFIER"), g, mid)[0] - g) < 1e-5
        assert abs(_mag(T("RLC_BELL"), g, mid)[0] - g) < tol * g
        for fam in ("RLC", "BWC", "LRX"):
            assert abs(_mag(T(fam + "_BELL"), g, lo)[0] - 1.0) < tol, (pre, fam, "bell skirt")
            assert _mag(T(fam + "_BELL"), g, mid)[0] > 1.2 and _mag(T(fam + "_BELL"), 1.0 / g, mid)[0] < 0.85, (pre, fam, "bell")
            assert abs(_mag(T(fam + "_LOSHELF"), g, lo)[0] - g) < tol * g, (pre, fam, "loshelf")
            assert abs(_mag(T(fam + "_HISHELF"), g, lo)[0] - 1.0) < tol, (pre, fam, "hishelf bottom")
            assert abs(_mag(T(fam + "_LOPASS"), g, lo)[0] - g) < tol * g, (pre, fam, "lopass")
            if pre == "BT":
                assert abs(_mag(T(fam + "_HISHELF"), g, hi)[0] - g) < 0.05 * g, (pre, fam, "hishelf top")
                assert abs(_mag(T(fam + "_HIPASS"), g, hi)[0] - g) < 0.05 * g, (pre, fam, "hipass")
            for name in ("_BELL", "_LOSHELF", "_HISHELF", "_LADDERPASS", "_LADDERREJ"):
                m = _mag(T(fam + name), 1.0, [lo, 300.0, mid, 5000.0])
                assert np.abs(m - 1.0).max() < (2e-3 if pre == "BT" else 0.05), (pre, fam, name, m)
        assert _mag(T("RLC_NOTCH"), g, mid)[0] < 1e-3 * g and abs(_mag(T("RLC_NOTCH"), g, lo)[0] - g) < tol * g


def test_set_params_orders_and_transforms_the_second_frequency():
    d = _filt(fd.FLT_BT_LRX_BANDPASS, 1, 4000.0, 500.0)         # f2 < f: swapped (DynamicFilters.cpp:160-166)
    p = d.get_params(0)
    assert p["fFreq"] == np.float32(500.0)
    nf = np.float32(np.pi) / np.float32(SR)
    assert p["fFreq2"] == np.float32(np.tan(np.float32(500.0) * nf, dtype=np.float32) / np.tan(np.float32(4000.0) * nf, dtype=np.float32))
    d = _filt(fd.FLT_MT_RLC_BANDPASS, 1, 500.0, 4000.0)
    assert d.get_params(0)["fFreq2"] == np.float32(500.0) / np.float32(4000.0)
    d = df.DynamicFilters(2)
    assert not d.set_filter_active(2, True) and d.set_filter_active(1, False) and d.active[1]      # sets true whatever is asked
    x = np.arange(8, dtype=np.float32)
    np.testing.assert_array_equal(d.process(0, x, np.ones(8, np.float32)), x)                      # inactive / FLT_NONE: a copy


@pytest.mark.parametrize("t", TYPES)
def test_product_builders_match_the_oracle(mi, t):
    """mi_dynfilter_sections / mi_dynfilter_freq_chart (C++, glibc math) against the numpy restatement: the two differ by
    the last bits of their libm only."""
    f = np.array([25.0, 400.0, 1000.0, 2500.0, 12000.0], np.float32)
    for slope, freq, freq2, q, gain in ((1, 1000.0, 4000.0, 0.0, 0.5), (2, 300.0, 2000.0, 0.7, 3.0), (3, 5000.0, 800.0, 1.5, 1.0),
                                        (4, 120.0, 9000.0, 0.3, 0.25)):
        if not (t & 1) and freq < 200.0:
            continue            # matched-Z at 120 Hz: the float normalisation next to z = 1 is only good to 2e-3 in either libm
        d = _filt(t, slope, freq, freq2, q)
        ref = d.coefficients(0, np.array([gain], np.float32))[:, 0, :]
        got = mi.dynfilter_sections(t, slope, freq, freq2, q, gain, SR)
        what = "%s slope %d f %g f2 %g q %g gain %g" % (fd.FILTER_TYPES[t], slope, freq, freq2, q, gain)
        assert got.shape == ref.shape, what
        # bilinear: last bits of expf / sinf / tanf.  Matched-Z: its amplitude normalisation evaluates k - k (e^a + e^b) z + k e^(a+b) z^2
        # next to z = 1 in float (Filter.cpp:2369-2411), which magnifies those last bits to 1e-4 of the numerator
        rtol = 3e-5 if (t & 1) else 1e-3
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=3e-6 * max(1.0, np.abs(ref).max()), err_msg=what)
        chart_ref = d.freq_chart(0, f, gain)
        chart = mi.dynfilter_freq_chart(f, t, slope, freq, freq2, q, gain, SR)
        np.testing.assert_allclose(chart, chart_ref, rtol=0, atol=2e-4 * max(np.abs(chart_ref).max(), 1e-3), err_msg=what)


def test_unsupported_types_are_refused(mi):
    with pytest.raises(ValueError):
        _filt(fd.FLT_BT_RLC_ENVELOPE)
    assert mi.dynfilter_sections(fd.FLT_BT_RLC_ENVELOPE, 2, 1000.0, 1000.0, 0.0, 1.0, SR).shape == (0, 5)
    assert mi.dynfilter_sections(fd.FLT_NONE, 2, 1000.0, 1000.0, 0.0, 1.0, SR).shape == (0, 5)
