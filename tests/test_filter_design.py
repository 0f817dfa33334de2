"""Host logic without a GPU: the product's C++ filter designer (mi_filter_design) against the Python oracle
restatement of the reference designer, for every filter_type_t, plus the anchors that pin the oracle itself."""
import json
import os

import numpy as np
import pytest

from oracle import filter_design as fd

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "filter_anchors.json")))

ALL_TYPES = list(range(len(fd.FILTER_TYPES)))
GRID = [(slope, freq, gain, q) for slope in (1, 2, 3, 4) for freq in (100.0, 1000.0, 10000.0)
        for gain in (0.5, 2.0) for q in (0.0, 0.7)]


def oracle_design(t, slope, freq, freq2, gain, q, sr):
    mode, casc, bq = fd.design(fd.Params(t, slope, freq, freq2, gain, q), sr)
    return mode, casc, bq


@pytest.mark.parametrize("t", ALL_TYPES)
def test_product_designer_matches_oracle_bit_for_bit(mi, t):
    """G3 of SURVEY.md 8c: coefficient table for every filter_type_t over a parameter grid."""
    for slope, freq, gain, q in GRID:
        for sr in (48000, 44100):
            freq2 = freq * 2.5
            mode_o, casc_o, bq_o = oracle_design(t, slope, freq, freq2, gain, q, sr)
            mode_p, casc_p, bq_p = mi.design_filter(t, slope, freq, freq2, gain, q, sr)
            what = "%s slope=%d f=%g g=%g q=%g sr=%d" % (fd.FILTER_TYPES[t], slope, freq, gain, q, sr)
            assert mode_p == mode_o, what
            assert bq_p.shape == bq_o.shape, what
            assert len(casc_p) == len(casc_o), what
            np.testing.assert_array_equal(bq_p.view(np.uint32), bq_o.view(np.uint32), err_msg=what)
            co = np.array([[c["t"], c["b"]] for c in casc_o], np.float32).reshape(-1, 2, 3)
            np.testing.assert_array_equal(casc_p.view(np.uint32), co.view(np.uint32), err_msg=what)


def test_anchor_k_weighting_table(mi):
    """ITU-R BS.1770 coefficients quoted in the reference (Filter.cpp:2103-2111), denominator signs negated
    as Filter.cpp:2261-2262 / ButterworthFilter.cpp:163-164 document."""
    _, _, bq = mi.design_filter(fd.FLT_K_WEIGHTED, sample_rate=48000)
    itu = GOLD["k_weighting_48k"]
    np.testing.assert_allclose(bq[0], [itu["shelf"]["b0"], itu["shelf"]["b1"], itu["shelf"]["b2"],
                                       -itu["shelf"]["a1"], -itu["shelf"]["a2"]], rtol=0, atol=2e-6)
    np.testing.assert_allclose(bq[1], [1.0, -2.0, 1.0, -itu["hipass"]["a1"], -itu["hipass"]["a2"]], rtol=0, atol=2e-6)


def test_anchor_readme_hishelf(mi):
    """C1: the two sections of FLT_BT_BWC_HISHELF slope 2 @1 kHz +6 dB @48 kHz (values recorded by the survey
    probe of the reference sources, SURVEY.md Appendix C, and the independent double-precision derivation
    of SURVEY.md 8c)."""
    gain = float(np.float32(np.exp(np.float32(6.0) * np.float32(np.log(10.0)) * np.float32(0.05))))
    _, _, bq = mi.design_filter(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, gain, 0.0, 48000)
    np.testing.assert_allclose(bq, np.array(GOLD["c1_hishelf_sections"], np.float32), rtol=0, atol=3e-7)


def test_lrx_slope4_is_eight_sections_and_minus_6db(mi):
    """The C2 parameterisation (reference's own anti-alias setup, Sample.cpp:1230-1235)."""
    _, _, bq = mi.design_filter(fd.FLT_BT_LRX_LOPASS, 4, 1000.0, 1000.0, 1.0, 0.0, 48000)
    assert bq.shape == (8, 5)
    h = fd.freq_response(bq, [1e-3, 1000.0], 48000)
    assert abs(abs(h[0]) - 1.0) < 1e-4 and abs(abs(h[1]) - 0.5) < 1e-3          # Linkwitz-Riley: -6.02 dB at fc


@pytest.mark.parametrize("t", [t for t in ALL_TYPES if fd.FILTER_TYPES[t].startswith("FLT_BT_")])
def test_bilinear_sections_realise_the_analog_prototype(mi, t):
    """Property pin for Filter::bilinear_transform (Filter.cpp:2192-2267): the digital cascade evaluated on the
    unit circle equals the analog prototype at the pre-warped frequency -- checked through the product's own
    freq_chart (Filter.cpp:602-632) and an independent float64 evaluation of the sections."""
    f = np.array([20.0, 200.0, 1000.0, 5000.0, 15000.0], np.float32)
    for slope, gain, q in ((1, 2.0, 0.0), (3, 0.5, 0.5)):
        _, _, bq = mi.design_filter(t, slope, 1000.0, 3000.0, gain, q, 48000)
        chart = mi.filter_freq_chart(f, t, slope, 1000.0, 3000.0, gain, q, 48000)
        direct = fd.freq_response(bq, f, 48000)
        scale = max(1e-6, np.abs(direct).max())
        assert np.abs(chart - direct).max() <= 2e-4 * scale, fd.FILTER_TYPES[t]


def test_limit(mi):
    from importlib import import_module
    capi = import_module("lsp-dsp-units_amd.capi")
    import ctypes
    fp = capi.FilterParams(fd.FLT_BT_RLC_BELL, 1000, 30000.0, -5.0, 1.0, 0.0)
    mi.check(mi.lib.mi_filter_limit(ctypes.byref(fp), 48000))
    assert fp.nSlope == 128 and fp.fFreq == np.float32(0.49) * np.float32(48000) and fp.fFreq2 == 0.0
