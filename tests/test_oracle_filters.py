"""Pins of the STREAMING IIR oracle (oracle/biquad_oracle.c) that do not refer to the oracle itself.

The reference holds no test with expected IIR output, and its arithmetic core (lsp-dsp-lib's
biquad_process_x*) is not in the tree.  What the reference does document is the transfer function every filter
must realise: Filter::freq_chart evaluates it from the analog prototype (src/main/filters/Filter.cpp:500-696).
A streaming recurrence with a wrong sign convention, state update, section order or an extra sample of delay does
not have that spectrum, so: spectrum of the oracle's impulse response == freq_chart, for every filter type."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import filter_design as fd

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "filter_anchors.json")))
SR = 48000
N = 1 << 16          # impulse response length: every case below has decayed far below float32 round-off by then

CASES = [(t, slope, freq, gain, q)
         for t in range(1, len(fd.FILTER_TYPES))
         for (slope, freq, gain, q) in ((1, 1000.0, 2.0, 0.7), (2, 3000.0, 0.5, 0.0), (3, 300.0, 1.5, 0.4))]


@pytest.mark.parametrize("t,slope,freq,gain,q", CASES)
def test_impulse_response_spectrum_is_the_reference_transfer_function(t, slope, freq, gain, q):
    p = fd.Params(t, slope, freq, freq * 2.5, gain, q)
    mode, casc, bq = fd.design(p, SR)
    if len(bq) == 0:
        pytest.skip("%s designs no section for these parameters" % fd.FILTER_TYPES[t])
    h = oracle.biquad_impulse_response(N, bq, np.zeros((len(bq), 2), np.float32))
    assert np.all(np.isfinite(h))
    if np.abs(h[-1024:]).max() > 1e-6 * max(np.abs(h).max(), 1e-30):
        pytest.skip("%s with these parameters rings beyond %d samples (pole next to the unit circle): its spectrum "
                    "cannot be read off a truncated response; the other parameter sets cover the type" % (fd.FILTER_TYPES[t], N))
    spec = np.fft.rfft(h.astype(np.float64))
    bins = np.array([8, 27, 137, 683, 1365, 2731, 6827, 13653, 20480, 27307])          # 6 Hz .. 20 kHz
    f = bins * (SR / float(N))
    chart, cmode = fd.freq_chart(p, SR, f.astype(np.float32))
    assert cmode == mode
    scale = max(np.abs(chart).max(), 1e-6)
    if mode == fd.FM_MATCHED:
        # matched-Z keeps poles and zeros, not the response: the chart is the analog curve (Filter.cpp:633-660), equal
        # to the digital one only well below Nyquist
        sel = f < 1500.0
        assert np.abs(np.abs(spec[bins][sel]) - np.abs(chart[sel])).max() <= 0.05 * scale, fd.FILTER_TYPES[t]
    elif mode == fd.FM_APO:
        # digital designs (Filter.cpp:661-696): the chart sums t0 + t1 e^{jw} + t2 e^{2jw} (positive powers,
        # Filter.cpp:405-498), so its phase is not that of the causal filter; the magnitudes are the same function
        # (above 50 Hz: next to DC the chart's float32 cos terms cancel and the chart itself is off by 1e-3 of the peak)
        sel = f > 50.0
        assert np.abs(np.abs(spec[bins][sel]) - np.abs(chart[sel])).max() <= 1e-3 * scale, fd.FILTER_TYPES[t]
    else:
        # bilinear: the chart is the analog prototype at the pre-warped frequency (Filter.cpp:602-632) -- complex equality
        # (1e-3 of the peak: float32 coefficients of a 300 Hz slope-3 design sit 4e-4 from their prototype; a structural
        # error in the recurrence is an O(1) difference)
        assert np.abs(spec[bins] - chart.astype(np.complex128)).max() <= 1e-3 * scale, fd.FILTER_TYPES[t]


def test_readme_filter_impulse_response_head_recorded_from_the_reference():
    """First eight output samples of the README filter (FLT_BT_BWC_HISHELF slope 2, 1 kHz, +6 dB, README.md:176-184)
    as the reference's own Filter/FilterBank objects produced them in the survey probe."""
    gain = float(np.float32(np.exp(np.float32(6.0) * np.float32(np.log(10.0)) * np.float32(0.05))))
    _, _, bq = fd.design(fd.Params(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, gain, 0.0), SR)
    h = oracle.biquad_impulse_response(8, bq, np.zeros((2, 2), np.float32))
    np.testing.assert_allclose(h, np.array(GOLD["c1_impulse_response_head"], np.float32), rtol=0, atol=4e-7)


def test_float64_state_oracle_matches_the_float32_one():
    """biquad_cascade_f64_state (scipy lfilter with zi) carries the same {d0, d1} as the C oracle."""
    rng = np.random.default_rng(5)
    _, _, bq = fd.design(fd.Params(fd.FLT_BT_RLC_BELL, 2, 2000.0, 0.0, 2.0, 1.0), SR)
    x = rng.standard_normal(3000).astype(np.float32)
    y32a, st32 = oracle.biquad_cascade(x[:1000], bq)
    y32b, st32b = oracle.biquad_cascade(x[1000:], bq, st32)
    y64a, st64 = oracle.biquad_cascade_f64_state(x[:1000], bq)
    y64b, st64b = oracle.biquad_cascade_f64_state(x[1000:], bq, st64)
    whole = oracle.biquad_cascade_f64(x, bq)
    np.testing.assert_allclose(np.concatenate([y64a, y64b]), whole, rtol=0, atol=1e-12)
    assert np.abs(np.concatenate([y32a, y32b]) - whole).max() <= 1e-5 * np.abs(whole).max()
    assert np.abs(st32b - st64b).max() <= 1e-5 * max(np.abs(st64b).max(), 1.0)
