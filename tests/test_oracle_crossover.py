"""Oracle of lsp::dspu::Crossover (oracle/crossover.py): no reference unit test exists for it, so the restatement is
anchored on what the reference's own documentation promises (util/Crossover.h:31-66): Linkwitz-Riley splits whose bands
sum to an all-pass -- flat magnitude, in the time domain and in the frequency charts -- plus plan/band bookkeeping."""
import numpy as np
import pytest

from oracle import crossover as oc
from oracle import filter_design as fd


def make(bands, splits, sr=48000):
    x = oc.Crossover(bands)
    x.set_sample_rate(sr)
    for i, (slope, freq) in enumerate(splits):
        x.set_slope(i, slope)
        x.set_frequency(i, freq)
    return x


def test_default_split_frequencies_and_inactive_bands():
    x = oc.Crossover(4)
    assert [round(float(s["freq"]), 2) for s in x.split] == [69.99, 489.9, 3429.02] or \
           np.allclose([float(s["freq"]) for s in x.split], 10.0 * np.exp(np.arange(1, 4) * np.log(2400.0) / 4), rtol=1e-5)
    # all split points off: one band, plain gain
    x.set_gain(0, 0.5)
    y = x.process(np.ones(16, np.float32))
    assert list(y.keys()) == [0] and np.all(y[0] == 0.5)
    assert x.band_info(0)["active"] and not x.band_info(1)["active"]
    assert x.freq_chart(2, [100.0, 1000.0]).tolist() == [0j, 0j]


@pytest.mark.parametrize("slope,nsplit", [(1, 1), (2, 3), (3, 3), (5, 3)])
@pytest.mark.parametrize("mode", [oc.MODE_BT])
def test_bands_sum_to_an_allpass(slope, nsplit, mode):
    """LR crossover: sum of all band outputs has the input's magnitude spectrum (an all-pass).
    LR2 is checked with a single split point only: with more, the reference compensates the lower bands with
    FLT_BT_RLC_ALLPASS of slope 1 = ((1-s)/(1+s))^2 (Filter.cpp:977-992) while an LR2 split itself sums to
    (1-s)/(1+s), so its LR2 bands do not add up to a flat magnitude -- reproduced as is, not asserted."""
    sr = 48000
    freqs = [200.0, 1500.0, 6000.0]
    x = make(4, [(slope if i < nsplit else 0, freqs[i]) for i in range(3)], sr)
    for i in range(3):
        x.set_mode(i, mode)
    rng = np.random.default_rng(1)
    n = 1 << 15
    sig = np.zeros(2 * n, np.float32)
    sig[:n] = (rng.standard_normal(n) * 0.25).astype(np.float32)        # zeros behind it: the filters ring out
    bands = x.process(sig)
    assert sorted(bands.keys()) == list(range(nsplit + 1))
    total = sum(bands[k].astype(np.float64) for k in bands)
    # an all-pass keeps the energy ...
    e_in, e_out = float(np.sum(sig.astype(np.float64) ** 2)), float(np.sum(total ** 2))
    assert abs(e_out / e_in - 1.0) < 2e-3, e_out / e_in
    # ... and the magnitude spectrum
    S = np.abs(np.fft.rfft(sig.astype(np.float64)))
    T = np.abs(np.fft.rfft(total))
    k = np.arange(64, n - 64, 197)
    smooth = lambda v: np.array([np.sqrt((v[i - 48:i + 48] ** 2).mean()) for i in k])
    ratio = smooth(T) / smooth(S)
    assert np.all(np.abs(20 * np.log10(ratio)) < 0.05), (ratio.min(), ratio.max())
    # frequency charts: |sum of the band charts| == 1
    f = np.geomspace(20.0, 20000.0, 200).astype(np.float32)
    h = sum(x.freq_chart(b, f).astype(np.complex128) for b in range(4))
    if nsplit == 1:
        assert np.all(np.abs(np.abs(h) - 1.0) < 2e-3), np.abs(np.abs(h) - 1.0).max()
    # with more split points the charts of the inner bands leave out the all-pass filters (Crossover.cpp:531-533: only
    # filter 0 of the low-pass equalizer), so their sum is not the all-pass the signal path is


def test_plan_is_sorted_and_gains_land_on_their_bands():
    sr = 48000
    x = make(4, [(2, 5000.0), (0, 300.0), (2, 400.0)], sr)          # split 1 off, split 2 below split 0
    x.set_gain(0, 2.0); x.set_gain(1, 3.0); x.set_gain(3, 0.5)
    x.reconfigure()
    assert x.plan == [2, 0]
    info = [x.band_info(b) for b in range(4)]
    assert info[0]["end"] == 400.0 and info[3]["start"] == 400.0 and info[3]["end"] == 5000.0
    assert info[1]["start"] == 5000.0 and info[1]["end"] == 24000.0 and not info[2]["active"]
    f = np.array([50.0, 1500.0, 15000.0], np.float32)
    # deep inside each band the chart magnitude is that band's gain: band 0 (gain 2), band 3 (0.5), band 1 (3)
    assert abs(abs(x.freq_chart(0, f)[0]) - 2.0) < 2e-2
    assert abs(abs(x.freq_chart(3, f)[1]) - 0.5) < 2e-2
    assert abs(abs(x.freq_chart(1, f)[2]) - 3.0) < 5e-2
    y = x.process(np.zeros(8, np.float32), handlers=[0, 3])
    assert sorted(y.keys()) == [0, 3]
