"""Oracle of SpectralSplitter / FFTCrossover / crossover::* (oracle/splitter.py).  The reference has manual tests only
(src/test/mtest/util/spectral_splitter.cpp, fft_crossover.cpp); the pins are properties of the algorithm."""
import numpy as np
import pytest

from oracle import splitter as osp


def _collect(n):
    buf = np.zeros(n, np.float32)

    def sink(s, first, count, buf=buf):
        buf[sink.pos:sink.pos + count] = s
        sink.pos += count
    sink.pos = 0
    return buf, sink


@pytest.mark.parametrize("rank,chunk,phase,calls", [(8, 0, 0.0, (1000,)), (9, 7, 0.0, (100, 3, 700, 197)), (8, 6, 0.5, (333, 667))])
def test_pass_through_handlers_delay_the_input_by_latency(rank, chunk, phase, calls):
    """func = identity: sqr_cosine windows at 50 % overlap sum to one, so each sink sees the input delayed by latency().
    The handler without a func takes the FIRST 2*frame samples of the analysis buffer (SpectralSplitter.cpp:330), so with a
    chunk rank below the rank it lags by the extra history."""
    n = sum(calls)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n).astype(np.float32)
    sp = osp.SpectralSplitter(rank, 3)
    sp.set_chunk_rank(chunk); sp.set_phase(phase)
    b0, s0 = _collect(n); b1, s1 = _collect(n)
    sp.bind(0, lambda spec, r: spec, s0)
    sp.bind(2, None, s1)
    pos = 0
    for k in calls:
        sp.process(x[pos:pos + k], k); pos += k
    lat = sp.latency()
    assert lat == (1 << (chunk if chunk else rank))
    frame = lat // 2
    skip = 2 * lat                                            # the first frames ramp in through the window
    assert np.abs(b0[skip:] - x[skip - lat:n - lat]).max() < 2e-5 * np.abs(x).max()
    lag = lat + ((1 << rank) - 2 * frame)
    assert np.abs(b1[skip + lag:] - x[skip:n - lag]).max() < 1e-6


def test_brickwall_bands_of_the_manual_test_sum_to_the_delayed_input():
    """spectral_splitter_func of the reference's manual test (mtest/util/spectral_splitter.cpp:38-84): bins in
    [imin, imax) kept; complementary bands add up to the input."""
    rank, chunk, n = 10, 8, 6000
    rng = np.random.default_rng(4)
    x = rng.standard_normal(n).astype(np.float32)
    N = 1 << rank
    edges = [0, 5, 40, 200, N // 2 + 1]
    sp = osp.SpectralSplitter(rank, 6)
    sp.set_rank(rank); sp.set_chunk_rank(chunk); sp.set_phase(0)
    outs = []
    for i in range(4):
        idx = np.minimum(np.arange(N), N - np.arange(N)); idx[0] = 0
        keep = ((idx >= edges[i]) & (idx < edges[i + 1])).astype(np.float32)

        def func(spec, r, keep=keep):
            spec[0::2] *= keep; spec[1::2] *= keep
            return spec
        buf, sink = _collect(n)
        outs.append(buf)
        sp.bind(i, func, sink)
    sp.process(x, n)
    lat = sp.latency()
    total = np.sum(outs, axis=0)
    assert np.abs(total[2 * lat:] - x[lat:n - lat]).max() < 5e-5 * np.abs(x).max()
    assert all(np.abs(o).max() > 0.01 for o in outs)


def test_crossover_curves():
    f0 = 1000.0
    # -6 dB at the crossover point, complementary magnitudes for the usual slopes
    for slope in (-12.0, -24.0, -64.0):
        assert osp.hipass(f0, f0, slope) == np.float32(0.5) and osp.lopass(f0, f0, slope) == np.float32(0.5)
        for f in (100.0, 700.0, 1500.0, 9000.0):
            assert abs(float(osp.hipass(f, f0, slope)) + float(osp.lopass(f, f0, slope)) - 1.0) < 1e-6
        # one octave above the crossover the low-pass is down by slope dB (relative to the -6 dB point)
        assert abs(20 * np.log10(float(osp.lopass(2 * f0, f0, slope)) / 0.5) - slope) < 1e-3
    # slopes above -3 dB/oct: the special -6 dB/oct transition of one octave
    assert osp.hipass(500.0, f0, 0.0) == np.float32(0.5) and osp.hipass(2000.0, f0, 0.0) == 1.0
    assert osp.lopass(500.0, f0, 0.0) == 1.0 and osp.lopass(2000.0, f0, 0.0) == np.float32(0.5)
    assert 0.5 < float(osp.hipass(1500.0, f0, 0.0)) < 1.0
    # FFT-ordered masks: symmetric, DC pinned (0 for high-pass, 1 for low-pass), apply == product of the set forms
    sr, rank = 48000.0, 9
    hp = osp.hipass_fft_set(300.0, -24.0, sr, rank); lp = osp.lopass_fft_set(3000.0, -32.0, sr, rank)
    N = 1 << rank
    assert hp[0] == 0.0 and lp[0] == 1.0
    assert np.array_equal(hp[1:N // 2], hp[:N // 2:-1]) and np.array_equal(lp[1:N // 2], lp[:N // 2:-1])
    both = osp.lopass_fft_apply(hp, 3000.0, -32.0, sr, rank)
    assert np.array_equal(both, (hp * lp).astype(np.float32))
    assert hp[7] == osp.hipass(np.float32(7) * np.float32(sr / N), 300.0, -24.0)


def test_fft_crossover_bands_and_flag_rules():
    """The manual test's band plan (mtest/util/fft_crossover.cpp:74-104) on noise: adjacent bands meet at -6 dB, so with
    flatten 1 the band outputs add up to the delayed input away from the crossover dips; the update flags follow the
    reference (a filter disabled through set_lpf(.., false) leaves the mask stale)."""
    rank, n, sr = 10, 8192, 48000
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n).astype(np.float32)
    xo = osp.FFTCrossover(rank, 3)
    xo.set_sample_rate(sr)
    xo.set_lpf(0, 400.0, -32.0, True)
    xo.set_hpf(1, 400.0, -32.0, True); xo.set_lpf(1, 4000.0, -32.0, True)
    xo.set_hpf(2, 4000.0, -32.0, True)
    bufs = []
    for i in range(3):
        buf = np.zeros(n, np.float32)
        pos = [0]

        def h(band, s, first, count, buf=buf, pos=pos):
            buf[pos[0]:pos[0] + count] = s; pos[0] += count
        bufs.append(buf)
        xo.enable_band(i, True)
        xo.set_handler(i, h)
    xo.process(x, n)
    lat = xo.latency()
    assert lat == 1 << rank
    # the masks of this plan add up to 1 within the steep-slope leakage: hp(f, f0) + lp(f, f0) = 1 and the far filter ~ 1
    msum = sum(b["fft"] for b in xo.b)
    assert np.abs(msum[1:] - 1.0).max() < 0.02 and msum[0] == 1.0
    total = np.sum(bufs, axis=0)
    assert np.abs(total[2 * lat:] - x[lat:n - lat]).max() < 0.05 * np.abs(x).max()
    # flag rules
    b = xo.b[0]
    assert not b["update"]
    xo.set_lpf(0, 400.0, -32.0, False)                       # disabling through set_lpf does not request an update
    assert not b["update"] and not b["lpf"]
    xo.set_gain(0, 2.0)
    assert b["update"]
    xo.update_band(b)
    assert np.all(b["fft"] == np.float32(2.0))               # no filter: flatten * gain everywhere
    chart = xo.freq_chart(1, np.array([100.0, 400.0, 1000.0, 4000.0, 10000.0], np.float32))
    assert chart[1] == np.float32(0.5) * osp.lopass(400.0, 4000.0, -32.0) and chart[2] > 0.95 and chart[0] < 0.01
