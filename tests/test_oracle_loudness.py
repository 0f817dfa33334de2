"""Oracle of lsp::dspu::LoudnessMeter (oracle/loudness.py).  No reference unit test covers it; the anchor is the
standard: ITU-R BS.1770-4 states that a 0 dBFS, 997 Hz sine on one front channel reads -3.01 LKFS, i.e. its K-weighted
mean square is 10^((-3.01 + 0.691) / 10) (the reference carries the same constant: DBFS_TO_LUFS_SHIFT_DB = -0.691,
misc/broadcast.h:96)."""
import numpy as np

from oracle import loudness as ol


def test_bs1770_sine_anchor():
    sr = 48000
    m = ol.LoudnessMeter(1)
    m.set_sample_rate(sr)
    t = np.arange(sr)                                        # 1 s: the 400 ms window is full and settled
    x = np.sin(2 * np.pi * 997.0 * t / sr).astype(np.float32)[None, :]
    out, _ = m.process(x)
    lkfs = -0.691 + 10.0 * np.log10(float(out[-1]) ** 2)
    assert abs(lkfs - (-3.01)) < 0.02, lkfs
    assert m.latency() == 19200 and m.period == 19200


def test_stereo_mix_lfe_exclusion_linking_and_refresh_consistency():
    sr = 44100
    rng = np.random.default_rng(2)
    n = 30000
    x = (rng.standard_normal((3, n)) * 0.2).astype(np.float32)
    m = ol.LoudnessMeter(3)
    m.set_sample_rate(sr)
    m.set_designation(0, ol.CHANNEL_LEFT); m.set_designation(1, 7); m.set_designation(2, ol.CHANNEL_LFE1)
    m.set_link(0, 0.0); m.set_link(1, 0.25); m.set_period(100.0)
    out, cho = m.process(x)
    # window mean squares straight from the definition, float64
    from oracle import filter_design as fd
    import oracle
    coef = fd.design(fd.Params(fd.FLT_K_WEIGHTED, 0, 0.0, 0.0, 1.0, 0.0), sr)[2]
    P = m.period
    ms = []
    for c in range(3):
        y = oracle.biquad_cascade_f64(x[c], coef) ** 2
        cs = np.concatenate([[0.0], np.cumsum(y)])
        j = np.arange(n)
        ms.append((cs[j + 1] - cs[np.maximum(j + 1 - P, 0)]) / P)
    loud = np.sqrt(1.0 * ms[0] + 1.41 * ms[1] + 0.0 * ms[2])
    assert np.abs(out - loud).max() <= 2e-5 * loud.max()
    assert np.abs(cho[0] - np.sqrt(ms[0])).max() <= 2e-5 * loud.max()                      # link 0: own RMS
    assert np.abs(cho[1] - (0.25 * loud + 0.75 * np.sqrt(ms[1]))).max() <= 2e-5 * loud.max()
    assert np.abs(cho[2] - loud).max() <= 2e-5 * loud.max()                                # link 1 (default): the mix
    # chunking of the calls does not matter beyond round-off
    m2 = ol.LoudnessMeter(3)
    m2.set_sample_rate(sr)
    m2.set_designation(0, ol.CHANNEL_LEFT); m2.set_designation(1, 7); m2.set_designation(2, ol.CHANNEL_LFE1)
    m2.set_link(0, 0.0); m2.set_link(1, 0.25); m2.set_period(100.0)
    parts = [m2.process(x[:, a:b])[0] for a, b in ((0, 777), (777, 9000), (9000, n))]
    assert np.abs(np.concatenate(parts) - out).max() <= 2e-6 * loud.max()
