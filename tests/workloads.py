"""Seeded synthetic workloads of BASELINE.md section 4 (shared by tests and bench.py)."""
import numpy as np

from oracle import filter_design as fd

SR = 48000


def c2_coefficients(channels, seed=3):
    """C2: FLT_BT_LRX_LOPASS slope 4 (8 biquads), Q 0.75, cutoff log-uniform 200 Hz..18 kHz per channel."""
    rng = np.random.default_rng(seed)
    fc = np.exp(rng.uniform(np.log(200.0), np.log(18000.0), size=channels))
    coef = np.zeros((channels, 8, 5), np.float32)
    for c in range(channels):
        _, _, bq = fd.design(fd.Params(fd.FLT_BT_LRX_LOPASS, 4, fc[c], fc[c], 1.0, 0.75), SR)
        assert bq.shape == (8, 5)
        coef[c] = bq
    return coef, fc


def c2_input(channels, samples, blocks=1, seed=2):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((blocks, channels, samples)) * 0.25).astype(np.float32)


def design(ftype, slope=1, freq=1000.0, freq2=1000.0, gain=1.0, q=0.0, sr=SR):
    return fd.design(fd.Params(ftype, slope, freq, freq2, gain, q), sr)[2]
