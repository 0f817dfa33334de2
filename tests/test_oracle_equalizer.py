"""Oracle Equalizer pinned by the reference's own unit test (src/test/utest/filters/equalizer.cpp:35-92)."""
import numpy as np
import pytest

from oracle import equalizer as oe
from oracle import filter_design as fd


@pytest.mark.parametrize("mode,label", [(oe.FIR, "FIR"), (oe.FFT, "FFT"), (oe.SPM, "SPM")])
def test_reference_utest_latency(mode, label):
    """1 x FLT_BT_LRX_HIPASS 100 Hz slope 2 @48 kHz, fir_rank 13, unit impulse over 1 << 15 samples:
    arg-max |out| == get_latency()  (12288 for FIR/FFT, 8192 for SPM)."""
    rank = 13
    eq = oe.Equalizer(1, rank)
    eq.set_mode(mode)
    eq.set_sample_rate(48000)
    eq.set_params(0, fd.Params(fd.FLT_BT_LRX_HIPASS, 2, 100.0, 100.0, 1.0, 0.0))
    src = np.zeros(1 << (rank + 2), np.float32)
    src[0] = 1.0
    dst = eq.process(src)
    lat = eq.get_latency()
    assert lat == ((1 << rank) + (1 << (rank - 1)) if mode != oe.SPM else (1 << rank))
    assert int(np.abs(dst).argmax()) == lat, label
    assert 0.9 < np.abs(dst).max() < 1.1


@pytest.mark.parametrize("mode", [oe.FFT])
def test_smooth_retune_is_a_position_weighted_mix_of_both_responses(mode):
    """Restatement check of EF_XFADE (Equalizer.cpp:486-501; no reference test covers it): the block that completes
    after a smooth retune equals old * (1 - w) + new * w with w = 0 up to N/2, a ramp i/N over the next N output
    positions of that block's 2N-long result and 1 after -- built here from two non-smooth equalizers."""
    rng = np.random.default_rng(5)
    rank = 9
    N = 1 << rank
    x = (rng.standard_normal(8 * N) * 0.25).astype(np.float32)
    pa = fd.Params(fd.FLT_BT_RLC_BELL, 1, 700.0, 700.0, 2.5, 2.0)
    pb = fd.Params(fd.FLT_BT_RLC_BELL, 1, 2500.0, 2500.0, 0.4, 1.0)

    def make(p, smooth):
        e = oe.Equalizer(1, rank); e.set_mode(mode); e.set_sample_rate(48000); e.set_smooth(smooth); e.set_params(0, p)
        return e
    ea, eb = make(pa, False), make(pb, False)
    ya, yb = ea.process(x), eb.process(x)
    # smooth equalizer: first configuration A (fades in from silence over block 0), retune to B after 3 blocks
    es = make(pa, True)
    ys = np.concatenate([es.process(x[:3 * N]), (es.set_params(0, pb), es.process(x[3 * N:]))[1]])
    # The reference convolves a buffered block lazily, when the first sample after it arrives (Equalizer.cpp:477-483):
    # block 2 is still waiting when the retune happens, so it is the cross-fade block.  Its 2N-long result starts at
    # output position 3N (the block itself + the buffering latency N).
    w = np.zeros(8 * N, np.float32)
    start = 3 * N
    w[start + N // 2: start + N // 2 + N] = np.arange(N, dtype=np.float32) / np.float32(N)
    w[start + N // 2 + N:] = 1.0
    # outputs are sums of per-block results, so mix block-wise: blocks < 2 old, block 2 weighted, blocks > 2 new.
    # Build the expectation from per-block contributions of both equalizers.
    def contributions(p):
        out = np.zeros((8, 10 * N), np.float32)
        for b in range(8):
            xb = np.zeros(8 * N, np.float32); xb[b * N:(b + 1) * N] = x[b * N:(b + 1) * N]
            out[b, :8 * N] = make(p, False).process(xb)
        return out
    ca, cb = contributions(pa), contributions(pb)
    # Quirk of the reference kept as is: the ramp-down is applied to vOutBuffer, which at that moment also holds the
    # overlap tail of the blocks before (Equalizer.cpp:496), so over the first half of the ramp those tails fade as well.
    w0 = np.zeros(8 * N, np.float32)                 # block 0: the first configuration fades in from silence
    w0[N + N // 2: N + N // 2 + N] = np.arange(N, dtype=np.float32) / np.float32(N)
    w0[N + N // 2 + N:] = 1.0
    exp = ca[0, :8 * N] * w0 + ca[1, :8 * N]
    first_half = slice(start + N // 2, start + N)    # where the old tails share the faded buffer with block 2
    exp[first_half] *= (1.0 - w[first_half])
    exp += ca[2, :8 * N] * (1.0 - w) + cb[2, :8 * N] * w
    for b in range(3, 8):
        exp += cb[b, :8 * N]
    peak = float(np.abs(exp).max())
    assert float(np.abs(ys - exp).max()) <= 2e-5 * peak
    assert float(np.abs(ya - exp).max()) > 1e-2 * peak and float(np.abs(yb - exp).max()) > 1e-2 * peak
