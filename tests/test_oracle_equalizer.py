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
