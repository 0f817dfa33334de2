"""Oracle of lsp::dspu::ILUFSMeter (oracle/ilufs.py).  The reference has only a manual test for it
(src/test/mtest/meters/ilufs.cpp, no expected values); the anchors are the standard -- a 0 dBFS 997 Hz sine integrates to
-3.01 LKFS (ITU-R BS.1770-4) -- and the definition of the gated mean evaluated in float64."""
import numpy as np

import oracle
from oracle import filter_design as fd
from oracle import ilufs as oi


def _sine(sr, n, amp=1.0):
    return (amp * np.sin(2 * np.pi * 997.0 * np.arange(n) / sr)).astype(np.float32)


def test_bs1770_sine_anchor_one_call_and_plugin_blocks():
    sr = 48000
    x = _sine(sr, 2 * sr)[None, :]
    m = oi.ILUFSMeter(1, 5.0)
    m.set_sample_rate(sr)
    out = m.process(x)                                       # gain = DBFS_TO_LUFS_SHIFT_GAIN
    assert abs(20.0 * np.log10(float(out[-1])) + 3.01) < 0.02
    assert out[0] == 0.0 and np.all(out[:4 * m.block_size - 1] == 0.0)      # nothing before the first full gating block
    assert m.block_size == 4800 and m.ms_int == (5 * sr - 2 * 4800 - 1) // 4800
    # host blocks of 1024 (the manual test's BUFFER_SIZE): F_BLK_FULL does not survive update_settings(), so a gating
    # block is evaluated only in the call where the quarter counter wraps -- the value still converges to the same level
    m2 = oi.ILUFSMeter(1, 5.0)
    m2.set_sample_rate(sr)
    outs = [m2.process(x[:, p:p + 1024]) for p in range(0, x.shape[1], 1024)]
    assert abs(20.0 * np.log10(float(m2.loud) * float(oi.DBFS_TO_LUFS_SHIFT_GAIN)) + 3.01) < 0.02
    changes = np.flatnonzero(np.diff(np.concatenate(outs)) != 0) + 1
    assert np.all(changes % (4 * 4800) == 0) and len(changes) == (x.shape[1] - 1) // (4 * 4800)


def test_gating_blocks_follow_the_definition():
    """Stereo noise with a silent stretch: the history holds 400 ms blocks every 100 ms, blocks below -70 LKFS are left
    out of the mean, and the integration window forgets old blocks."""
    sr = 44100
    rng = np.random.default_rng(5)
    n = 3 * sr
    x = (rng.standard_normal((2, n)) * 0.1).astype(np.float32)
    x[:, sr:sr + sr // 2] *= 1e-5                             # half a second under the absolute gate
    x[:, 2 * sr:] *= 3.0
    m = oi.ILUFSMeter(2, 1.5)
    m.set_sample_rate(sr)
    out = m.process(x, gain=1.0)
    blk = m.block_size
    coef = fd.design(fd.Params(fd.FLT_K_WEIGHTED, 0, 0.0, 0.0, 1.0, 0.0), sr)[2]
    sq = sum(oracle.biquad_cascade_f64(x[c], coef) ** 2 for c in range(2))
    cs = np.concatenate([[0.0], np.cumsum(sq)])
    nq = n // blk
    blocks = np.array([(cs[(q + 1) * blk] - cs[(q - 3) * blk]) / (4 * blk) for q in range(3, nq)])
    for q in range(3, nq):                                   # value after the q-th quarter boundary
        hist = blocks[max(0, q - 3 + 1 - m.ms_int):q - 3 + 1]
        kept = hist[hist > float(oi.GATING_ABS_THRESH)]
        want = np.sqrt(kept.mean()) if kept.size else 0.0
        got = float(out[(q + 1) * blk]) if (q + 1) * blk < n else float(m.loud)
        assert abs(got - want) <= 2e-5 * max(want, 1e-3), (q, got, want)
    assert (blocks <= float(oi.GATING_ABS_THRESH)).sum() >= 1     # the silent stretch did produce gated-out blocks


def test_infinite_mode_running_mean_and_halving():
    sr = 8000
    m = oi.ILUFSMeter(1, 0.0, 40.0)                           # max_int_time 0: integrate since clear()
    m.set_sample_rate(sr)
    assert m.block_size == 80 and m.ms_size == 64
    rng = np.random.default_rng(6)
    n = 80 * 300                                              # 297 gating blocks: more than the 0x100 halving point
    x = (rng.standard_normal((1, n)) * 0.2).astype(np.float32)
    out = m.process(x, gain=1.0)
    assert m.ms_int == 0 and m.ms_count == (0x100 >> 1) + (297 - 0x100)
    coef = fd.design(fd.Params(fd.FLT_K_WEIGHTED, 0, 0.0, 0.0, 1.0, 0.0), sr)[2]
    sq = oracle.biquad_cascade_f64(x[0], coef) ** 2
    cs = np.concatenate([[0.0], np.cumsum(sq)])
    blocks = np.array([(cs[(q + 1) * 80] - cs[(q - 3) * 80]) / 320 for q in range(3, 300)])
    # before the first halving the value is the plain mean of all blocks
    q = 200
    assert abs(float(out[(q + 1) * 80]) - np.sqrt(blocks[:q - 2].mean())) <= 2e-5 * float(out[(q + 1) * 80])
    # after it, older blocks weigh half
    w = np.ones(297); w[:0x100] = 0.5
    assert abs(float(m.loud) - np.sqrt((blocks * w).sum() / m.ms_count)) <= 2e-5 * float(m.loud)
    m.clear()
    assert m.loud == 0 and m.ms_count == 0 and not m.hist.any()


def test_period_change_and_disabled_channel():
    sr = 48000
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((2, sr)) * 0.1).astype(np.float32)
    m = oi.ILUFSMeter(2, 10.0)
    m.set_sample_rate(sr)
    m.set_active(1, False)
    m.process(x)
    one = oi.ILUFSMeter(1, 10.0)
    one.set_sample_rate(sr)
    one.process(x[:1])
    assert m.loud == one.loud                                 # a disabled channel adds nothing
    m.set_integration_period(1.0)
    m.process(x[:, :100])
    assert m.ms_int == (sr - 2 * 4800 - 1) // 4800 and m.ms_count <= m.ms_int
