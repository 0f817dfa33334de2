"""GPU parity of mi_ilufs_bank_* (lsp::dspu::ILUFSMeter) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from oracle import ilufs as oi
from oracle import loudness as ol

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _run(gpu, bank, refs, x, calls, K, gain=None):
    M = len(refs)
    n = sum(calls)
    got = np.zeros((M, n), np.float32); ref = np.zeros((M, n), np.float32)
    pos = 0
    kw = {} if gain is None else {"gain": gain}
    for k in calls:
        out = gpu.DeviceBuffer((M, k))
        bank.process(out, gpu.DeviceBuffer.from_host(x[:, pos:pos + k]), k, **kw)
        got[:, pos:pos + k] = out.download()
        for m in range(M):
            ref[m, pos:pos + k] = refs[m].process(x[m * K:(m + 1) * K, pos:pos + k], **kw)
        pos += k
    return got, ref


def test_bs1770_sine_anchor_on_gpu(gpu):
    """ITU-R BS.1770-4: a 0 dBFS 997 Hz sine integrates to -3.01 LKFS (and half the amplitude to -9.03)."""
    sr = 48000
    n = 2 * sr
    t = np.arange(n)
    x = np.stack([np.sin(2 * np.pi * 997.0 * t / sr), 0.5 * np.sin(2 * np.pi * 997.0 * t / sr)]).astype(np.float32)
    bank = gpu.ILUFSBank(2, 1, 5.0)
    bank.set_sample_rate(sr)
    out = gpu.DeviceBuffer((2, n))
    bank.process(out, gpu.DeviceBuffer.from_host(x), n)
    y = out.download()
    lkfs = 20.0 * np.log10(y[:, -1])
    assert abs(lkfs[0] + 3.01) < 0.02 and abs(lkfs[1] + 9.03) < 0.02, lkfs
    assert np.all(y[:, :4 * 4800 - 1] == 0.0)
    np.testing.assert_allclose(bank.loudness() * np.float32(bank.DBFS_TO_LUFS_SHIFT_GAIN), y[:, -1], rtol=1e-5)   # one more gating block ended with the last sample
    bank.close()


@pytest.mark.parametrize("calls", [(3 * 44100,), (50000, 1024, 1024, 30000, 777, 49475), (1024,) * 100])
def test_gated_integration_matches_oracle(gpu, calls):
    """Two meters of three channels (one LFE, one of the +1.5 dB group), a stretch under the absolute gate, an integration
    window shorter than the signal; single call, ragged calls and plugin-sized calls (where only the call in which the
    quarter counter wraps evaluates gating blocks, as in the reference)."""
    sr, M, K = 44100, 2, 3
    n = sum(calls)
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((M * K, n)) * 0.1).astype(np.float32)
    x[:, sr:sr + sr // 2] *= 1e-5
    x[:, 2 * sr:] *= 3.0
    x[K:] *= 0.25
    bank = gpu.ILUFSBank(M, K, 1.5)
    refs = [oi.ILUFSMeter(K, 1.5) for _ in range(M)]
    for obj in [bank] + refs:
        obj.set_sample_rate(sr)
        obj.set_designation(0, ol.CHANNEL_LEFT); obj.set_designation(1, 7); obj.set_designation(2, ol.CHANNEL_LFE1)
    got, ref = _run(gpu, bank, refs, x, calls, K)
    peak = float(np.abs(ref).max())
    assert peak > 0
    assert np.abs(got - ref).max() <= TOL * peak, np.abs(got - ref).max() / peak
    np.testing.assert_allclose(bank.loudness(), [float(r.loud) for r in refs], rtol=2e-5)
    bank.close()


def test_infinite_mode_halving_clear_and_disabled_channel(gpu):
    sr, K = 8000, 2
    n = 80 * 300
    rng = np.random.default_rng(22)
    x = (rng.standard_normal((K, n)) * 0.2).astype(np.float32)
    bank = gpu.ILUFSBank(1, K, 0.0, 40.0)
    ref = oi.ILUFSMeter(K, 0.0, 40.0)
    for obj in (bank, ref):
        obj.set_sample_rate(sr)
        obj.set_active(1, False)
    got, want = _run(gpu, bank, [ref], x, (n // 2, n // 2), K, gain=1.0)
    assert ref.ms_count < 0x100 and ref.ms_count > 0x80      # the halving did happen
    peak = float(want.max())
    assert np.abs(got - want).max() <= TOL * peak, np.abs(got - want).max() / peak
    for obj in (bank, ref):
        obj.clear()
        obj.set_active(1, True)
    assert bank.loudness()[0] == 0.0
    got, want = _run(gpu, bank, [ref], x[:, :8000], (8000,), K, gain=1.0)
    assert np.abs(got - want).max() <= TOL * float(want.max())
    bank.close()


def test_weighting_and_integration_period_change(gpu):
    sr, K = 48000, 1
    rng = np.random.default_rng(23)
    x = (rng.standard_normal((K, 3 * sr)) * 0.2).astype(np.float32)
    x[:, sr:] *= 0.3
    bank = gpu.ILUFSBank(1, K, 10.0)
    ref = oi.ILUFSMeter(K, 10.0)
    for obj in (bank, ref):
        obj.set_sample_rate(sr)
        obj.set_weighting(ol.WEIGHT_NONE)
    g1, w1 = _run(gpu, bank, [ref], x[:, :2 * sr], (2 * sr,), K)
    for obj in (bank, ref):
        obj.set_integration_period(0.8)                       # shrinks the window: older blocks drop out
    g2, w2 = _run(gpu, bank, [ref], x[:, 2 * sr:], (sr,), K)
    got = np.concatenate([g1, g2], 1); want = np.concatenate([w1, w2], 1)
    assert np.abs(got - want).max() <= TOL * float(want.max())
    assert ref.ms_int == (int(np.float32(0.8) * np.float32(sr)) - 2 * 4800 - 1) // 4800
    bank.close()


def test_argument_errors(gpu):
    lib = gpu.lib
    bank = gpu.ILUFSBank(1, 1)
    x = gpu.DeviceBuffer((1, 64))
    with pytest.raises(gpu.MiError):
        bank.process(None, x, 64)                            # no sample rate yet
    with pytest.raises(gpu.MiError):
        bank.set_designation(3, 1)
    bank.close()


def test_disabled_channel_filter_freezes_until_reenabled(gpu):
    sr, K, n = 48000, 2, 19200
    rng = np.random.default_rng(24)
    x = (rng.standard_normal((K, 3 * n)) * 0.2).astype(np.float32)
    x[1] += 0.7
    bank = gpu.ILUFSBank(1, K, 4.0)
    ref = oi.ILUFSMeter(K, 4.0)
    for obj in (bank, ref):
        obj.set_sample_rate(sr)
    got, want = [], []
    for blk, active in ((0, True), (1, False), (2, True)):
        for obj in (bank, ref):
            obj.set_active(1, active)
        g, w = _run(gpu, bank, [ref], x[:, blk * n:(blk + 1) * n], (n,), K)
        got.append(g); want.append(w)
    got = np.concatenate(got, 1); want = np.concatenate(want, 1)
    # DC into the K weighting's 38 Hz high-pass: float32 round-off of the recursion itself, see test_loudness_gpu.py
    assert np.abs(got - want).max() <= 3e-5 * float(want.max())
    np.testing.assert_allclose(bank.loudness(), [float(ref.loud)], rtol=3e-5)
    bank.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_operation_sequences(gpu, seed):
    """Differential stress of the integrated meter bank: integration period, weighting, designation and activity changes,
    clear() and ragged process() calls (every call starts with the block-full flag cleared, as in the reference)."""
    rng = np.random.default_rng(17000 + seed)
    M, K, sr = 2, 2, 48000
    max_int = float(rng.choice([0.0, 2.0]))
    bank = gpu.ILUFSBank(M, K, max_int, 100.0)
    refs = [oi.ILUFSMeter(K, max_int, 100.0) for _ in range(M)]
    for obj in [bank] + refs:
        obj.set_sample_rate(sr)
    blk = refs[0].block_size
    weight = ol.WEIGHT_K
    log = []
    level = 1e-4                                             # loudest input since the filters were last cleared
    GATE = float(oi.GATING_ABS_THRESH)
    at_gate = [False] * M
    for step in range(40):
        op = rng.choice(["process", "process", "process", "period", "weighting", "designation", "active", "clear"])
        if op == "process":
            n = int(rng.choice([1, blk - 1, blk, blk + 1, 4 * blk, 4 * blk + 1, 9 * blk + 7, int(rng.integers(1, 12 * blk))]))
            # loud, or far below the absolute gate (-70 LKFS is an amplitude of 3e-4): a block AT the gate would be kept or
            # dropped by the last bit of its sum
            x = (rng.standard_normal((M * K, n)) * float(rng.choice([0.2, 1e-6]))).astype(np.float32)
            got, want = _run(gpu, bank, refs, x, (n,), K, gain=float(rng.choice([1.0, 0.9235])))
            tol = TOL if weight in (ol.WEIGHT_NONE, ol.WEIGHT_K) else 5e-5
            # relative to the level the weighting filters have seen: after loud material their decaying memory is what a
            # quiet stretch measures, and that tail is reproducible to the float32 round-off of the loud part only
            level = max(level, float(np.abs(x).max()))
            # A gating block made of the weighting filters' decaying memory can land next to the absolute gate, and that
            # tail is only reproducible to the round-off of the loud material before it (a few per cent at -70 LKFS):
            # whether such a block counts is decided by that round-off, in the reference's own builds as well.  While a
            # meter's window holds a block within 5 % of the gate its outputs are not compared sample by sample; instead
            # its history is downloaded and checked block by block, and its held loudness must be the gated mean of ITS
            # OWN history -- either gating outcome of the borderline block is accepted, nothing else.  (With
            # max_int_time = 0 the gate decides what is ADDED to the running mean, so there the histories themselves part
            # ways and the meter is left out until clear().)
            hist, head, count = bank.history()
            for m in range(M):
                r = refs[m]
                h = np.asarray(r.hist, np.float64).ravel()
                if max_int == 0.0:
                    # (the running mean keeps no record of a block the gate dropped: the oracle notes how close any
                    # evaluated block came -- seed 12445 of the round-2 sweep had one the GPU kept and the oracle dropped)
                    if r.gate_margin < 0.05:
                        at_gate[m] = True
                    continue
                live = [(r.ms_head + r.ms_size - 1 - k) % r.ms_size for k in range(r.ms_count)]
                near = [i for i in live if abs(float(h[i]) - GATE) < 0.05 * GATE]
                # (a borderline block may also have come AND gone inside this call when the window is short: the oracle notes
                # how close any block evaluated by the call came -- seed 23821 of the round-2 sweep, a window of one block)
                # (and the held value outlives the window that produced it: a call that evaluates no block still shows the
                # loudness of the last evaluation, borderline block included, even when the integration period has since shrunk
                # the window to blocks far from the gate -- seed 34162 of the round-3 sweep)
                evaluated = np.isfinite(r.call_gate_margin)
                at_gate[m] = bool(near) or r.call_gate_margin < 0.05 or (at_gate[m] and not evaluated)
                if not at_gate[m]:
                    continue
                assert int(head[m]) == r.ms_head and int(count[m]) == r.ms_count, (seed, step, m)
                gh = hist[m].astype(np.float64)
                for i in live:                               # every block of the window, the borderline ones included
                    assert abs(gh[i] - h[i]) <= 3 * tol * level * level, (seed, step, m, i, gh[i], h[i])
                # the held value is the gated mean of the window AS OF THE LAST EVALUATION: only a call in which a block was
                # evaluated leaves it equal to the mean of the window as it is now (seed 13582 of the round-2 sweep: the
                # integration period shrank the window, then a call of one quarter evaluated nothing)
                if r.blk_full:
                    kept = [gh[i] for i in live if gh[i] > GATE]
                    own = float(np.sqrt(np.mean(kept))) if kept else 0.0
                    assert abs(float(bank.loudness()[m]) - own) <= 3 * tol * max(level, own), (seed, step, m, own)
            ok = [m for m in range(M) if not at_gate[m]]
            if ok:
                err = float(np.abs(got[ok] - want[ok]).max())
                assert err <= 3 * tol * level, (seed, step, n, err / level, float(np.abs(want).max()), log[-8:])
                np.testing.assert_allclose(np.asarray(bank.loudness())[ok], [float(refs[m].loud) for m in ok], rtol=0, atol=3 * tol * level)
        elif op == "period":
            p = float(rng.choice([0.05, 0.4, 1.0, 2.0, 5.0]))
            for obj in [bank] + refs:
                obj.set_integration_period(p)
        elif op == "weighting":
            weight = int(rng.choice([ol.WEIGHT_NONE, ol.WEIGHT_K, ol.WEIGHT_K, ol.WEIGHT_A]))
            for obj in [bank] + refs:
                obj.set_weighting(weight)
        elif op == "designation":
            k, d = int(rng.integers(0, K)), int(rng.choice([ol.CHANNEL_LEFT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1]))
            for obj in [bank] + refs:
                obj.set_designation(k, d)
        elif op == "active":
            k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2))
            for obj in [bank] + refs:
                obj.set_active(k, a)
        else:
            for obj in [bank] + refs:
                obj.clear()
            level = 1e-4
            at_gate = [False] * M
        log.append(str(op))
    bank.close()


@pytest.mark.parametrize("mode", ["finite", "infinite"])
def test_bookkeeping_riding_on_the_filter_launch_equals_two_launches(gpu, monkeypatch, mode):
    """Calls that are a multiple of 16 samples and longer than 2048 do the meter's bookkeeping on the last workgroup of the
    weighting filter's launch (biquad_sumsq_ilufs_kernel); MI_ILUFS_TWO_LAUNCHES keeps the separate kernel.  Both lay their
    sums out over the same 256 virtual threads: the same floats bit for bit, output rows, loudness and history, over
    ragged and whole calls, five channels per meter (more rows per meter than the two of a stereo pair)."""
    sr, M, K = 48000, 6, 5
    calls = (4096, 4800, 2064, 19200, 8192, 1024, 4096, 38400, 4112)
    n = sum(calls)
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((M * K, n)) * 0.2).astype(np.float32)
    x[:, 30000:42000] *= 1e-5
    results = []
    for two in (False, True):
        if two:
            monkeypatch.setenv("MI_ILUFS_TWO_LAUNCHES", "1")
        else:
            monkeypatch.delenv("MI_ILUFS_TWO_LAUNCHES", raising=False)
        bank = gpu.ILUFSBank(M, K, 0.0 if mode == "infinite" else 1.2)
        bank.set_sample_rate(sr)
        for c, d in enumerate((ol.CHANNEL_LEFT, ol.CHANNEL_RIGHT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1)):
            bank.set_designation(c, d)
        bank.set_active(3, False)
        if mode == "infinite":
            bank.set_integration_period(0.0)
        ys, pos = [], 0
        for k in calls:
            out = gpu.DeviceBuffer((M, k))
            bank.process(out, gpu.DeviceBuffer.from_host(x[:, pos:pos + k]), k, gain=0.7)
            ys.append(out.download())
            pos += k
        results.append((np.concatenate(ys, axis=1), bank.loudness().copy(), [np.array(v) for v in bank.history()]))
        bank.close()
    (y0, l0, h0), (y1, l1, h1) = results
    assert float(np.abs(y1).max()) > 0
    assert np.array_equal(y0, y1) and np.array_equal(l0, l1)
    for a, b in zip(h0, h1):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("K", [2, 1])
@pytest.mark.parametrize("mode", ["finite", "infinite"])
def test_stereo_meter_in_one_workgroup_equals_the_other_forms(gpu, monkeypatch, mode, K):
    """Two channels per meter, every channel enabled: the meter's two rows run the weighting filter side by side in ONE
    workgroup, which goes straight on with the bookkeeping (biquad_sumsq_ilufs_pair_kernel; no hand-over through memory).
    Same floats as the rows in workgroups of their own with the bookkeeping riding on the last (MI_DSPU_TEST_PATH=ilufs_rows_apart) and
    as two launches (MI_ILUFS_TWO_LAUNCHES) -- output rows, loudness and history; and a bank with a channel switched off,
    which cannot pair its rows, still agrees with its two-launch form.  K = 1: the mono meter, whose single row needs no
    hand-over either (the same kernel with one row per workgroup)."""
    sr, M = 48000, 7
    calls = (4096, 4800, 2064, 19200, 8192, 1024, 4096, 38400, 4112)
    rng = np.random.default_rng(78)
    x = (rng.standard_normal((M * K, sum(calls))) * 0.2).astype(np.float32)
    x[:, 30000:42000] *= 1e-5

    def run(env, off=None):
        for k in ("MI_DSPU_TEST_PATH", "MI_ILUFS_TWO_LAUNCHES"):
            monkeypatch.delenv(k, raising=False)
        if env == "MI_ILUFS_ROWS_APART":
            monkeypatch.setenv("MI_DSPU_TEST_PATH", "ilufs_rows_apart")
        elif env:
            monkeypatch.setenv(env, "1")
        bank = gpu.ILUFSBank(M, K, 0.0 if mode == "infinite" else 1.2)
        bank.set_sample_rate(sr)
        bank.set_designation(0, ol.CHANNEL_LEFT)
        if K > 1:
            bank.set_designation(1, ol.CHANNEL_RIGHT)
        if off is not None:
            bank.set_active(off, False)
        if mode == "infinite":
            bank.set_integration_period(0.0)
        ys, pos = [], 0
        for k in calls:
            out = gpu.DeviceBuffer((M, k))
            bank.process(out, gpu.DeviceBuffer.from_host(x[:, pos:pos + k]), k, gain=0.7)
            ys.append(out.download())
            pos += k
        res = (np.concatenate(ys, axis=1), bank.loudness().copy(), [np.array(v) for v in bank.history()])
        bank.close()
        return res

    def same(a, b):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for u, v in zip(a[2], b[2]):
            assert np.array_equal(u, v)
    pair, apart, two = run(None), run("MI_ILUFS_ROWS_APART"), run("MI_ILUFS_TWO_LAUNCHES")
    assert float(np.abs(pair[0]).max()) > 0
    same(pair, apart); same(pair, two)
    same(run(None, off=K - 1), run("MI_ILUFS_TWO_LAUNCHES", off=K - 1))
