"""GPU parity of mi_loudness_bank_* (lsp::dspu::LoudnessMeter) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from oracle import loudness as ol
from conftest import IIR_REF_FACTOR, record_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5
_NOISE = {}


def _weighting_noise(weight, sr):
    """Float32 round-off of a weighting filter's recursion relative to the peak of its output: |float32 oracle - the same
    recurrence in float64| on 8192 samples of noise.  K-weighting (a high-pass pole at 38 Hz) gives
    4e-5 ... 9e-5: the filtered signal itself is only reproducible to that, a loudness value hides it by averaging the
    squares over a window of thousands of samples."""
    if (weight, sr) not in _NOISE:
        from oracle import binding as B, filter_design as fd
        coef = fd.design(fd.Params(ol._TYPES[weight], 0, 0.0, 0.0, 1.0, 0.0), sr)[2]
        coef = np.asarray(coef, np.float32).reshape(-1, 5)[:4]
        worst = 0.0
        for k in range(4):
            x = (np.random.default_rng(900 + k).standard_normal(8192) * 0.2).astype(np.float32)
            if coef.shape[0] == 0:
                break
            y32, _ = B.biquad_cascade(x, coef, np.zeros((coef.shape[0], 2), np.float32))
            y64 = B.biquad_cascade_f64(x, coef)
            worst = max(worst, float(np.abs(y32 - y64).max() / np.abs(y64).max()))
        _NOISE[(weight, sr)] = worst
    return _NOISE[(weight, sr)]


def test_bs1770_sine_anchor_on_gpu(gpu):
    """ITU-R BS.1770-4: a 0 dBFS 997 Hz sine reads -3.01 LKFS."""
    sr = 48000
    bank = gpu.LoudnessBank(2, 1)
    bank.set_sample_rate(sr)
    t = np.arange(sr)
    x = np.stack([np.sin(2 * np.pi * 997.0 * t / sr), 0.5 * np.sin(2 * np.pi * 997.0 * t / sr)]).astype(np.float32)
    out = gpu.DeviceBuffer((2, sr))
    bank.process(out, None, gpu.DeviceBuffer.from_host(x), sr)
    y = out.download()
    lkfs = -0.691 + 20.0 * np.log10(y[:, -1])
    assert abs(lkfs[0] + 3.01) < 0.02 and abs(lkfs[1] + 9.03) < 0.02, lkfs
    np.testing.assert_allclose(bank.loudness(), y[:, -1], rtol=0, atol=0)
    assert bank.latency() == 19200
    bank.close()


@pytest.mark.parametrize("calls", [(30000,), (777, 8223, 4096, 16904)])
def test_three_channel_meters_match_oracle(gpu, calls):
    """Two independent meters of three channels: designations (incl. the +1.5 dB group and an LFE), links, a short
    period, a disabled channel, gain -- per-sample loudness and linked per-channel values against the oracle."""
    sr, M, K = 44100, 2, 4
    n = sum(calls)
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((M * K, n)) * 0.2).astype(np.float32)
    x[K:] *= 0.5
    bank = gpu.LoudnessBank(M, K)
    refs = [ol.LoudnessMeter(K) for _ in range(M)]
    for obj in [bank] + refs:
        obj.set_sample_rate(sr)
        obj.set_designation(0, ol.CHANNEL_LEFT); obj.set_designation(1, 7); obj.set_designation(2, ol.CHANNEL_LFE1)
        obj.set_designation(3, ol.CHANNEL_RIGHT)
        obj.set_link(0, 0.0); obj.set_link(1, 0.25)
        obj.set_period(100.0)
        obj.set_active(3, False)
    got = np.zeros((M, n), np.float32); gch = np.zeros((M * K, n), np.float32)
    ref = np.zeros((M, n), np.float32); rch = np.zeros((M * K, n), np.float32)
    pos = 0
    for k in calls:
        out = gpu.DeviceBuffer((M, k)); ch = gpu.DeviceBuffer.from_host(np.full((M * K, k), -1.0, np.float32))
        bank.process(out, ch, gpu.DeviceBuffer.from_host(x[:, pos:pos + k]), k, gain=0.5)
        got[:, pos:pos + k] = out.download(); gch[:, pos:pos + k] = ch.download()
        for m in range(M):
            o, c = refs[m].process(x[m * K:(m + 1) * K, pos:pos + k], gain=0.5)
            ref[m, pos:pos + k] = o; rch[m * K:(m + 1) * K, pos:pos + k] = c
        pos += k
    peak = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= TOL * peak, np.abs(got - ref).max() / peak
    for r in range(M * K):
        if r % K == 3:
            assert np.all(gch[r] == -1.0)                    # disabled channel: its output is not touched
        else:
            assert np.abs(gch[r] - rch[r]).max() <= TOL * peak, (r, np.abs(gch[r] - rch[r]).max() / peak)
    np.testing.assert_allclose(bank.loudness(), [float(refs[m].loud) for m in range(M)], rtol=2e-5)
    bank.close()


def test_weightings_clear_and_reenable(gpu):
    sr, K = 48000, 2
    rng = np.random.default_rng(13)
    x = (rng.standard_normal((K, 12000)) * 0.3).astype(np.float32)
    for w in (ol.WEIGHT_NONE, ol.WEIGHT_A, ol.WEIGHT_C):
        # A and C weighting have their poles at 20.6 Hz: the float32 recursion's own round-off
        # leaves a little more than 1e-5 between two correct evaluations
        tol = TOL if w == ol.WEIGHT_NONE else 3e-5
        bank = gpu.LoudnessBank(1, K)
        ref = ol.LoudnessMeter(K)
        for obj in (bank, ref):
            obj.set_sample_rate(sr); obj.set_weighting(w); obj.set_period(50.0)
        out = gpu.DeviceBuffer((1, 6000))
        bank.process(out, None, gpu.DeviceBuffer.from_host(x[:, :6000]), 6000)
        r1, _ = ref.process(x[:, :6000])
        assert np.abs(out.download()[0] - r1).max() <= tol * float(np.abs(r1).max()), w
        bank.clear(); ref.clear()
        bank.set_active(1, False); ref.set_active(1, False)
        bank.process(out, None, gpu.DeviceBuffer.from_host(x[:, 6000:]), 6000)
        r2, _ = ref.process(x[:, 6000:])
        assert np.abs(out.download()[0] - r2).max() <= tol * float(np.abs(r2).max()), w
        bank.close()


def test_disabled_channel_filter_freezes_until_reenabled(gpu):
    """A disabled channel's weighting filter is not run (LoudnessMeter.cpp:420-422): when the channel comes back, its
    filter continues from the memory it had when it was switched off, not from what the skipped samples would have left."""
    sr, K, n = 48000, 2, 4000
    rng = np.random.default_rng(14)
    x = (rng.standard_normal((K, 3 * n)) * 0.3).astype(np.float32)
    x[1] += 0.5                                              # DC through the K-weighting high-pass: long filter memory
    x[1, n:2 * n] -= 1.0                                     # ... and the opposite offset while the channel is off
    bank = gpu.LoudnessBank(1, K)
    ref = ol.LoudnessMeter(K)
    for obj in (bank, ref):
        obj.set_sample_rate(sr); obj.set_period(20.0); obj.set_link(1, 0.0)
    got, want, wch = [], [], []
    gch = []
    for blk, active in ((0, True), (1, False), (2, True)):
        for obj in (bank, ref):
            obj.set_active(1, active)
        out = gpu.DeviceBuffer((1, n)); ch = gpu.DeviceBuffer.from_host(np.zeros((K, n), np.float32))
        bank.process(out, ch, gpu.DeviceBuffer.from_host(x[:, blk * n:(blk + 1) * n]), n)
        got.append(out.download()[0]); gch.append(ch.download()[1])
        o, c = ref.process(x[:, blk * n:(blk + 1) * n])
        want.append(o); wch.append(c[1])
    got, want, gch, wch = map(np.concatenate, (got, want, gch, wch))
    peak = float(want.max())
    # DC into the 38 Hz high-pass of the K weighting: as with the A and C curves above, the float32 recursion's own
    # round-off leaves a little more than 1e-5 between two correct evaluations
    tol = 3e-5
    assert np.abs(got - want).max() <= tol * peak
    assert np.abs(gch - wch).max() <= tol * peak
    # the property itself: a filter that had kept running would start block 2 from a settled high-pass instead
    naive = ol.LoudnessMeter(K)
    naive.set_sample_rate(sr); naive.set_period(20.0); naive.set_link(1, 0.0)
    naive.process(x[:, :2 * n])
    naive.set_active(1, False); naive.set_active(1, True)
    _, c = naive.process(x[:, 2 * n:])
    assert np.abs(c[1][:200] - wch[2 * n:2 * n + 200]).max() > 0.05 * peak
    bank.close()


@pytest.mark.parametrize("blocks", [(4096,) * 24, (4096, 1000, 4096, 3333, 4096, 4095, 2049, 4096, 777) * 3])
def test_exact_resummation_inside_long_runs(gpu, blocks):
    """The exact re-summation of the window (refresh_rms(), LoudnessMeter.cpp:381-407) falls INSIDE the blocks here and
    is served from the segment sums kept beside the lines: several laps of the 16384-cell lines, block starts on and off
    the segment grid, against the oracle that re-sums cell by cell on the reference's schedule.  The third channel loses
    its input for a while (its line stands still, its window is still re-summed) and gets it back (the segment sums of
    its line are stale for one lap: the bank re-sums cell by cell until the line has been written all round)."""
    sr, M, K = 48000, 2, 3
    rng = np.random.default_rng(16)
    bank = gpu.LoudnessBank(M, K, 200.0)
    refs = [ol.LoudnessMeter(K, 200.0) for _ in range(M)]
    for obj in [bank] + refs:
        obj.set_sample_rate(sr); obj.set_period(200.0)
        obj.set_designation(0, ol.CHANNEL_LEFT); obj.set_designation(1, ol.CHANNEL_RIGHT); obj.set_designation(2, ol.CHANNEL_CENTER)
        obj.set_link(0, 0.0); obj.set_link(2, 0.5)
    level = 0.0
    for step, n in enumerate(blocks):
        if step == 5 or step == 17:
            for obj in [bank] + refs:
                obj.set_bound(2, False)
        if step == 9 or step == 19:
            for obj in [bank] + refs:
                obj.set_bound(2, True)
        amp = 0.3 if (step // 4) % 2 == 0 else 0.02             # level steps: a drifting running sum would show
        x = (rng.standard_normal((M * K, n)) * amp).astype(np.float32)
        out = gpu.DeviceBuffer((M, n)); ch = gpu.DeviceBuffer.from_host(np.full((M * K, n), -1.0, np.float32))
        bank.process(out, ch, gpu.DeviceBuffer.from_host(x), n)
        y, yc = out.download(), ch.download()
        for m in range(M):
            o, c = refs[m].process(x[m * K:(m + 1) * K])
            level = max(level, float(o.max()))
            assert np.abs(y[m] - o).max() <= TOL * level, (step, m, np.abs(y[m] - o).max() / level)
            for k in range(K):
                if refs[m].ch[k]["bound"]:
                    assert np.abs(yc[m * K + k] - c[k]).max() <= TOL * level, (step, m, k)
    bank.close()


def test_vector_and_scalar_block_kernels_agree(gpu, monkeypatch):
    """Blocks whose length, line position and period are multiples of four take the kernel with four samples per lane,
    anything else the one-sample kernel (MI_DSPU_TEST_PATH=loudness_scalar forces it): the same stream through both, several laps of the
    lines with the exact re-summation inside the blocks, must agree to float32 round-off of the scan order (and both with
    the oracle through the other tests)."""
    sr, M, K, n = 48000, 3, 2, 4096
    rng = np.random.default_rng(17)
    xs = [(rng.standard_normal((M * K, n)) * (0.3 if i % 3 else 0.03)).astype(np.float32) for i in range(12)]
    runs = []
    for scalar in (False, True):
        if scalar:
            monkeypatch.setenv("MI_DSPU_TEST_PATH", "loudness_scalar")
        else:
            monkeypatch.delenv("MI_DSPU_TEST_PATH", raising=False)
        bank = gpu.LoudnessBank(M, K, 400.0)
        bank.set_sample_rate(sr); bank.set_link(1, 0.4)
        outs = []
        for x in xs:
            out = gpu.DeviceBuffer((M, n)); ch = gpu.DeviceBuffer((M * K, n))
            bank.process(out, ch, gpu.DeviceBuffer.from_host(x), n)
            outs.append((out.download(), ch.download()))
        runs.append(outs)
        bank.close()
    peak = max(float(o.max()) for o, _ in runs[0])
    for (a, ac), (b, bc) in zip(*runs):
        assert np.abs(a - b).max() <= 2e-6 * peak and np.abs(ac - bc).max() <= 2e-6 * peak


@pytest.mark.parametrize("seed", range(8))
def test_random_operation_sequences(gpu, seed):
    """Differential stress of the momentary / short-term meter bank: period, weighting, designation, link and activity
    changes, clear() and ragged process() calls in random order, two meters of three channels."""
    rng = np.random.default_rng(15000 + seed)
    M, K, sr = 2, 3, 48000
    bank = gpu.LoudnessBank(M, K, 200.0)
    refs = [ol.LoudnessMeter(K, 200.0) for _ in range(M)]
    for obj in [bank] + refs:
        obj.set_sample_rate(sr)
    weight = ol.WEIGHT_K
    log = []
    recent = [np.zeros(0) for _ in range(M)]
    xpeak = np.zeros((M, K))                                 # |input| peak of the last calls: what the weighting filter's memory holds
    level2 = 0.0
    MS_TOL = 2e-5           # mean-square domain: 1e-5 of the amplitude at full level, looser only where the output is small
    for step in range(40):
        op = rng.choice(["process", "process", "process", "period", "weighting", "designation", "link", "active", "bound", "clear"])
        if op == "process":
            n = int(rng.choice([1, 100, 1023, 1024, 1025, 4096, 4097, int(rng.integers(1, 9000))]))
            x = (rng.standard_normal((M * K, n)) * 0.2).astype(np.float32)
            g = [None, 1.0, 0.5][int(rng.integers(0, 3))]      # None: process(out, count), the form that records loudness()
            out = gpu.DeviceBuffer((M, n)); ch = gpu.DeviceBuffer.from_host(np.full((M * K, n), -1.0, np.float32))
            bank.process(out, ch, gpu.DeviceBuffer.from_host(x), n, gain=g)
            y, yc = out.download(), ch.download()
            # the A, B, C and D curves have poles at 20 Hz: float32 round-off of the recursion
            tol = TOL if weight in (ol.WEIGHT_NONE, ol.WEIGHT_K) else 5e-5
            for m in range(M):
                held = [0 if cc["data"] is None else int(np.count_nonzero(cc["data"])) for cc in refs[m].ch]
                o, c = refs[m].process(x[m * K:(m + 1) * K], gain=g)
                # (the weighting filter's round-off is relative to the level in ITS memory, not to the few samples a window
                # that has just been cleared holds: 1.6 = the K curve's gain at the top of the band, +4 dB)
                xpeak[m] = np.maximum(0.5 * xpeak[m], np.abs(x[m * K:(m + 1) * K]).max(axis=1))
                fpk2 = [max(float(refs[m].ch[k]["data"].max()) if refs[m].ch[k]["data"] is not None else 0.0,
                            (1.6 * float(xpeak[m][k])) ** 2) for k in range(K)]
                # a call of a few samples has no meaningful peak of its own (and the square root magnifies the running
                # sum's round-off while the window is nearly empty): relate the errors to the last 1024 samples' peak
                recent[m] = np.concatenate([recent[m], np.abs(o), np.abs(c).max(axis=0)])[-2048:]
                peak = max(float(recent[m].max()), 1e-3)
                err = float(np.abs(y[m] - o).max())
                # While the window is nearly empty (a channel has just come back) the output is the square root of what
                # is left of the running sum: the round-off the sum has collected from its earlier, larger contents
                # (about 1e-7 of them, different in any two evaluation orders) is all there is.  The error is linear in
                # the mean-square domain, so it is judged there when the output is small.
                level2 = max(level2, float((o / (g or 1.0)).max()) ** 2)
                d_amp = np.abs(y[m].astype(np.float64) - o.astype(np.float64))
                d_ms = np.abs(y[m].astype(np.float64) ** 2 - o.astype(np.float64) ** 2) / (g or 1.0) ** 2
                # the mean-square criterion only where it belongs: samples whose output is below 3 % of the loudest level
                # seen (1e-3 in the mean-square domain) -- the nearly empty window; everywhere else the amplitude rule
                # holds, or the strict 1e-6 of the mean square
                small = (o.astype(np.float64) / (g or 1.0)) ** 2 < 1e-3 * level2
                # A window that has only just started to fill (every contributing line holds less than 1/16 of the period:
                # the channel was switched on or cleared a moment ago) averages nothing: the value IS the filtered signal,
                # which two float32 evaluations of the weighting filter reproduce only to the filter's own round-off.
                # |out_a - out_b| <= sqrt(sum_k w_k sum_window e_k^2 / N) (triangle inequality of the l2 norm) with
                # e_k <= the IIR rule of DESIGN.md section 4 applied to the line's peak -- that bound, sample by sample.
                N = refs[m].period
                live = [k for k in range(K) if refs[m].ch[k]["enabled"] and refs[m].ch[k]["bound"] and float(refs[m].ch[k]["weight"]) != 0.0]
                # (sample by sample: a long call that begins in a window which has just started to fill is judged by this rule
                # for its first samples only -- seed 22680 of the round-2 sweep: the ninth sample after a channel came back)
                filling = len(live) > 0 and all(held[k] + 1 <= N // 16 for k in live)
                if filling:
                    e_rel = max(TOL, IIR_REF_FACTOR * _weighting_noise(weight, sr))
                    j = np.arange(1, n + 1, dtype=np.float64)
                    l2 = sum(float(refs[m].ch[k]["weight"]) * fpk2[k] * np.minimum(held[k] + j, N) / N for k in live)
                    fill_bound = e_rel * (g or 1.0) * np.sqrt(l2)
                    fill_bound[max(0, N // 16 - max(held[k] for k in live)):] = 0.0
                else:
                    fill_bound = np.zeros(n)
                # the floor of the mean-square rule: a float32 running sum updated once per sample walks away from exact by
                # 2^-24 sqrt(updates) of its largest contents until the next exact re-summation, which comes every
                # max(4096, period / 4) samples (LoudnessMeter.cpp:381-407) -- 3.8e-6 at 4096
                ms_floor = 2.0 ** -24 * np.sqrt(max(ol.BUFFER_SIZE << 2, refs[m].period >> 2))
                # TWO rules, sample by sample; a sample is judged by the one that admits it more easily, and each rule's ledger
                # row only holds the samples it judged (so no row can exceed 1 in a green run):
                #   A  amplitude:    |gpu - oracle| <= max(tol x recent peak, window-filling bound)
                #   B  mean square:  |gpu^2 - oracle^2| / gain^2 <= (running-sum floor, or 2e-5 where the output is below 3 %
                #                    of the loudest level seen) x the loudest mean square seen
                allow_a = np.maximum(tol * peak, fill_bound)
                allow_b = np.where(small, max(MS_TOL, ms_floor), ms_floor) * level2
                ra, rb = d_amp / allow_a, d_ms / np.maximum(allow_b, 1e-300)
                by_a = ra <= rb
                if by_a.any():
                    record_parity("loudness A: |gpu - oracle| <= max(tol x recent peak, window-filling bound)", float(ra[by_a].max()), 1.0)
                if (~by_a).any():
                    record_parity("loudness B: |gpu^2 - oracle^2| <= (running-sum floor | 2e-5 below 3 %) x loudest mean square",
                                  float(rb[~by_a].max()), 1.0)
                bad = np.minimum(ra, rb) > 1.0
                i_bad = int(np.argmax(bad)) if bad.any() else 0
                assert not bad.any(), \
                    (seed, step, m, n, int(bad.sum()), d_amp[i_bad] / peak, d_ms[i_bad] / level2, i_bad, peak, level2, g, log[-8:],
                     y[m][max(0, i_bad - 20):i_bad + 4].tolist(), o[max(0, i_bad - 20):i_bad + 4].tolist())
                for k in range(K):
                    if refs[m].ch[k]["enabled"] and refs[m].ch[k]["bound"]:
                        level2 = max(level2, float((c[k] / (g or 1.0)).max()) ** 2)
                        # sample by sample like the meter's own value; a channel's output is the linked mix of the meter's
                        # value and the channel's own mean square, whose window may be the one that has just started to fill
                        dc = np.abs(yc[m * K + k].astype(np.float64) - c[k])
                        dcm = np.abs(yc[m * K + k].astype(np.float64) ** 2 - c[k].astype(np.float64) ** 2) / (g or 1.0) ** 2
                        cerr, cms = float(dc.max()), float(dcm.max())
                        lk = float(refs[m].ch[k]["link"])
                        jj = np.arange(1, n + 1, dtype=np.float64)
                        own = np.zeros(n)
                        if held[k] + 1 <= N // 16:
                            e_own = max(TOL, IIR_REF_FACTOR * _weighting_noise(weight, sr))
                            own = e_own * (g or 1.0) * np.sqrt(fpk2[k] * np.minimum(held[k] + jj, N) / N)
                            own[max(0, N // 16 - held[k]):] = 0.0
                        # (the link mixes the meter's value and the channel's own mean square, Loudness.cpp; links outside
                        # [0, 1] extrapolate, the bound follows with the absolute weights)
                        allowed = np.maximum(abs(lk) * np.maximum(tol * peak, fill_bound) + abs(1.0 - lk) * np.maximum(tol * peak, own),
                                             tol * peak)
                        small_c = (c[k].astype(np.float64) / (g or 1.0)) ** 2 < 1e-3 * level2
                        allow_cb = np.where(small_c, max(MS_TOL, ms_floor), ms_floor) * level2
                        rca, rcb = dc / allowed, dcm / np.maximum(allow_cb, 1e-300)
                        c_by_a = rca <= rcb
                        if c_by_a.any():
                            record_parity("loudness A: |gpu - oracle| <= max(tol x recent peak, window-filling bound)", float(rca[c_by_a].max()), 1.0)
                        if (~c_by_a).any():
                            record_parity("loudness B: |gpu^2 - oracle^2| <= (running-sum floor | 2e-5 below 3 %) x loudest mean square",
                                          float(rcb[~c_by_a].max()), 1.0)
                        assert bool(np.all(np.minimum(rca, rcb) <= 1.0)), (seed, step, m, k, cerr / peak, cms / level2, log[-8:])
                    else:
                        assert np.all(yc[m * K + k] == -1.0)
            np.testing.assert_allclose(bank.loudness(), [float(r.loud) for r in refs], rtol=0, atol=tol * 2.0)
        elif op == "period":
            # (not shorter: the running sum `ms += new - old` cancels catastrophically over a few samples and the square
            # root that follows magnifies its float32 round-off near zero -- two correct evaluations differ visibly)
            p = float(rng.choice([50.0, 120.0, 200.0, 400.0]))
            for obj in [bank] + refs:
                obj.set_period(p)
        elif op == "weighting":
            weight = int(rng.choice([ol.WEIGHT_NONE, ol.WEIGHT_K, ol.WEIGHT_K, ol.WEIGHT_A]))
            for obj in [bank] + refs:
                obj.set_weighting(weight)
        elif op == "designation":
            k, d = int(rng.integers(0, K)), int(rng.choice([ol.CHANNEL_LEFT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1, ol.CHANNEL_NONE]))
            for obj in [bank] + refs:
                obj.set_designation(k, d)
        elif op == "link":
            k, l = int(rng.integers(0, K)), float(rng.choice([0.0, 0.3, 1.0, 1.5, -0.5]))
            for obj in [bank] + refs:
                obj.set_link(k, l)
        elif op == "active":
            k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2))
            for obj in [bank] + refs:
                obj.set_active(k, a)
        elif op == "bound":
            k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2))
            for obj in [bank] + refs:
                obj.set_bound(k, a)
        else:
            for obj in [bank] + refs:
                obj.clear()
        log.append(str(op))
    bank.close()
