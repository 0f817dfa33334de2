"""Every bank on a caller's own non-blocking stream: same bits as on the default stream.  A non-blocking stream does not
wait for the null stream and is not waited for by it, so an internal launch or copy that went to the wrong stream races
with the rest of the call and shows up as a difference."""
import ctypes

import numpy as np
import pytest

from oracle import filter_design as fd
import workloads as wl

pytestmark = pytest.mark.gpu
C, N, CALLS = 64, 4096, 3


@pytest.fixture(scope="module")
def side_stream(gpu):
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0          # hipStreamNonBlocking
    yield s.value
    hip.hipStreamDestroy(s)


def _run(gpu, kind, st):
    rng = np.random.default_rng(33)
    calls = 6 if kind == "ilufs" else CALLS              # the integrated meter reads 0 until its first 400 ms block is full
    xs = [(rng.standard_normal((C, N)) * 0.25).astype(np.float32) for _ in range(calls)]
    outs = []
    buf = lambda shape=(C, N): gpu.DeviceBuffer(shape)                    # noqa: E731
    if kind == "biquad":
        b = gpu.BiquadBank(C, 8)
        q = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 3000.0, 0, 1.0, 0.75)
        for c in range(C):
            b.set_chains(c, q)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf())          # noqa: E731
    elif kind == "convolver":
        b = gpu.ConvolverBank(rng.standard_normal((C, 9000)).astype(np.float32), 11, stream=st)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf())          # noqa: E731
    elif kind == "equalizer":
        b = gpu.EqualizerBank(C, 4, 11); b.set_mode(2); b.set_sample_rate(48000)
        for c in range(C):
            b.set_params(0, fd.FLT_BT_RLC_BELL, 1, 500.0 + 100.0 * c, 1000.0, 2.0, 1.0, channel=c)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf())          # noqa: E731
    elif kind == "spectral":
        b = gpu.SpectralBank(C, 12); b.set_rank(11)
        b.bind_mask(np.linspace(0.0, 1.0, 2 << 11).astype(np.float32), stream=st)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf())          # noqa: E731
    elif kind == "analyzer":
        b = gpu.AnalyzerBank(C, 12, 48000, 1.0, 0)
        for what, v in ((b.SAMPLE_RATE, 48000), (b.RATE, 48000 / 2048.0), (b.RANK, 12), (b.WINDOW, 0), (b.REACTIVITY, 0.2), (b.SHIFT, 1.0)):
            b.configure(what, v)

        def step(d):
            b.process(d, N, stream=st)
            r = buf((2049,))
            b.reduce_bins(r, stream=st)
            return [r]
    elif kind == "delay":
        b = gpu.DelayBank(C, 6000)
        for c in range(C):
            b.set_delay(50 * c, channel=c)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf())          # noqa: E731
    elif kind == "loudness":
        b = gpu.LoudnessBank(C // 2, 2, 400.0); b.set_sample_rate(48000)
        step = lambda d: (lambda y, c: (b.process(y, c, d, N, stream=st), [y, c])[1])(buf((C // 2, N)), buf())   # noqa: E731
    elif kind == "ilufs":
        b = gpu.ILUFSBank(C // 2, 2, 10.0, 400.0); b.set_sample_rate(48000)
        step = lambda d: (lambda y: (b.process(y, d, N, stream=st), [y])[1])(buf((C // 2, N)))                    # noqa: E731
    elif kind == "splitter":
        b = gpu.SplitterBank(C, 12, 3); b.set_rank(11); b.set_chunk_rank(9)
        b.bind_copy(0, stream=st)
        b.bind_mask(1, np.linspace(1.0, 0.0, 1 << 11).astype(np.float32), stream=st)
        b.bind_mask(2, np.linspace(0.0, 1.0, 1 << 11).astype(np.float32), stream=st)
        step = lambda d: (lambda ys: (b.process(ys, d, N, stream=st), ys)[1])([buf(), buf(), buf()])             # noqa: E731
    else:
        b = gpu.CrossoverBank(C, 3); b.set_sample_rate(48000)
        for i, f in enumerate((500.0, 4000.0)):
            b.set_slope(i, 2); b.set_frequency(i, f)
        step = lambda d: (lambda ys: (b.process(ys, d, N, stream=st), ys)[1])([buf(), buf(), buf()])             # noqa: E731
    for x in xs:
        d = gpu.DeviceBuffer.from_host(x, stream=st)
        outs.extend(o.download(stream=st) for o in step(d))
    b.close()
    return outs


@pytest.mark.parametrize("kind", ["biquad", "convolver", "equalizer", "spectral", "analyzer", "delay", "loudness", "ilufs",
                                  "splitter", "crossover"])
def test_side_stream_gives_the_same_bits(gpu, side_stream, kind):
    ref = _run(gpu, kind, None)
    for _ in range(2):
        got = _run(gpu, kind, side_stream)
        assert len(got) == len(ref)
        assert max(float(np.abs(r).max()) for r in ref) > 0.0
        for a, r in zip(got, ref):
            assert np.isfinite(r).all()
            np.testing.assert_array_equal(a, r)


def test_two_banks_on_two_streams_do_not_disturb_each_other(gpu):
    """Two convolver banks and two spectral banks fed alternately on two non-blocking streams, nothing waited for until the
    end: each gives what it gives alone (the banks share only read-only tables)."""
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        streams.append(s.value)
    rng = np.random.default_rng(44)
    irs = [rng.standard_normal((C, 20000)).astype(np.float32) for _ in range(2)]
    xs = [[(rng.standard_normal((C, N)) * 0.25).astype(np.float32) for _ in range(4)] for _ in range(2)]
    masks = [np.linspace(0.0, 1.0, 2 << 11).astype(np.float32), np.linspace(1.0, 0.0, 2 << 11).astype(np.float32)]

    def make(i, st):
        cv = gpu.ConvolverBank(irs[i], 11, stream=st)
        sp = gpu.SpectralBank(C, 12); sp.set_rank(11); sp.bind_mask(masks[i], stream=st)
        return cv, sp

    alone = []
    for i in range(2):
        cv, sp = make(i, None)
        res = []
        for x in xs[i]:
            d, y, z = gpu.DeviceBuffer.from_host(x), gpu.DeviceBuffer((C, N)), gpu.DeviceBuffer((C, N))
            cv.process(y, d, N); sp.process(z, y, N)
            res.append(z.download())
        alone.append(res)
        cv.close(); sp.close()

    banks = [make(i, streams[i]) for i in range(2)]
    ins = [[gpu.DeviceBuffer.from_host(x, stream=streams[i]) for x in xs[i]] for i in range(2)]
    mids = [[gpu.DeviceBuffer((C, N)) for _ in range(4)] for _ in range(2)]
    outs = [[gpu.DeviceBuffer((C, N)) for _ in range(4)] for _ in range(2)]
    for k in range(4):
        for i in range(2):
            banks[i][0].process(mids[i][k], ins[i][k], N, stream=streams[i])
            banks[i][1].process(outs[i][k], mids[i][k], N, stream=streams[i])
    for i in range(2):
        for k in range(4):
            np.testing.assert_array_equal(outs[i][k].download(stream=streams[i]), alone[i][k])
        banks[i][0].close(); banks[i][1].close()
    for s in streams:
        hip.hipStreamDestroy(ctypes.c_void_p(s))


def test_biquad_bank_inside_a_hip_graph(gpu):
    """The biquad bank keeps everything that changes from call to call (filter memory) on the device and takes no host
    decision in a steady-state process(): a run of calls can be captured once into a hipGraph and replayed.  Three replays
    of a four-call graph equal twelve eager calls, bit for bit."""
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    st = s.value
    rng = np.random.default_rng(55)
    q = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 3000.0, 0, 1.0, 0.75)
    xs = [(rng.standard_normal((C, N)) * 0.25).astype(np.float32) for _ in range(4)]

    def bank():
        b = gpu.BiquadBank(C, 8)
        for c in range(C):
            b.set_chains(c, q)
        b.commit(st)
        return b

    eager = bank()
    ref = []
    for rep in range(3):
        for x in xs:
            d, y = gpu.DeviceBuffer.from_host(x, stream=st), gpu.DeviceBuffer((C, N))
            eager.process(y, d, N, stream=st)
            ref.append(y.download(stream=st))
    eager.close()

    b = bank()
    ins = [gpu.DeviceBuffer.from_host(x, stream=st) for x in xs]
    outs = [gpu.DeviceBuffer((C, N)) for _ in xs]
    graph, exe = ctypes.c_void_p(), ctypes.c_void_p()
    del d, y                                    # (a buffer released while the stream captures would end the capture)
    import gc
    gc.collect()
    assert hip.hipStreamBeginCapture(s, 0) == 0
    for k in range(4):
        b.process(outs[k], ins[k], N, stream=st)
    assert hip.hipStreamEndCapture(s, ctypes.byref(graph)) == 0
    assert hip.hipGraphInstantiate(ctypes.byref(exe), graph, None, None, ctypes.c_size_t(0)) == 0
    for rep in range(3):
        assert hip.hipGraphLaunch(exe, s) == 0
        for k, y in enumerate(outs):
            np.testing.assert_array_equal(y.download(stream=st), ref[rep * 4 + k])
    hip.hipGraphExecDestroy(exe); hip.hipGraphDestroy(graph)
    b.close()
    hip.hipStreamDestroy(s)
