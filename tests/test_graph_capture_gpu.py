"""hipGraph capture of every bank's process().

Banks without positions (biquad, crossover, dynamic filters) replay any captured run of calls.  Banks that keep ring
positions on the host hand them to their kernels by value, so a captured run is right for every replay exactly when it
brings the bank back to the positions it started from: mi_dspu_graph_end_capture() checks that and refuses anything else,
and a capture started behind the library's back is refused by process() itself.  Every case below captures the shortest
accepted run (a position period), replays it three times and compares every output bit for bit with a twin bank driven
by eager calls; shorter runs must have been refused where the bank has a period longer than one call."""
import ctypes
import gc

import numpy as np
import pytest

import workloads as wl
from oracle import filter_design as fd

pytestmark = pytest.mark.gpu


class Case:
    """make() -> bank; call(bank, d_in, d_outs, n, stream); in_shape / out_shapes for one call."""
    def __init__(self, name, make, call, in_shape, out_shapes, min_period=1):
        self.name, self.make, self.call = name, make, call
        self.in_shape, self.out_shapes, self.min_period = in_shape, out_shapes, min_period


def _stream():
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return hip, s


def _run(gpu, case, max_k=12, replays=3, seed=77):
    hip, s = _stream()
    st = s.value
    rng = np.random.default_rng(seed)
    xs = [(rng.standard_normal(case.in_shape) * 0.25).astype(np.float32) for _ in range(max_k)]
    refused = []
    accepted = None
    for K in range(1, max_k + 1):
        bank = case.make(st)
        ins = [gpu.DeviceBuffer.from_host(xs[k], stream=st) for k in range(K)]
        outs = [[gpu.DeviceBuffer(shape) for shape in case.out_shapes] for _ in range(K)]
        for k in range(K):                                   # one eager run first: settings applied, scratch allocated
            case.call(bank, ins[k], outs[k], st)
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
        gc.collect()
        gc.disable()
        try:
            gpu.check(gpu.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(st)))
            for k in range(K):
                case.call(bank, ins[k], outs[k], st)
            exe = ctypes.c_void_p()
            rc = gpu.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(st), ctypes.byref(exe))
        finally:
            gc.enable()
        if rc != 0:
            msg = gpu.lib.mi_dspu_last_error().decode()
            assert "starting positions" in msg, msg
            refused.append(K)
            bank.close()
            continue
        accepted = (K, bank, ins, outs, exe)
        break
    assert accepted is not None, (case.name, "no run of up to %d calls was accepted" % max_k, refused)
    K, bank, ins, outs, exe = accepted
    assert K >= case.min_period, (case.name, K, refused)
    assert refused == list(range(1, K)), (case.name, K, refused)

    # the twin: the same calls, eagerly
    twin = case.make(st)
    ref = []
    for rep in range(1 + replays):
        for k in range(K):
            o = [gpu.DeviceBuffer(shape) for shape in case.out_shapes]
            case.call(twin, ins[k], o, st)
            ref.append([b.download(stream=st) for b in o])
    for rep in range(replays):
        gpu.check(gpu.lib.mi_dspu_graph_launch(exe, ctypes.c_void_p(st)))
        for k in range(K):
            for j, b in enumerate(outs[k]):
                np.testing.assert_array_equal(b.download(stream=st), ref[(1 + rep) * K + k][j],
                                              err_msg="%s: replay %d call %d output %d" % (case.name, rep, k, j))
    gpu.lib.mi_dspu_graph_destroy(exe)
    twin.close()
    bank.close()
    hip.hipStreamDestroy(s)
    return K


def test_delay_bank_replays_over_a_lap_of_the_line(gpu):
    C, n = 6, 512

    def make(st):
        b = gpu.DelayBank(C, 1500)                           # size = align(1500 + 512, 512) = 2048: four calls per lap
        for c in range(C):
            b.set_delay(100 + 200 * c, channel=c)
        return b

    K = _run(gpu, Case("delay", make, lambda b, x, o, st: b.process(o[0], x, n, gain=0.5, stream=st), (C, n), [(C, n)], 4))
    assert K == 4


def test_ring_bank_replays_over_a_lap(gpu):
    C, n = 4, 256

    def call(b, x, o, st):
        b.append(x, n, stream=st)
        b.get(o[0], 300, n, stream=st)

    K = _run(gpu, Case("ring", lambda st: gpu.RingBank(C, 1024), call, (C, n), [(C, n)], 4))
    assert K == 4


def test_convolver_bank_replays_over_its_ring_of_frames(gpu):
    C, rank, taps = 4, 8, 1024
    irs = (np.random.default_rng(3).standard_normal((C, taps)) * 0.1).astype(np.float32)
    frame = 1 << (rank - 1)
    K = _run(gpu, Case("convolver", lambda st: gpu.ConvolverBank(irs, rank, stream=st),
                       lambda b, x, o, st: b.process(o[0], x, frame, stream=st), (C, frame), [(C, frame)], 2))
    assert K > 1


def test_convolver_block_stream_replays(gpu):
    """A stream of 256-sample calls on a partitioned bank with frames of 1024 (rank 11: four block kernels and one folding
    completion per frame): the captured run must cover whole laps of the frame ring -- two frames, eight calls -- and its
    replays equal the same calls made eagerly, bit for bit."""
    C, rank, taps = 5, 11, 3000                               # three partitions of 1024: a ring of two frames
    irs = (np.random.default_rng(4).standard_normal((C, taps)) * 0.1).astype(np.float32)
    K = _run(gpu, Case("convolver block stream", lambda st: gpu.ConvolverBank(irs, rank, stream=st),
                       lambda b, x, o, st: b.process(o[0], x, 256, stream=st), (C, 256), [(C, 256)], 8), max_k=16)
    assert K == 8


def test_equalizer_fir_replays(gpu):
    C, n = 4, 512

    def make(st):
        eq = gpu.EqualizerBank(C, 2, 9)
        eq.set_mode(gpu.EqualizerBank.FIR)
        eq.set_sample_rate(48000)
        eq.set_params(0, fd.FLT_BT_RLC_BELL, 1, 1000.0, 1000.0, 2.0, 2.0)
        eq.set_params(1, fd.FLT_BT_RLC_HISHELF, 1, 6000.0, 6000.0, 0.5, 0.0)
        return eq

    # the delay line in front of the convolver is 9216 cells long: 18 calls of 512 samples per lap
    K = _run(gpu, Case("equalizer", make, lambda b, x, o, st: b.process(o[0], x, n, stream=st), (C, n), [(C, n)]), max_k=20)
    assert K == 18


def test_analyzer_bank_replays_over_an_even_number_of_strobes(gpu):
    C, rank = 8, 8

    def make(st):
        an = gpu.AnalyzerBank(C, rank, 48000, 100.0)
        an.configure(an.SAMPLE_RATE, 48000); an.configure(an.RANK, rank); an.configure(an.RATE, 187.5)
        an.configure(an.REACTIVITY, 0.2)
        return an
    period = 256                                             # 48000 / 187.5

    def call(b, x, o, st):
        b.process(x, period, stream=st)
        b.reduce_bins(o[0], stream=st)

    bins = (1 << (rank - 1)) + 1
    K = _run(gpu, Case("analyzer", make, call, (C, period), [(bins,)], 2), max_k=16)
    assert K % 2 == 0                                        # the two spectrum buffers swap roles at every strobe


def test_spectral_bank_replays_every_call(gpu):
    C, rank = 4, 8

    def make(st):
        b = gpu.SpectralBank(C, rank)
        b.set_rank(rank)
        return b
    n = 1 << rank
    K = _run(gpu, Case("spectral", make, lambda b, x, o, st: b.process(o[0], x, n, stream=st), (C, n), [(C, n)]))
    assert K == 1


def test_splitter_bank_replays(gpu):
    C, rank, H = 4, 8, 2
    n = 1 << (rank - 1)                                      # one hop per call: the analysis buffers swap every hop

    def make(st):
        sp = gpu.SplitterBank(C, rank, H)
        sp.set_rank(rank)
        sp.bind_mask(0, np.linspace(1.0, 0.0, 1 << rank).astype(np.float32), stream=st)
        sp.bind_mask(1, np.linspace(0.0, 1.0, 1 << rank).astype(np.float32), stream=st)
        return sp

    K = _run(gpu, Case("splitter", make, lambda b, x, o, st: b.process(o, x, n, stream=st), (C, n), [(C, n), (C, n)], 2))
    assert K == 2


def test_loudness_and_ilufs_banks_replay(gpu):
    M, Kc, n = 2, 2, 4096

    def make_lm(st):
        lm = gpu.LoudnessBank(M, Kc, 100.0)                  # 100 ms: lines of 8192 cells, re-summation every 4096 samples
        lm.set_sample_rate(48000, stream=st)
        return lm

    K = _run(gpu, Case("loudness", make_lm, lambda b, x, o, st: b.process(o[0], None, x, n, stream=st),
                       (M * Kc, n), [(M, n)], 2))
    assert K == 2

    def make_im(st):
        im = gpu.ILUFSBank(M, Kc, 5.0, 400.0)                # gating blocks of four quarters of 4800 samples
        im.set_sample_rate(48000, stream=st)
        return im

    K = _run(gpu, Case("ilufs", make_im, lambda b, x, o, st: b.process(o[0], x, 4800, stream=st), (M * Kc, 4800), [(M, 4800)], 4))
    assert K == 4


def test_banks_without_positions_replay_any_run(gpu):
    C, n = 8, 4096

    def make_xo(st):
        xo = gpu.CrossoverBank(C, 3)
        xo.set_sample_rate(48000)
        for i, f in enumerate((300.0, 3000.0)):
            xo.set_slope(i, 2); xo.set_frequency(i, f)
        return xo

    K = _run(gpu, Case("crossover", make_xo, lambda b, x, o, st: b.process(o, x, n, stream=st), (C, n), [(C, n)] * 3))
    assert K == 1


def test_capture_behind_the_librarys_back_is_refused(gpu):
    """A positional bank on a stream captured with hipStreamBeginCapture directly: process() refuses (nothing could check
    the positions at the end); a biquad bank is fine there (tests/test_streams_gpu.py)."""
    hip, s = _stream()
    st = s.value
    C, n = 4, 256
    b = gpu.DelayBank(C, 1000)
    b.set_delay(300)
    x, y = gpu.DeviceBuffer.from_host(np.ones((C, n), np.float32), stream=st), gpu.DeviceBuffer((C, n))
    b.process(y, x, n, stream=st)
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
    graph = ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(s, 0) == 0
    with pytest.raises(gpu.MiError) as e:
        b.process(y, x, n, stream=st)
    assert "mi_dspu_graph_begin_capture" in str(e.value)
    hip.hipStreamEndCapture(s, ctypes.byref(graph))
    if graph.value:
        hip.hipGraphDestroy(graph)
    head = b.get()["head"]
    b.process(y, x, n, stream=st)                            # the refused call changed nothing
    assert b.get()["head"] == (head + n) % b.get()["size"]
    b.close()
    hip.hipStreamDestroy(s)


def test_refused_capture_leaves_the_bank_consistent(gpu):
    """end_capture() refuses a run that does not close a position period -- and by then the bank's host-side positions have
    moved although no kernel ran.  The library executes the captured calls once in that case, so the eager calls that follow
    (bench.py's fallback) continue the stream exactly as if every call had been eager."""
    hip, s = _stream()
    st = s.value
    C, n = 5, 512
    rng = np.random.default_rng(5)
    xs = [(rng.standard_normal((C, n)) * 0.25).astype(np.float32) for _ in range(6)]

    def make():
        b = gpu.DelayBank(C, 1500)                           # a lap is four calls of 512
        for c in range(C):
            b.set_delay(90 + 170 * c, channel=c)
        return b

    bank, twin = make(), make()
    ins = [gpu.DeviceBuffer.from_host(x, stream=st) for x in xs]
    outs = [gpu.DeviceBuffer((C, n)) for _ in xs]
    bank.process(outs[0], ins[0], n, gain=0.5, stream=st)                  # eager
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
    gpu.check(gpu.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(st)))
    for k in (1, 2):                                                       # two calls: half a lap, refused
        bank.process(outs[k], ins[k], n, gain=0.5, stream=st)
    exe = ctypes.c_void_p()
    rc = gpu.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(st), ctypes.byref(exe))
    assert rc == -5 and not exe.value, (rc, gpu.lib.mi_dspu_last_error())
    assert b"executed ONCE" in gpu.lib.mi_dspu_last_error()
    for k in (3, 4, 5):                                                    # eager again
        bank.process(outs[k], ins[k], n, gain=0.5, stream=st)
    for k in range(6):
        o = gpu.DeviceBuffer((C, n))
        twin.process(o, ins[k], n, gain=0.5, stream=st)
        np.testing.assert_array_equal(outs[k].download(stream=st), o.download(stream=st), err_msg="call %d" % k)
    bank.close(); twin.close()
    hip.hipStreamDestroy(s)


def _blocks(buf, k, cells):
    """Device addresses of the k blocks of `cells` floats a DeviceBuffer holds one behind the other."""
    return [buf.ptr + 4 * cells * i for i in range(k)]


def test_runs_of_blocks_in_one_launch_replay(gpu):
    """The calls that carry several blocks (frames) per launch under capture: the biquad bank's process_blocks (no positions:
    any run), the equalizer's (one lap of its delay line in ONE call, half a lap in two) and the analyzer's
    process_reduce_frames (strobes in pairs: every call brings the spectrum buffers back)."""
    C, n, Kb = 8, 4096, 6
    coefs, _ = wl.c2_coefficients(C)

    def make_bq(st):
        b = gpu.BiquadBank(C, 8)
        b.set_all_chains(coefs, clear=True)
        return b

    K = _run(gpu, Case("biquad blocks", make_bq,
                       lambda b, x, o, st: b.process_blocks(_blocks(o[0], Kb, C * n), _blocks(x, Kb, C * n), n, stream=st),
                       (Kb, C, n), [(Kb, C, n)]), max_k=2)
    assert K == 1

    Ce, ne = 4, 512

    def make_eq(st):
        eq = gpu.EqualizerBank(Ce, 2, 9)
        eq.set_mode(gpu.EqualizerBank.FIR)
        eq.set_sample_rate(48000)
        eq.set_params(0, fd.FLT_BT_RLC_BELL, 1, 1000.0, 1000.0, 2.0, 2.0)
        eq.set_params(1, fd.FLT_BT_RLC_HISHELF, 1, 6000.0, 6000.0, 0.5, 0.0)
        return eq

    for kb, want in ((18, 1), (9, 2)):                       # the delay line in front of the convolver: 18 blocks of 512 per lap
        K = _run(gpu, Case("equalizer blocks", make_eq,
                           lambda b, x, o, st: b.process_blocks(_blocks(o[0], kb, Ce * ne), _blocks(x, kb, Ce * ne), ne, stream=st),
                           (kb, Ce, ne), [(kb, Ce, ne)], want), max_k=4)
        assert K == want

    # the spectral bank's and the splitter's runs of blocks: the positions (a frame in hand) are where they were after every call
    Cs, rs, ns, Ks = 4, 9, 512, 3

    def make_sp(st):
        b = gpu.SpectralBank(Cs, rs)
        b.set_rank(rs)
        b.bind_mask(np.linspace(1.0, 0.3, (1 << (rs - 1)) + 1).astype(np.float32))
        return b

    K = _run(gpu, Case("spectral blocks", make_sp,
                       lambda b, x, o, st: b.process_blocks(_blocks(o[0], Ks, Cs * ns), _blocks(x, Ks, Cs * ns), ns, stream=st),
                       (Ks, Cs, ns), [(Ks, Cs, ns)]), max_k=2)
    assert K == 1

    def make_spl(st):
        b = gpu.SplitterBank(Cs, rs, 2)
        b.bind_mask(0, np.linspace(1.0, 0.0, 1 << rs).astype(np.float32))
        b.bind_mask(1, np.linspace(0.0, 1.0, 1 << rs).astype(np.float32))
        return b

    def call_spl(b, x, o, st):
        outs = [[o[0].ptr + 4 * Cs * ns * k, o[1].ptr + 4 * Cs * ns * k] for k in range(Ks)]
        b.process_blocks(outs, _blocks(x, Ks, Cs * ns), ns, stream=st)

    K = _run(gpu, Case("splitter blocks", make_spl, call_spl, (Ks, Cs, ns), [(Ks, Cs, ns), (Ks, Cs, ns)], 2), max_k=2)
    assert K == 2                                            # (a launch swaps the roles of the two analysis buffers)

    Ca, rank, F = 8, 10, 6
    period = 512                                             # half a frame: the strobes of a run go out in one launch
    bins = (1 << (rank - 1)) + 1

    def make_an(st):
        an = gpu.AnalyzerBank(Ca, rank, 49600, 50.0)         # ring of 1024 + 2 * 992 + 64 = 3072 cells: a call of six hops is a lap
        an.configure(an.SAMPLE_RATE, 49600); an.configure(an.RANK, rank); an.configure(an.RATE, 96.875)
        an.configure(an.REACTIVITY, 0.2)
        return an

    K = _run(gpu, Case("analyzer frames", make_an,
                       lambda b, x, o, st: b.process_reduce_frames(_blocks(x, F, Ca * period), period, o[0], stream=st),
                       (F, Ca, period), [(F, bins)]), max_k=3)
    assert K == 1


def test_convolver_bank_replays_after_a_batch_has_grown_its_ring(gpu):
    """mi_convolver_bank_process_blocks grows the ring of frames once (room for a batch of sixteen next to the frames its tails
    take): frame-by-frame calls captured afterwards repeat over a lap of the grown ring, bit for bit the eager calls."""
    C, rank, taps = 3, 10, 2000                              # four partitions of 512: a ring of 3, grown to 19
    frame = 1 << (rank - 1)
    irs = (np.random.default_rng(8).standard_normal((C, taps)) * 0.1).astype(np.float32)
    x0 = (np.random.default_rng(9).standard_normal((2, C, frame)) * 0.25).astype(np.float32)

    def make(st):
        b = gpu.ConvolverBank(irs, rank, stream=st)
        ins = [gpu.DeviceBuffer.from_host(x0[k], stream=st) for k in range(2)]
        outs = [gpu.DeviceBuffer((C, frame)) for _ in range(2)]
        b.process_blocks(outs, ins, frame, stream=st)
        gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
        return b

    K = _run(gpu, Case("convolver after a batch", make, lambda b, x, o, st: b.process(o[0], x, frame, stream=st),
                       (C, frame), [(C, frame)], 19), max_k=20)
    assert K == 19


def test_convolver_process_blocks_under_capture_is_the_loop_of_calls(gpu):
    """Under capture mi_convolver_bank_process_blocks does not batch (a batch may have to allocate and to grow the ring): the
    frames go one by one, and a captured lap of the ring replays bit for bit."""
    C, rank, taps = 3, 10, 2000                              # four partitions of 512: a ring of three frames, nineteen once the
    frame = 1 << (rank - 1)                                  # eager call in front of the capture has been through as batches
    irs = (np.random.default_rng(12).standard_normal((C, taps)) * 0.1).astype(np.float32)
    Kb = 19

    def call(b, x, o, st):
        b.process_blocks(_blocks(o[0], Kb, C * frame), _blocks(x, Kb, C * frame), frame, stream=st)

    K = _run(gpu, Case("convolver blocks", lambda st: gpu.ConvolverBank(irs, rank, stream=st), call, (Kb, C, frame), [(Kb, C, frame)]), max_k=3)
    assert K == 1


def test_graph_captured_before_the_ring_grew_is_refused(gpu):
    """a graph captured on a convolver bank holds the ring's address, size and slot in its launches; the bank's first
    batch of frames re-makes the ring.  The library keeps every captured bank's epoch with the executable graph: the stale
    graph is refused at mi_dspu_graph_launch (MI_ESTATE, nothing launched), a graph captured afterwards replays bit for bit."""
    hip, s = _stream()
    st = s.value
    C, rank, taps = 3, 10, 2000                              # four partitions of 512: a ring of 3 frames, 19 after a batch
    frame = 1 << (rank - 1)
    rng = np.random.default_rng(31)
    irs = (rng.standard_normal((C, taps)) * 0.1).astype(np.float32)
    xs = [(rng.standard_normal((C, frame)) * 0.25).astype(np.float32) for _ in range(3)]
    bank = gpu.ConvolverBank(irs, rank, stream=st)
    ins = [gpu.DeviceBuffer.from_host(x, stream=st) for x in xs]
    outs = [gpu.DeviceBuffer((C, frame)) for _ in range(3)]
    for k in range(3):
        bank.process(outs[k], ins[k], frame, stream=st)
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
    gpu.check(gpu.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(st)))
    for k in range(3):                                       # a lap of the ring of three
        bank.process(outs[k], ins[k], frame, stream=st)
    exe = ctypes.c_void_p()
    gpu.check(gpu.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(st), ctypes.byref(exe)))
    gpu.check(gpu.lib.mi_dspu_graph_launch(exe, ctypes.c_void_p(st)))          # fine so far
    bank.process_blocks(outs[:2], ins[:2], frame, stream=st)                    # the first batch: the ring is re-made
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
    before = [o.download(stream=st) for o in outs]
    rc = gpu.lib.mi_dspu_graph_launch(exe, ctypes.c_void_p(st))
    assert rc == -5, rc                                      # MI_ESTATE (include/mi_dspu.h:40)
    assert "capture again" in gpu.lib.mi_dspu_last_error().decode()
    gpu.check(gpu.lib.mi_dspu_stream_synchronize(ctypes.c_void_p(st)))
    for o, w in zip(outs, before):
        np.testing.assert_array_equal(o.download(stream=st), w)                 # nothing ran
    gpu.lib.mi_dspu_graph_destroy(exe)
    assert bank.faults(stream=st) == 0
    bank.close()
    hip.hipStreamDestroy(s)
