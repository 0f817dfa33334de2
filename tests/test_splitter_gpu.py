"""GPU parity of mi_splitter_bank_* (lsp::dspu::SpectralSplitter / the engine of FFTCrossover) against the CPU oracle,
through the C-ABI."""
import ctypes

import os

import numpy as np
import pytest

from oracle import splitter as osp

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle_run(rank, chunk, phase, handlers, x, calls):
    """handlers: list of None (unbound) | "copy" | mask array.  Returns [handler][channels][n]."""
    C, n = x.shape
    outs = np.zeros((len(handlers), C, n), np.float32)
    for c in range(C):
        sp = osp.SpectralSplitter(rank, len(handlers))
        sp.set_chunk_rank(chunk); sp.set_phase(phase)
        pos = [0] * len(handlers)
        for i, h in enumerate(handlers):
            if h is None:
                continue

            def sink(s, first, count, i=i, c=c):
                outs[i, c, pos[i]:pos[i] + count] = s; pos[i] += count
            if isinstance(h, str):
                sp.bind(i, None, sink)
            else:
                m = np.asarray(h, np.float32)
                m = m if m.ndim == 1 else m[c]

                def func(spec, r, m=m):
                    spec[0::2] *= m; spec[1::2] *= m
                    return spec
                sp.bind(i, func, sink)
        p = 0
        for k in calls:
            sp.process(x[c, p:p + k], k); p += k
    return outs


def _gpu_run(gpu, rank, chunk, phase, handlers, x, calls, max_rank=None):
    C, n = x.shape
    bank = gpu.SplitterBank(C, max_rank or rank, len(handlers))
    bank.set_rank(rank); bank.set_chunk_rank(chunk); bank.set_phase(phase)
    for i, h in enumerate(handlers):
        if h is None:
            continue
        if isinstance(h, str):
            bank.bind_copy(i)
        else:
            bank.bind_mask(i, h)
    got = np.zeros((len(handlers), C, n), np.float32)
    p = 0
    for k in calls:
        bufs = [gpu.DeviceBuffer.from_host(np.full((C, k), -7.0, np.float32)) if h is not None else None for h in handlers]
        bank.process(bufs, gpu.DeviceBuffer.from_host(x[:, p:p + k]), k)
        for i, b in enumerate(bufs):
            if b is not None:
                got[i, :, p:p + k] = b.download()
        p += k
    lat = bank.latency()
    bank.close()
    return got, lat


@pytest.mark.parametrize("rank,chunk,phase,calls", [
    (5, 0, 0.0, (300,)), (8, 0, 0.0, (1000, 24)), (9, 7, 0.0, (100, 3, 700, 197)), (10, 8, 0.5, (333, 1667)),
    (12, 10, 0.0, (8192, 5000)), (13, 0, 0.25, (20000,)), (12, 5, 1.0, (6000,)), (14, 12, 0.0, (40000,)),
    (15, 0, 0.0, (70000, 1234)), (16, 13, 0.5, (70000, 5000)), (15, 14, 0.25, (40000,)),     # frames above the LDS limit
])
def test_masks_and_copy_handlers_match_oracle(gpu, rank, chunk, phase, calls):
    """Several bands in one pass: a shared symmetric mask, per-channel masks, an ASYMMETRIC real mask (only the real part
    of the inverse is kept, so its even part acts), a handler without a spectral function, and an unbound slot."""
    C, n = 3, sum(calls)
    N = 1 << rank
    rng = np.random.default_rng(rank * 100 + chunk)
    x = (rng.standard_normal((C, n)) * 0.5).astype(np.float32)
    sym = osp.hipass_fft_set(2000.0, -24.0, 48000.0, rank)
    per = np.stack([osp.lopass_fft_set(500.0 * (c + 1), -32.0, 48000.0, rank) for c in range(C)])
    asym = rng.uniform(0.0, 1.5, N).astype(np.float32)
    handlers = [sym, None, per, "copy", asym]
    want = _oracle_run(rank, chunk, phase, handlers, x, calls)
    got, lat = _gpu_run(gpu, rank, chunk, phase, handlers, x, calls)
    assert lat == 1 << (min(max(chunk, 5), rank) if chunk > 0 else rank)
    peak = float(np.abs(x).max())
    for i, h in enumerate(handlers):
        if h is None:
            continue
        err = float(np.abs(got[i] - want[i]).max())
        assert err <= TOL * max(peak, float(np.abs(want[i]).max())), (i, err)
        assert float(np.abs(want[i]).max()) > 0.01


@pytest.mark.parametrize("rank", [6, 8, 9, 10, 11, 12, 13])
def test_hops_of_a_call_sharing_one_launch(gpu, rank, monkeypatch):
    """A call that brings several whole frames to a splitter whose listening handlers are all masks runs its hops in ONE
    launch (splitter_hop_kernel, hops > 1): against the oracle, and bit for bit against one launch per hop
    (MI_DSPU_TEST_PATH=splitter_hop_launches); calls whose blocks cannot be read as pairs (odd position / odd row stride), a call of
    silence and the calls that follow see the same state either way."""
    C, F = 3, 1 << (rank - 1)
    rng = np.random.default_rng(900 + rank)
    sym = osp.hipass_fft_set(2000.0, -24.0, 48000.0, rank)
    per = np.stack([osp.lopass_fft_set(500.0 * (c + 1), -32.0, 48000.0, rank) for c in range(C)])
    asym = rng.uniform(0.0, 1.5, 1 << rank).astype(np.float32)
    handlers = [sym, None, per, asym]
    calls = (4 * F, F // 2, F // 2 + 3 * F, 2 * F + 1, 3, 5 * F, F - 4, 2 * F)
    x = (rng.standard_normal((C, sum(calls))) * 0.5).astype(np.float32)
    want = _oracle_run(rank, 0, 0.0, handlers, x, calls)
    got, _ = _gpu_run(gpu, rank, 0, 0.0, handlers, x, calls)
    monkeypatch.setenv("MI_DSPU_TEST_PATH", "splitter_hop_launches")
    one, _ = _gpu_run(gpu, rank, 0, 0.0, handlers, x, calls)
    monkeypatch.delenv("MI_DSPU_TEST_PATH")
    for i, h in enumerate(handlers):
        if h is None:
            continue
        err = float(np.abs(got[i] - want[i]).max())
        assert err <= TOL * max(float(np.abs(x).max()), float(np.abs(want[i]).max())), (i, err)
        assert np.array_equal(got[i], one[i]), i
    # silence through the same path, then signal again
    bank = gpu.SplitterBank(C, rank, 1)
    bank.set_rank(rank); bank.bind_mask(0, asym)
    ref = [osp.SpectralSplitter(rank, 1) for _ in range(C)]
    col = [[] for _ in range(C)]
    for c in range(C):
        def func(spec, r):
            spec[0::2] *= asym; spec[1::2] *= asym
            return spec
        ref[c].bind(0, func, lambda s, first, count, c=c: col[c].append(s.copy()))
    for k, quiet in ((3 * F, False), (4 * F, True), (2 * F, False)):
        xs = None if quiet else (rng.standard_normal((C, k)) * 0.5).astype(np.float32)
        buf = gpu.DeviceBuffer.from_host(np.full((C, k), -7.0, np.float32))
        bank.process([buf], gpu.DeviceBuffer.from_host(xs) if xs is not None else None, k)
        y = buf.download()
        for c in range(C):
            col[c].clear()
            ref[c].process(xs[c] if xs is not None else None, k)
            w = np.concatenate(col[c])
            assert float(np.abs(y[c] - w).max()) <= TOL * max(1.0, float(np.abs(w).max())), (k, quiet, c)
    bank.close()


def test_hops_sharing_one_launch_with_more_channels_than_the_one_hop_grid_takes(gpu, monkeypatch):
    """Above 512 channels the one-hop launch runs one workgroup per channel (all handlers in turn); the several-hops launch
    keeps one per (channel, handler).  Same floats either way, and the oracle's on the first and last channels."""
    C, rank = 600, 9
    F = 1 << (rank - 1)
    rng = np.random.default_rng(4242)
    lo = osp.lopass_fft_set(3000.0, -24.0, 48000.0, rank)
    hi = osp.hipass_fft_set(3000.0, -24.0, 48000.0, rank)
    calls = (3 * F, 5 * F)
    x = (rng.standard_normal((C, sum(calls))) * 0.5).astype(np.float32)
    got, _ = _gpu_run(gpu, rank, 0, 0.0, [lo, hi], x, calls)
    monkeypatch.setenv("MI_DSPU_TEST_PATH", "splitter_hop_launches")
    one, _ = _gpu_run(gpu, rank, 0, 0.0, [lo, hi], x, calls)
    monkeypatch.delenv("MI_DSPU_TEST_PATH")
    assert np.array_equal(got, one)
    pick = [0, 1, C - 1]
    want = _oracle_run(rank, 0, 0.0, [lo, hi], x[pick], calls)
    for i in range(2):
        assert float(np.abs(got[i][pick] - want[i]).max()) <= TOL * max(1.0, float(np.abs(want[i]).max()))


def test_rank_below_max_rank_rebind_unbind_clear_and_silence(gpu):
    C, rank, n = 2, 9, 2048
    rng = np.random.default_rng(77)
    x = rng.standard_normal((C, 3 * n)).astype(np.float32)
    m1 = osp.lopass_fft_set(1000.0, -24.0, 48000.0, rank)
    m2 = osp.hipass_fft_set(1000.0, -24.0, 48000.0, rank)
    bank = gpu.SplitterBank(C, 11, 2)
    bank.set_rank(rank)
    refs = []
    for c in range(C):
        sp = osp.SpectralSplitter(11, 2); sp.set_rank(rank); refs.append(sp)
    col = {(c, i): [] for c in range(C) for i in range(2)}

    def mk(m):
        def func(spec, r):
            spec[0::2] *= m; spec[1::2] *= m
            return spec
        return func

    def bind_ref(i, m):
        for c, sp in enumerate(refs):
            sp.bind(i, mk(m), lambda s, first, count, c=c, i=i: col[(c, i)].append(s.copy()))

    def step(x_blk, k, listen):
        bufs = [gpu.DeviceBuffer.from_host(np.zeros((C, k), np.float32)) if listen[i] else None for i in range(2)]
        bank.process(bufs, gpu.DeviceBuffer.from_host(x_blk) if x_blk is not None else None, k)
        for c, sp in enumerate(refs):
            sp.process(x_blk[c] if x_blk is not None else None, k)
        return [b.download() if b is not None else None for b in bufs]

    bank.bind_mask(0, m1); bind_ref(0, m1)
    g = step(x[:, :n], n, (True, False))
    want = np.stack([np.concatenate(col[(c, 0)]) for c in range(C)])
    assert np.abs(g[0] - want).max() <= TOL * 4.0
    # second handler joins mid-stream (its line starts from zero), first one gets new gains without losing its line
    bank.bind_mask(1, m2); bind_ref(1, m2)
    bank.bind_mask(0, m2)
    for sp in refs:
        sp.h[0]["func"] = mk(m2)                             # FFTCrossover::update_band: same binding, new vFFT
    for k in col:
        col[k].clear()
    g = step(x[:, n:2 * n], n, (True, True))
    for i in range(2):
        want = np.stack([np.concatenate(col[(c, i)]) for c in range(C)])
        assert np.abs(g[i] - want).max() <= TOL * 4.0, i
    # unbind one, feed silence (src == NULL): the other drains its tail
    bank.unbind(0)
    for sp in refs:
        sp.unbind(0)
    with pytest.raises(gpu.MiError):
        bank.unbind(0)
    for k in col:
        col[k].clear()
    g = step(None, 1024, (True, True))
    assert np.all(g[0] == 0.0)                               # an unbound handler's buffer is not written
    want = np.stack([np.concatenate(col[(c, 1)]) for c in range(C)])
    assert np.abs(g[1] - want).max() <= TOL * 4.0 and np.abs(want).max() > 0.1
    bank.clear()
    for sp in refs:
        sp.clear()
    for k in col:
        col[k].clear()
    g = step(None, 600, (False, True))
    assert np.all(g[1] == 0.0)
    bank.close()


def test_callback_handler_gets_the_full_spectrum(gpu):
    """CALLBACK path: the function sees [channels][2^rank] complex bins on the device and leaves its result in `out`;
    here it is a device-to-device copy, so the handler must reproduce the input delayed by the latency."""
    C, rank, chunk, n = 2, 10, 9, 6000
    rng = np.random.default_rng(5)
    x = rng.standard_normal((C, n)).astype(np.float32)
    bank = gpu.SplitterBank(C, rank, 2)
    bank.set_chunk_rank(chunk)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    seen = []

    def func(out, inp, r, ch, st):
        seen.append((r, ch))
        assert hip.hipMemcpyAsync(out, inp, ch * (2 << r) * 4, 3, st) == 0
    bank.bind_callback(0, func)
    bank.bind_mask(1, np.ones(1 << rank, np.float32))
    o0 = gpu.DeviceBuffer((C, n)); o1 = gpu.DeviceBuffer((C, n))
    bank.process([o0, o1], gpu.DeviceBuffer.from_host(x), n)
    y0, y1 = o0.download(), o1.download()
    lat = bank.latency()
    assert seen and seen[0] == (rank, C) and len(seen) == (n - 1) // (lat // 2)
    assert np.abs(y0[:, 2 * lat:] - x[:, lat:n - lat]).max() < 2e-5 * np.abs(x).max()
    assert np.abs(y0 - y1).max() < 2e-5 * np.abs(x).max()     # same result through the fused mask path
    bank.close()


def test_largest_frames_callback_copy_and_rank_switch(gpu):
    """Ranks 15 .. 18 (frames that do not fit the LDS): a CALLBACK handler sees the complex spectrum and hands it back, so it
    reproduces the input delayed by the latency; switching the same bank down to rank 12 and back up works like a fresh one."""
    C, n = 2, 50000
    rng = np.random.default_rng(15)
    x = rng.standard_normal((C, n)).astype(np.float32)
    bank = gpu.SplitterBank(C, 17, 2)
    bank.set_rank(15); bank.set_chunk_rank(13)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    seen = []

    def func(out, inp, r, ch, st):
        seen.append((r, ch))
        assert hip.hipMemcpyAsync(out, inp, ch * (2 << r) * 4, 3, st) == 0
    bank.bind_callback(0, func)
    bank.bind_mask(1, np.ones(1 << 15, np.float32))
    o0 = gpu.DeviceBuffer((C, n)); o1 = gpu.DeviceBuffer((C, n))
    bank.process([o0, o1], gpu.DeviceBuffer.from_host(x), n)
    y0, y1 = o0.download(), o1.download()
    lat = bank.latency()
    assert lat == 1 << 13 and seen and seen[0] == (15, C) and len(seen) == (n - 1) // (lat // 2)
    assert np.abs(y0[:, 2 * lat:] - x[:, lat:n - lat]).max() < 2e-5 * np.abs(x).max()
    assert np.abs(y0 - y1).max() < 2e-5 * np.abs(x).max()
    # down to a frame that fits the LDS, then up again: each switch restarts the unit (update_settings clears it)
    for rank in (12, 16):
        bank.set_rank(rank); bank.set_chunk_rank(11)
        bank.bind_mask(1, np.ones(1 << rank, np.float32))
        o0 = gpu.DeviceBuffer((C, n)); o1 = gpu.DeviceBuffer((C, n))
        bank.process([o0, o1], gpu.DeviceBuffer.from_host(x), n)
        y0, y1 = o0.download(), o1.download()
        lat = bank.latency()
        assert lat == 1 << 11
        assert np.abs(y1[:, lat:] - x[:, :n - lat]).max() < 2e-5 * np.abs(x).max(), rank
        assert np.abs(y0 - y1).max() < 2e-5 * np.abs(x).max(), rank
    bank.close()


def test_argument_errors(gpu):
    with pytest.raises(gpu.MiError):
        gpu.SplitterBank(1, 4, 1)
    with pytest.raises(gpu.MiError):
        gpu.SplitterBank(1, 19, 1)
    bank = gpu.SplitterBank(1, 8, 2)
    with pytest.raises(gpu.MiError):
        bank.bind_copy(2)
    bank.set_rank(9)                                          # above max_rank: ignored
    assert bank.rank() == 8 and bank.latency() == 256 and bank.remaining() == 128
    bank.set_chunk_rank(3)
    assert bank.chunk_rank() == 5 and bank.latency() == 32
    bank.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_operation_sequences_match_oracle(gpu, seed):
    """Differential stress against one oracle SpectralSplitter per channel: rank / chunk rank / phase changes, handlers
    bound (gains or plain copy), re-bound and unbound, clear(), silence and ragged process() calls in random order."""
    rng = np.random.default_rng(5000 + seed)
    C, max_rank, H = 2, 8, 3
    bank = gpu.SplitterBank(C, max_rank, H)
    refs = [osp.SpectralSplitter(max_rank, H) for _ in range(C)]
    col = {}

    def sink_for(c, h):
        def sink(s, first, count):
            col[(c, h)].append(s.copy())
        return sink

    def mk(m):
        def func(spec, r):
            spec[0::2] *= m; spec[1::2] *= m
            return spec
        return func
    bound = [None] * H                                       # None | "copy" | "mask"
    log = []
    for step in range(45):
        op = rng.choice(["process", "process", "process", "silence", "rank", "chunk", "phase", "mask", "copy", "unbind", "clear"])
        rank = refs[0].rank
        if op in ("process", "silence"):
            k = int(rng.choice([1, 15, 16, 17, 100, 256, int(rng.integers(1, 700))]))
            x = rng.standard_normal((C, k)).astype(np.float32) if op == "process" else None
            bufs = [gpu.DeviceBuffer.from_host(np.full((C, k), -3.0, np.float32)) if bound[h] is not None else None for h in range(H)]
            bank.process(bufs, gpu.DeviceBuffer.from_host(x) if x is not None else None, k)
            for key in col:
                col[key].clear()
            for c in range(C):
                refs[c].process(x[c] if x is not None else None, k)
            for h in range(H):
                if bound[h] is None:
                    continue
                y = bufs[h].download()
                for c in range(C):
                    ref = np.concatenate(col[(c, h)]) if col[(c, h)] else np.zeros(0, np.float32)
                    assert ref.size == k, (seed, step, h, c, ref.size, k, log)
                    err = float(np.abs(y[c] - ref).max())
                    assert err <= TOL * max(float(np.abs(ref).max()), 1.0), (seed, step, h, c, err, log)
            assert bank.latency() == refs[0].latency()
        elif op == "rank":
            r = int(rng.integers(5, max_rank + 2))
            bank.set_rank(r)
            for s in refs:
                s.set_rank(r)
            if refs[0].rank != rank:                            # gains belong to one rank: start over with the handlers
                for h in range(H):
                    if bound[h] is not None:
                        bank.unbind(h)
                        for s in refs:
                            s.unbind(h)
                        bound[h] = None
        elif op == "chunk":
            cr = int(rng.choice([0, 3, 5, 6, 7, 8, 12]))
            bank.set_chunk_rank(cr)
            for s in refs:
                s.set_chunk_rank(cr)
        elif op == "phase":
            ph = float(rng.uniform(-0.2, 1.2))
            bank.set_phase(ph)
            for s in refs:
                s.set_phase(ph)
        elif op in ("mask", "copy"):
            h = int(rng.integers(0, H))
            if op == "mask":
                masks = rng.uniform(0.0, 1.5, (C, 1 << rank)).astype(np.float32)
                if bound[h] == "mask":                          # new gains for a bound handler: the line lives on
                    bank.bind_mask(h, masks)
                    for c in range(C):
                        refs[c].h[h]["func"] = mk(masks[c])
                else:
                    bank.bind_mask(h, masks)
                    for c in range(C):
                        col.setdefault((c, h), [])
                        refs[c].bind(h, mk(masks[c]), sink_for(c, h))
                bound[h] = "mask"
            else:
                bank.bind_copy(h)
                for c in range(C):
                    col.setdefault((c, h), [])
                    refs[c].bind(h, None, sink_for(c, h))
                bound[h] = "copy"
        elif op == "unbind":
            h = int(rng.integers(0, H))
            if bound[h] is not None:
                bank.unbind(h)
                for s in refs:
                    s.unbind(h)
                bound[h] = None
        else:
            bank.clear()
            for s in refs:
                s.clear()
        log.append(str(op))
    bank.close()


def test_an_output_that_overlaps_the_input_is_refused(gpu):
    """The bands are buffers here (sink functions in the reference, SpectralSplitter.cpp:344-356): an output row that overlaps the
    input rows is refused (MI_EINVAL) instead of being filled with something else; the bank goes on with the next proper call."""
    C, rank, n = 2, 9, 1024
    sp = gpu.SplitterBank(C, rank, 2)
    sp.bind_copy(0)
    sp.bind_copy(1)
    x = (np.random.default_rng(4).standard_normal((C, n)) * 0.25).astype(np.float32)
    din = gpu.DeviceBuffer.from_host(x)
    other = gpu.DeviceBuffer((C, n))
    with pytest.raises(gpu.MiError):
        sp.process([din, other], din, n)
    with pytest.raises(gpu.MiError):
        sp.process([other, din.ptr + 64], din, n - 16, out_stride=n, in_stride=n)
    a, b2 = gpu.DeviceBuffer((C, n)), gpu.DeviceBuffer((C, n))
    sp.process([a, b2], din, n)
    assert np.isfinite(a.download()).all()
    sp.close()


@pytest.mark.parametrize("rank,bands,n_frames,K,listen", [(12, 4, 2, 5, None), (9, 3, 1, 7, None), (10, 2, 3, 4, None), (11, 4, 2, 70, None),
                                                           (10, 4, 2, 6, [0, 2]), (12, 4, 2, 37, None), (12, 4, 2, 70, [1, 3]), (12, 3, 2, 9, [2]),
                                                           (12, 6, 2, 5, None)])
def test_process_blocks_equal_block_by_block(gpu, rank, bands, n_frames, K, listen):
    """mi_splitter_bank_process_blocks: K blocks of whole frames as ONE launch of the several-hops kernel (70 blocks: two) against
    K process() calls on a twin bank -- every band bit for bit, and the state left behind (an odd-sized call and a block through
    both); handlers nobody listens to are skipped in both.
    Rank 12 with blocks of exactly one frame rides splitter_wave_blocks_kernel (a wave per channel and segment of the run, two
    frames per complex transform on the wave-resident core, one forward transform for all bands): the same sums through another
    transform -- within 1e-6 of the peak of the calls, the state it leaves included (MI_DSPU_COMPAT_BITS=1 keeps the workgroup kernel:
    test_process_blocks_rank_12_on_the_workgroup_kernel_are_the_calls_bits)."""
    rng = np.random.default_rng(900 + rank + K)
    C, frame = 3, 1 << (rank - 1)
    n = n_frames * frame
    listen = list(range(bands)) if listen is None else listen
    x = (rng.standard_normal((K + 2, C, n)) * 0.25).astype(np.float32)
    masks = [np.clip(rng.uniform(0.0, 1.2, 1 << rank), 0.0, 1.0).astype(np.float32) for _ in range(bands)]

    def make():
        sp = gpu.SplitterBank(C, rank, bands)
        for i in range(bands):
            sp.bind_mask(i, masks[i])
        return sp

    def outs_of():
        return [gpu.DeviceBuffer.from_host(np.full((C, n), 7.0, np.float32)) if i in listen else None for i in range(bands)]
    a, b = make(), make()
    ins = [gpu.DeviceBuffer.from_host(x[k]) for k in range(K + 2)]
    oa = [outs_of() for _ in range(K + 2)]
    ob = [outs_of() for _ in range(K + 2)]
    a.process(oa[0], ins[0], n)                              # the steady state: a frame is in hand
    a.process_blocks(oa[1:K + 1], ins[1:K + 1], n)
    a.process(oa[K + 1], ins[K + 1], n - 37, n, n)
    for k in range(K + 1):
        b.process(ob[k], ins[k], n)
    b.process(ob[K + 1], ins[K + 1], n - 37, n, n)
    waves = rank == 12 and n_frames == 2 and len(listen) <= 4 and os.environ.get("MI_DSPU_COMPAT_BITS") is None
    differs = False
    for k in range(K + 2):
        m = n if k <= K else n - 37
        for i in listen:
            ya, yb = oa[k][i].download()[:, :m], ob[k][i].download()[:, :m]
            assert k == 0 or np.abs(yb).max() > 1e-4
            if waves:
                assert np.abs(ya - yb).max() <= 1e-6 * max(float(np.abs(yb).max()), 0.25), (k, i, float(np.abs(ya - yb).max()))
                differs = differs or not np.array_equal(ya, yb)
            else:
                np.testing.assert_array_equal(ya, yb, err_msg="block %d band %d" % (k, i))
    assert differs == waves                                     # (it IS the other kernel that ran)
    a.close(); b.close()


def test_long_call_at_rank_12_rides_the_wave_kernel(gpu):
    """A process() call of eight or more whole blocks at rank 12 (listening masks shared by the channels, the bands in buffers of
    their own) goes out on splitter_wave_blocks_kernel -- the blocks as column slices of the caller's buffers, a channel's run in
    segments -- against the oracle, every band; the state it leaves serves an odd-sized call and a shorter one (workgroup kernels)."""
    rng = np.random.default_rng(4242)
    C, rank, bands = 3, 12, 3
    N = 1 << rank
    sizes = [N, 17 * N, 300, 5 * N, 9 * N]
    x = (rng.standard_normal((C, sum(sizes))) * 0.25).astype(np.float32)
    masks = [np.clip(rng.uniform(0.0, 1.2, N), 0.0, 1.0).astype(np.float32) for _ in range(bands)]
    want = _oracle_run(rank, rank, 0.0, masks, x, sizes)
    bank = gpu.SplitterBank(C, rank, bands)
    bank.set_rank(rank); bank.set_chunk_rank(rank); bank.set_phase(0.0)
    for i in range(bands):
        bank.bind_mask(i, masks[i])
    got, pos = [[] for _ in range(bands)], 0
    for n in sizes:
        d = gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, pos:pos + n]))
        outs = [gpu.DeviceBuffer((C, n)) for _ in range(bands)]
        bank.process(outs, d, n)
        # (which launch the call took: the column slices of a call's buffers interleave row by row -- ADVICE r05 found them
        # refused as overlapping and this test green on the workgroup kernels)
        # (the 17-block call finds the bank on a frame boundary; the 9-block one behind the 300-sample call does not and stays
        # on the workgroup kernels, as does every call when MI_DSPU_COMPAT_BITS is set)
        if n == 17 * N and os.environ.get("MI_DSPU_COMPAT_BITS") is None:
            assert gpu.last_launch().startswith("(splitter_wave_blocks_kernel"), (n, gpu.last_launch())
        for i in range(bands):
            got[i].append(outs[i].download())
        pos += n
    bank.close()
    peak = float(np.abs(x).max())
    for i in range(bands):
        y = np.concatenate(got[i], axis=1)
        err = float(np.abs(y - want[i]).max())
        assert float(np.abs(want[i]).max()) > 0.01
        assert err <= TOL * max(peak, float(np.abs(want[i]).max())), (i, err)


def test_process_blocks_rank_12_on_the_workgroup_kernel_are_the_calls_bits():
    """MI_DSPU_COMPAT_BITS=1: runs of 4096-sample blocks at rank 12 on splitter_hops_blocks_kernel<11> -- the bits of block-by-block calls."""
    import subprocess
    import sys
    env = dict(os.environ, MI_DSPU_COMPAT_BITS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.abspath(__file__) + "::test_process_blocks_equal_block_by_block"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rank,bands,n_frames,K", [(12, 4, 2, 6), (10, 3, 1, 9), (12, 4, 2, 33)])
def test_runs_of_blocks_match_the_oracle(gpu, rank, bands, n_frames, K):
    """mi_splitter_bank_process_blocks -- K blocks of whole frames as ONE launch of splitter_hops_blocks_kernel (what bench.py's
    splitter row times: rank 12, four masks, 4096-sample blocks) -- directly against the oracle's SpectralSplitter fed the same
    blocks, every band, not only bit for bit against the per-block launches."""
    rng = np.random.default_rng(1200 + rank + K)
    C, frame = 3, 1 << (rank - 1)
    n = n_frames * frame
    x = (rng.standard_normal((C, (K + 1) * n)) * 0.25).astype(np.float32)
    masks = [np.clip(rng.uniform(0.0, 1.2, 1 << rank), 0.0, 1.0).astype(np.float32) for _ in range(bands)]
    want = _oracle_run(rank, rank, 0.0, masks, x, [n] * (K + 1))
    bank = gpu.SplitterBank(C, rank, bands)
    bank.set_rank(rank); bank.set_chunk_rank(rank); bank.set_phase(0.0)
    for i in range(bands):
        bank.bind_mask(i, masks[i])
    ins = [gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, k * n:(k + 1) * n])) for k in range(K + 1)]
    outs = [[gpu.DeviceBuffer((C, n)) for _ in range(bands)] for _ in range(K + 1)]
    bank.process(outs[0], ins[0], n)                         # the steady state: a frame is in hand
    bank.process_blocks(outs[1:], ins[1:], n)
    peak = float(np.abs(x).max())
    for i in range(bands):
        got = np.concatenate([outs[k][i].download() for k in range(K + 1)], axis=1)
        err = float(np.abs(got - want[i]).max())
        assert float(np.abs(want[i]).max()) > 0.01
        assert err <= TOL * max(peak, float(np.abs(want[i]).max())), (i, err)
    bank.close()
