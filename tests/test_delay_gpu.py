"""GPU Delay / RingBuffer banks: bit-exact against the oracle (and, for RingBuffer, the reference's own KAT)."""
import numpy as np
import pytest

from oracle import delay as od
from ringbuffer_kat import KAT

pytestmark = pytest.mark.gpu


def test_reference_ringbuffer_kat_on_gpu(gpu):
    """src/test/utest/util/ringbuffer.cpp:30-192, every channel of a 3-channel bank."""
    C = 3
    rb = None
    for op, arg, exp in KAT:
        if op == "init":
            rb = gpu.RingBank(C, arg)
            assert rb.info()["capacity"] == 8
        elif op in ("append1", "append"):
            items = [[v] for v in arg] if op == "append1" else [arg]
            for it in items:
                d = gpu.DeviceBuffer.from_host(np.tile(np.array(it, np.float32), (C, 1)))
                n = rb.append(d, len(it))
                if op == "append":
                    assert n == exp
        elif op == "get1":
            for o, e in zip(arg, exp):
                out = gpu.DeviceBuffer((C, 1))
                rb.get(out, o, 1)
                assert np.all(out.download() == np.float32(e)), (o, e)
        elif op == "get":
            out = gpu.DeviceBuffer.from_host(np.full((C, arg[1]), 99.0, np.float32))
            n = rb.get(out, arg[0], arg[1])
            assert n == exp[0]
            np.testing.assert_array_equal(out.download(), np.tile(np.array(exp[1], np.float32), (C, 1)))
    rb.close()


@pytest.mark.parametrize("in_place", [False, True])
def test_delay_process_variants_bit_exact(gpu, in_place):
    rng = np.random.default_rng(1)
    C, maxd = 4, 1000
    delays = [0, 1, 511, 1000]
    x = rng.standard_normal((C, 6000)).astype(np.float32)
    g = rng.uniform(0.5, 2.0, (C, 6000)).astype(np.float32)
    base = rng.standard_normal((C, 6000)).astype(np.float32)
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    for c, d in enumerate(delays):
        bank.set_delay(d, c); refs[c].set_delay(d)
    assert bank.get(3) == {"delay": 1000, "size": 1536, "head": 0, "tail": 536}
    pos = 0
    for n, mode in ((700, "plain"), (1536, "scalar"), (64, "vector"), (2000, "add"), (1700, "add_vector")):
        xs, gs, bs = x[:, pos:pos + n], g[:, pos:pos + n], base[:, pos:pos + n]
        din = gpu.DeviceBuffer.from_host(xs)
        dout = din if (in_place and not mode.startswith("add")) else gpu.DeviceBuffer.from_host(bs)
        dg = gpu.DeviceBuffer.from_host(gs)
        if mode == "plain":
            bank.process(dout, din, n); ref = [r.process(xs[c]) for c, r in enumerate(refs)]
        elif mode == "scalar":
            bank.process(dout, din, n, gain=0.37); ref = [r.process(xs[c], gain=0.37) for c, r in enumerate(refs)]
        elif mode == "vector":
            bank.process(dout, din, n, gain_vec=dg); ref = [r.process(xs[c], gain=gs[c]) for c, r in enumerate(refs)]
        elif mode == "add":
            bank.process(dout, din, n, add=True); ref = [r.process(xs[c], add_to=bs[c]) for c, r in enumerate(refs)]
        else:
            bank.process(dout, din, n, add=True, gain_vec=dg); ref = [r.process(xs[c], gain=gs[c], add_to=bs[c]) for c, r in enumerate(refs)]
        np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg=mode)
        pos += n
    bank.close()


@pytest.mark.parametrize("in_place", [False, True])
def test_delay_short_blocks_single_pass_bit_exact(gpu, in_place):
    """Blocks no longer than the shortest delay (and than size - longest delay) take the one-kernel exchange path
    (delay_exchange_kernel): same samples as the reference's push/pull order, in place or not, with wrap-around."""
    rng = np.random.default_rng(3)
    C, maxd = 4, 1000
    delays = [300, 512, 700, 1000]
    x = rng.standard_normal((C, 4000)).astype(np.float32)
    g = rng.uniform(0.5, 2.0, (C, 4000)).astype(np.float32)
    base = rng.standard_normal((C, 4000)).astype(np.float32)
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    for c, d in enumerate(delays):
        bank.set_delay(d, c); refs[c].set_delay(d)
    pos = 0
    for n, mode in ((300, "plain"), (256, "scalar"), (300, "vector"), (299, "add"), (300, "plain"), (300, "plain"),
                    (300, "scalar"), (1, "plain"), (300, "add"), (300, "plain")):
        xs, gs, bs = x[:, pos:pos + n], g[:, pos:pos + n], base[:, pos:pos + n]
        din = gpu.DeviceBuffer.from_host(xs)
        dout = din if (in_place and mode != "add") else gpu.DeviceBuffer.from_host(bs)
        dg = gpu.DeviceBuffer.from_host(gs)
        if mode == "plain":
            bank.process(dout, din, n); ref = [r.process(xs[c]) for c, r in enumerate(refs)]
        elif mode == "scalar":
            bank.process(dout, din, n, gain=0.37); ref = [r.process(xs[c], gain=0.37) for c, r in enumerate(refs)]
        elif mode == "vector":
            bank.process(dout, din, n, gain_vec=dg); ref = [r.process(xs[c], gain=gs[c]) for c, r in enumerate(refs)]
        else:
            bank.process(dout, din, n, add=True); ref = [r.process(xs[c], add_to=bs[c]) for c, r in enumerate(refs)]
        np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg="%s at %d" % (mode, pos))
        pos += n
    bank.close()


@pytest.mark.parametrize("in_place", [False, True])
def test_delay_quad_paths_bit_exact(gpu, in_place):
    """Blocks, write position and delays that are all multiples of four take the 16-byte kernels (delay_exchange4_kernel for
    blocks no longer than the shortest delay, delay_direct4_kernel beyond): every process form against the oracle, across
    the wrap of the line, then one odd call that knocks the line off the quad grid and back onto the scalar kernels."""
    rng = np.random.default_rng(44)
    C, maxd = 5, 3000
    delays = [0, 4, 512, 1500, 3000] if not in_place else [1024, 2000, 512, 1500, 3000]
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    for c, d in enumerate(delays):
        bank.set_delay(d, c); refs[c].set_delay(d)
    for n, mode in ((256, "plain"), (512, "scalar"), (1024, "vector"), (2048, "add"), (512, "add_vector"), (3584, "plain"),
                    (4096, "vector"), (7, "plain"), (512, "plain"), (1021, "scalar"), (256, "add")):
        x = rng.standard_normal((C, n)).astype(np.float32)
        g = rng.uniform(0.5, 2.0, (C, n)).astype(np.float32)
        base = rng.standard_normal((C, n)).astype(np.float32)
        ip = in_place and not mode.startswith("add")
        din = gpu.DeviceBuffer.from_host(x)
        dout = din if ip else gpu.DeviceBuffer.from_host(base)
        dg = gpu.DeviceBuffer.from_host(g)
        kw = {"plain": {}, "scalar": {"gain": 0.37}, "vector": {"gain_vec": dg}, "add": {"add": True},
              "add_vector": {"add": True, "gain_vec": dg}}[mode]
        bank.process(dout, din, n, **kw)
        ref = []
        for c, r in enumerate(refs):
            rk = {}
            if mode == "scalar": rk["gain"] = 0.37
            if mode in ("vector", "add_vector"): rk["gain"] = g[c]
            if mode.startswith("add"): rk["add_to"] = base[c]
            ref.append(r.process(x[c], **rk))
        np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg=str((mode, n)))
        for c, r in enumerate(refs):
            st = bank.get(c)
            assert (st["head"], st["tail"]) == (r.head, r.tail), (mode, n, c)
    bank.close()


@pytest.mark.parametrize("in_place", [False, True])
def test_delay_ramping_bit_exact(gpu, in_place):
    """Delay::process_ramping index (old_tail + ssize_t(delta * offset)) % size, incl. pieces and wrap."""
    rng = np.random.default_rng(2)
    C, maxd = 5, 2000
    x = rng.standard_normal((C, 9000)).astype(np.float32)
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    start = [0, 100, 1500, 2000, 700]
    for c, d in enumerate(start):
        bank.set_delay(d, c); refs[c].set_delay(d)
    pos = 0
    for n, targets in ((1000, [50, 100, 20, 1999, 701]), (3000, [2000, 0, 1999, 3, 700]), (37, [0, 1, 2, 3, 4]),
                       (4000, [1000, 1000, 1000, 1000, 1000])):
        xs = x[:, pos:pos + n]
        din = gpu.DeviceBuffer.from_host(xs)
        dout = din if in_place else gpu.DeviceBuffer((C, n))
        bank.process_ramping(dout, din, targets, n, gain=1.25)
        ref = np.stack([r.process_ramping(xs[c], targets[c], gain=1.25) for c, r in enumerate(refs)])
        np.testing.assert_array_equal(dout.download(), ref, err_msg=str((n, targets)))
        pos += n
    # the line keeps working with plain process afterwards
    xs = x[:, pos:pos + 500]
    din = gpu.DeviceBuffer.from_host(xs); dout = gpu.DeviceBuffer((C, 500))
    bank.process(dout, din, 500)
    np.testing.assert_array_equal(dout.download(), np.stack([r.process(xs[c]) for c, r in enumerate(refs)]))
    bank.close()


def test_delay_append_and_clear(gpu):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5000)).astype(np.float32)
    bank = gpu.DelayBank(2, 100)          # size 512
    bank.set_delay(100)
    bank.append(gpu.DeviceBuffer.from_host(x[:, :3000]), 3000)      # longer than the line
    din = gpu.DeviceBuffer.from_host(x[:, 3000:3300]); dout = gpu.DeviceBuffer((2, 300))
    bank.process(dout, din, 300)
    np.testing.assert_array_equal(dout.download(), x[:, 2900:3200])
    bank.clear()
    bank.process(dout, din, 50, out_stride=300, in_stride=300)
    np.testing.assert_array_equal(dout.download()[:, :50], np.zeros((2, 50), np.float32))
    bank.close()


@pytest.mark.parametrize("seed", range(10))
def test_delay_random_operation_sequences_bit_exact(gpu, seed):
    """Differential stress, bit for bit: set_delay, clear, append and every process form (plain / gain / add / ramping,
    in place or not) with random lengths around the sizes where the write and read positions wrap."""
    rng = np.random.default_rng(7000 + seed)
    C, maxd = 3, int(rng.choice([100, 511, 513, 1000]))
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    size = refs[0].size
    log = []
    for step in range(60):
        op = rng.choice(["plain", "scalar", "vector", "add", "add_vector", "ramp", "ramp_gain", "set", "clear", "append"])
        n = int(rng.choice([1, 2, 63, size - 1, size, size + 1, 2 * size + 5, int(rng.integers(1, 3 * size))]))
        if op == "set":
            d = int(rng.integers(0, maxd + 1))
            c = int(rng.integers(-1, C))
            if c < 0:
                bank.set_delay(d)
                for r in refs:
                    r.set_delay(d)
            else:
                bank.set_delay(d, c); refs[c].set_delay(d)
        elif op == "clear":
            bank.clear()
            for r in refs:
                r.buf[:] = 0                                  # Delay::clear(): the buffer only (Delay.cpp:574-579)
        elif op == "append":
            x = rng.standard_normal((C, n)).astype(np.float32)
            bank.append(gpu.DeviceBuffer.from_host(x), n)
            for c, r in enumerate(refs):
                r.append(x[c])
        else:
            x = rng.standard_normal((C, n)).astype(np.float32)
            g = rng.uniform(0.5, 2.0, (C, n)).astype(np.float32)
            base = rng.standard_normal((C, n)).astype(np.float32)
            in_place = bool(rng.integers(0, 2))
            delays = [r.delay for r in refs]
            # In place WITHOUT a delay the reference appends the block as a whole (a block of at least the line's length
            # restarts the line at cell 0); the bank keeps one write position for all its channels, so it follows that path
            # when no channel has a delay.  Channels with and without a delay in one in-place call of such a length would
            # need a write position each: that one combination is left to separate banks.
            if in_place and n >= size and min(delays) == 0 and max(delays) > 0:
                in_place = False
            if op.startswith("add") and in_place:
                base = x                                      # dst == src: the accumulator IS the input
            ip = in_place and max(delays) == 0                # the reference's shortcut (Delay.cpp:107-111 and its siblings)
            din = gpu.DeviceBuffer.from_host(x)
            dout = din if in_place else gpu.DeviceBuffer.from_host(base)
            dg = gpu.DeviceBuffer.from_host(g)
            if op == "plain":
                bank.process(dout, din, n); ref = [r.process(x[c], in_place=ip) for c, r in enumerate(refs)]
            elif op == "scalar":
                bank.process(dout, din, n, gain=0.37); ref = [r.process(x[c], gain=0.37, in_place=ip) for c, r in enumerate(refs)]
            elif op == "vector":
                bank.process(dout, din, n, gain_vec=dg); ref = [r.process(x[c], gain=g[c], in_place=ip) for c, r in enumerate(refs)]
            elif op == "add":
                bank.process(dout, din, n, add=True); ref = [r.process(x[c], add_to=base[c], in_place=ip) for c, r in enumerate(refs)]
            elif op == "add_vector":
                bank.process(dout, din, n, add=True, gain_vec=dg); ref = [r.process(x[c], gain=g[c], add_to=base[c], in_place=ip) for c, r in enumerate(refs)]
            else:
                targets = [int(rng.integers(0, maxd + 1)) for _ in range(C)]
                if rng.integers(0, 4) == 0:
                    targets[0] = refs[0].delay                # no change on one channel: plain process (Delay.cpp:404)
                if op == "ramp":
                    bank.process_ramping(dout, din, targets, n)
                    ref = [r.process_ramping(x[c], targets[c]) for c, r in enumerate(refs)]
                else:
                    bank.process_ramping(dout, din, targets, n, gain_vec=dg)
                    ref = [r.process_ramping(x[c], targets[c], gain=g[c]) for c, r in enumerate(refs)]
            np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg=str((seed, step, op, n, log[-6:])))
        for c, r in enumerate(refs):
            st = bank.get(c)
            # the absolute positions matter too: a fast-growing delay makes process_ramping's read index wrap modulo 2^64
            assert (st["delay"], st["head"], st["tail"], st["size"]) == (r.delay, r.head, r.tail, r.size), (seed, step, c, log[-6:])
        log.append((str(op), n))
    bank.close()


@pytest.mark.parametrize("seed", range(10))
def test_delay_lines_with_positions_of_their_own(gpu, seed):
    """Delay objects with DIFFERENT call histories in one bank (util/Delay.h:35: every object has its nHead): calls on random
    subsets of the lines (mi_delay_bank_process_rows / append_rows, rows in random order) mixed with calls on the whole
    bank, every process form, lengths around the wrap points, whole-line appends that restart a line at cell 0 -- outputs
    and the absolute positions of every line bit for bit against one oracle object per line."""
    rng = np.random.default_rng(9100 + seed)
    C, maxd = 6, int(rng.choice([100, 511, 513, 1000]))
    bank = gpu.DelayBank(C, maxd)
    refs = [od.Delay(maxd) for _ in range(C)]
    size = refs[0].size
    log = []
    for c in range(C):
        d = int(rng.integers(0, maxd + 1))
        bank.set_delay(d, c); refs[c].set_delay(d)
    for step in range(70):
        op = str(rng.choice(["plain", "scalar", "vector", "add", "add_vector", "append", "set", "ramp"]))
        whole = bool(rng.integers(0, 3) == 0)
        rows = list(range(C)) if whole else [int(v) for v in rng.permutation(C)[:int(rng.integers(1, C + 1))]]
        n = int(rng.choice([1, 2, 63, size - 1, size, size + 1, 2 * size + 5, int(rng.integers(1, 3 * size))]))
        R = len(rows)
        if op == "set":
            c = int(rng.integers(0, C)); d = int(rng.integers(0, maxd + 1))
            bank.set_delay(d, c); refs[c].set_delay(d)
        elif op == "ramp":                                    # the whole bank or a subset (mi_delay_bank_process_ramping_rows)
            x = rng.standard_normal((R, n)).astype(np.float32)
            targets = [int(rng.integers(0, maxd + 1)) for _ in range(R)]
            din = gpu.DeviceBuffer.from_host(x); dout = gpu.DeviceBuffer((R, n))
            if whole:
                bank.process_ramping(dout, din, targets, n)
            else:
                bank.process_ramping_rows(rows, dout, din, targets, n)
            ref = [refs[c].process_ramping(x[k], targets[k]) for k, c in enumerate(rows)]
            np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg=str((seed, step, op, rows, n, log[-6:])))
        elif op == "append":
            x = rng.standard_normal((R, n)).astype(np.float32)
            if whole:
                bank.append(gpu.DeviceBuffer.from_host(x), n)
            else:
                bank.append_rows(rows, gpu.DeviceBuffer.from_host(x), n)
            for k, c in enumerate(rows):
                refs[c].append(x[k])
        else:
            x = rng.standard_normal((R, n)).astype(np.float32)
            g = rng.uniform(0.5, 2.0, (R, n)).astype(np.float32)
            base = rng.standard_normal((R, n)).astype(np.float32)
            in_place = bool(rng.integers(0, 2))
            delays = [refs[c].delay for c in rows]
            if in_place and n >= size and min(delays) == 0 and max(delays) > 0:
                in_place = False                              # (lines with and without a delay restart differently: two calls)
            if op.startswith("add") and in_place:
                base = x
            ip = in_place and max(delays) == 0
            din = gpu.DeviceBuffer.from_host(x)
            dout = din if in_place else gpu.DeviceBuffer.from_host(base)
            dg = gpu.DeviceBuffer.from_host(g)
            kw = {"plain": {}, "scalar": {"gain": 0.37}, "vector": {"gain_vec": dg}, "add": {"add": True},
                  "add_vector": {"add": True, "gain_vec": dg}}[op]
            if whole:
                bank.process(dout, din, n, **kw)
            else:
                bank.process_rows(rows, dout, din, n, **kw)
            ref = []
            for k, c in enumerate(rows):
                rk = {}
                if op == "scalar": rk["gain"] = 0.37
                if op in ("vector", "add_vector"): rk["gain"] = g[k]
                if op.startswith("add"): rk["add_to"] = base[k]
                ref.append(refs[c].process(x[k], in_place=ip, **rk))
            np.testing.assert_array_equal(dout.download(), np.stack(ref), err_msg=str((seed, step, op, rows, n, log[-6:])))
        for c, r in enumerate(refs):
            st = bank.get(c)
            assert (st["delay"], st["head"], st["tail"], st["size"]) == (r.delay, r.head, r.tail, r.size), (seed, step, c, op, rows, log[-6:])
        log.append((op, "all" if whole else tuple(rows), n))
    bank.close()


def test_delay_rows_argument_errors(gpu):
    bank = gpu.DelayBank(4, 100)
    buf = gpu.DeviceBuffer((4, 16))
    for rows in ([0, 0], [4], [0, 1, 2, 3, 1]):
        with pytest.raises(gpu.MiError):
            bank.process_rows(rows, buf, buf, 16)
        with pytest.raises(gpu.MiError):
            bank.append_rows(rows, buf, 16)
    bank.process_rows([], buf, buf, 16)                       # nothing to do
    bank.close()


def test_delay_rows_calls_are_refused_while_the_stream_captures(gpu):
    """A call on a subset of the lines reads its row list from the caller's memory when it is made: inside a graph capture it
    is refused (MI_ESTATE) before anything is copied or synchronised, the capture stays valid, and whole-bank calls of a bank
    whose lines already stand at positions of their own are still captured (the offsets live on the device)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    st = s.value
    C, n = 4, 64
    rng = np.random.default_rng(5)
    refs = [od.Delay(300) for _ in range(C)]
    K = refs[0].size // n                                    # a graph of the bank covers whole laps of the write position
    x = rng.standard_normal((1 + K, C, n)).astype(np.float32)
    bank = gpu.DelayBank(C, 300)
    for c in range(C):
        bank.set_delay(40 + 8 * c, channel=c)
        refs[c].set_delay(40 + 8 * c)
    # lines 1 and 3 move ahead of the others
    d0 = gpu.DeviceBuffer.from_host(x[0][[1, 3]], stream=st)
    o0 = gpu.DeviceBuffer((2, n))
    bank.process_rows([1, 3], o0, d0, n, stream=st)
    for k, c in enumerate((1, 3)):
        np.testing.assert_array_equal(o0.download(stream=st)[k], refs[c].process(x[0][c]))
    din = [gpu.DeviceBuffer.from_host(x[1 + b], stream=st) for b in range(K)]
    dout = [gpu.DeviceBuffer((C, n)) for _ in range(K)]
    gpu.check(gpu.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(st)))
    with pytest.raises(gpu.MiError):
        bank.process_rows([0, 2], o0, d0, n, stream=st)
    for b in range(K):
        bank.process(dout[b], din[b], n, stream=st)
    h = ctypes.c_void_p()
    gpu.check(gpu.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(st), ctypes.byref(h)))
    gpu.check(gpu.lib.mi_dspu_graph_launch(h, ctypes.c_void_p(st)))
    for b in range(K):
        y = dout[b].download(stream=st)
        for c in range(C):
            np.testing.assert_array_equal(y[c], refs[c].process(x[1 + b][c]))
    gpu.lib.mi_dspu_graph_destroy(h)
    bank.close()
    hip.hipStreamDestroy(s)


@pytest.mark.parametrize("seed", range(8))
def test_ring_random_operation_sequences_bit_exact(gpu, seed):
    """RingBuffer: append (also more than a whole buffer: the reference restarts at cell 0), block get with every kind of
    zero fill, fill(), head and tail positions -- bit for bit against the oracle."""
    rng = np.random.default_rng(21000 + seed)
    C = 2
    size = int(rng.choice([8, 100, 1000]))
    fill0 = float(rng.choice([0.0, 0.5]))
    rb = gpu.RingBank(C, size, fill0)
    refs = [od.RingBuffer(size, fill0) for _ in range(C)]
    cap = refs[0].cap
    assert rb.info()["capacity"] == cap
    for step in range(60):
        op = rng.choice(["append", "append", "get", "get", "fill", "info"])
        if op == "append":
            n = int(rng.choice([1, 2, cap - 1, cap, cap + 1, 3 * cap, int(rng.integers(1, 2 * cap + 2))]))
            x = rng.standard_normal((C, n)).astype(np.float32)
            got = rb.append(gpu.DeviceBuffer.from_host(x), n)
            want = [r.append(x[c]) for c, r in enumerate(refs)]
            assert got == want[0], (seed, step, n)
        elif op == "get":
            off = int(rng.integers(0, 2 * cap + 2)); n = int(rng.integers(1, 2 * cap + 3))
            out = gpu.DeviceBuffer.from_host(np.full((C, n), 77.0, np.float32))
            got = rb.get(out, off, n)
            y = out.download()
            for c, r in enumerate(refs):
                dst, to_read = r.get(off, n)
                assert got == to_read, (seed, step, off, n)
                keep = ~np.isnan(dst)                          # cells the reference leaves untouched stay as they were
                np.testing.assert_array_equal(y[c][keep], dst[keep], err_msg=str((seed, step, off, n)))
                assert np.all(y[c][~keep] == 77.0)
        elif op == "fill":
            v = float(rng.standard_normal())
            rb.fill(v)
            for r in refs:
                r.data[:] = np.float32(v); r.head = 0         # RingBuffer.cpp:115-120
        else:
            off = int(rng.integers(0, cap + 3))
            i = rb.info(off)
            assert (i["head"], i["tail_position"]) == (refs[0].head, refs[0].tail_position(off)), (seed, step, off)
    rb.close()
