"""Oracle Delay / RingBuffer pinned by the reference's exact RingBuffer utest and the delay-line identity."""
import numpy as np

from oracle import delay as od
from ringbuffer_kat import KAT


def test_reference_ringbuffer_kat():
    rb = None
    for op, arg, exp in KAT:
        if op == "init":
            rb = od.RingBuffer(arg)
            assert rb.cap == 8
        elif op == "append1":
            for v in arg:
                rb.append_one(v)
        elif op == "append":
            assert rb.append(arg) == exp
        elif op == "get1":
            assert [float(rb.get_one(o)) for o in arg] == [float(v) for v in exp]
        elif op == "get":
            dst, n = rb.get(*arg)
            assert n == exp[0] and list(dst) == [float(v) for v in exp[1]], (arg, dst, n)


def test_delay_identity_and_wrap():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(5000).astype(np.float32)
    for d in (0, 1, 100, 511, 512, 1000):
        dl = od.Delay(1000)
        assert dl.size == 1536
        dl.set_delay(d)
        y = np.concatenate([dl.process(x[i:i + 700]) for i in range(0, 5000, 700)])
        ref = np.concatenate([np.zeros(d, np.float32), x[:5000 - d]])
        np.testing.assert_array_equal(y, ref)


def test_ramping_endpoints():
    x = np.arange(1, 2001, dtype=np.float32)
    dl = od.Delay(600)
    dl.set_delay(100)
    dl.process(x[:1000])
    y = dl.process_ramping(x[1000:], 300)
    assert dl.delay == 300
    assert y[0] == x[1000 - 100]                     # starts at the old delay
    z = dl.process(np.arange(2001, 2011, dtype=np.float32))
    assert z[0] == x[2000 - 300]                     # continues at the new one
