"""The oracle's FFT primitives and Convolver, pinned with the reference's own unit-test vectors."""
import numpy as np

import oracle


def test_fft_matches_numpy_definition():
    rng = np.random.default_rng(0)
    for rank in (1, 2, 5, 8, 12):
        n = 1 << rank
        z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        x = np.empty(2 * n, np.float32); x[0::2] = z.real; x[1::2] = z.imag
        y = oracle.packed_direct_fft(x, rank)
        ref = np.fft.fft(z.astype(np.complex128))           # unnormalised, e^{-jwn}
        got = y[0::2] + 1j * y[1::2]
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() * max(1, rank)
        back = oracle.packed_reverse_fft(y, rank)            # 1/N scaled inverse
        assert np.abs(back - x).max() <= 2e-6 * max(1, rank)


def test_reference_utest_small():
    """src/test/utest/util/convolver.cpp:88-136: rank 9, conv[i]=i+1 (31 taps), sparse source, 31-sample chunks."""
    conv = np.arange(1, 0x20, dtype=np.float32)
    src = np.zeros(0x2000 + conv.size, np.float32)
    for j, i in enumerate(range(0, 0x2000, 5)):
        src[i] = (1.0, 0.1, 0.01)[j % 3]
    d1 = oracle.convolve_f64(src, conv, 0x2000)[:src.size]
    d2 = oracle.convolve(src, conv, 0x2000)[:src.size]
    d3 = oracle.Convolver(conv, 9).process_chunked(src, 31)

    def equals_relative(a, b, tol):
        return bool(np.all(np.abs(a - b) <= tol * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)) )
    # FloatBuffer::equals_relative with 1e-4 (convolver.cpp:123); values below float32 noise compare absolute
    assert np.abs(d2 - d1).max() <= 1e-4 * np.abs(d1).max()
    mask = np.abs(d2) > 1e-3
    assert np.all(np.abs(d3[mask] - d2[mask]) <= 1e-4 * np.abs(d2[mask]))
    assert np.abs(d3 - d2).max() <= 1e-4


def test_reference_utest_large():
    """convolver.cpp:184-223: rank 10, 0x2000-tap random IR, 0x20 random samples then zeros, 31-sample chunks."""
    rng = np.random.default_rng(1234)
    conv = rng.uniform(0.0, 1.0, 0x2000).astype(np.float32)     # FloatBuffer default fill is random [0,1)
    src = np.zeros(0x20 + conv.size, np.float32)
    src[:0x20] = rng.uniform(0.0, 1.0, 0x20).astype(np.float32)
    d1 = oracle.convolve_f64(src, conv, 0x20)[:src.size]
    d2 = oracle.convolve(src, conv, 0x20)[:src.size]
    d3 = oracle.Convolver(conv, 10).process_chunked(src, 31)
    assert np.abs(d2 - d1).max() <= 1e-4                         # equals_absolute 1e-4 (convolver.cpp:210)
    assert np.abs(d3 - d2).max() <= 1e-4


def test_collisions_subset():
    """convolver.cpp:138-182 (disabled upstream): two unit impulses, rank 10, 127-sample chunks + flush."""
    rng = np.random.default_rng(7)
    conv = rng.uniform(-1.0, 1.0, 4096).astype(np.float32)
    for gap in (1, 2, 127, 128, 129, 1000, 4095):
        src = np.zeros(4096 + conv.size, np.float32)
        src[0] = 1.0; src[gap] = 1.0
        ref = oracle.convolve_f64(src, conv, 4096)[:src.size]
        got = oracle.Convolver(conv, 10).process_chunked(src, 127)
        assert np.abs(got - ref).max() <= 1e-5, gap


def test_frame_sized_calls_c3_shape():
    """C3 shape scaled down: rank 13 frame calls, decaying-noise IR of 5 partitions + odd remainder."""
    rng = np.random.default_rng(4)
    taps = 4096 * 5 + 777
    ir = (rng.standard_normal(taps) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    x = rng.standard_normal(4096 * 3).astype(np.float32)
    c = oracle.Convolver(ir, 13)
    got = np.concatenate([c.process(x[i:i + 4096]) for i in range(0, x.size, 4096)])
    ref = oracle.convolve_f64(x, ir)[:x.size]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert (c.data_size, c.rank) == (taps, 13)


def test_uninitialised_outputs_zero():
    c = oracle.Convolver(np.zeros(0, np.float32), 9)
    np.testing.assert_array_equal(c.process(np.ones(100, np.float32)), np.zeros(100, np.float32))
