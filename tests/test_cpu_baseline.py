"""bench.py's CPU baseline (oracle/cpu_baseline: the reference's x8 software pipeline, -O3 -march=native) computes the
same cascade as the parity oracle, to the float32 recursion's own round-off (it may fuse multiply-adds)."""
import ctypes
import os
import subprocess

import numpy as np

import oracle
import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_x8_pipeline_matches_the_oracle():
    base = os.path.join(ROOT, "oracle", "cpu_baseline")
    subprocess.check_call(["make", "-s", "-B", "-C", base])
    lib = ctypes.CDLL(os.path.join(base, "libcpubase.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpu_biquad_x8_run.argtypes = [fp, fp] + [ctypes.c_size_t] * 4 + [fp, fp, ctypes.c_int]
    C, n, nb = 64, 1000, 3                                  # ragged block length, three blocks with carried state
    coef, fc = wl.c2_coefficients(C)
    x = wl.c2_input(C, n, blocks=nb)
    y = np.empty_like(x)
    st = np.zeros((C, 8, 2), np.float32)
    f = lambda a: a.ctypes.data_as(fp)
    for threads in (1, 3):
        st[:] = 0
        assert lib.cpu_biquad_x8_run(f(y), f(x), C, n, nb, nb, f(coef), f(st), threads) == threads
        ref_state = np.zeros((C, 8, 2), np.float32)
        nsec = np.full(C, 8, np.uint32)
        for b in range(nb):
            ref = oracle.biquad_bank(x[b], coef, nsec, ref_state)
            for c in range(C):
                exact = oracle.biquad_cascade_f64(x[:b + 1, c].reshape(-1), coef[c])[b * n:]
                peak = np.abs(exact).max()
                noise = np.abs(ref[c] - exact).max() / peak
                assert np.abs(y[b, c] - exact).max() / peak <= max(1e-5, 4.0 * noise), (threads, b, c, fc[c])
    assert lib.cpu_biquad_x8_run(f(y), f(x), C + 1, n, 1, 1, f(coef), f(st), 1) == -1    # channels not a multiple of 4


def _cpubase():
    base = os.path.join(ROOT, "oracle", "cpu_baseline")
    subprocess.check_call(["make", "-s", "-B", "-C", base])
    return ctypes.CDLL(os.path.join(base, "libcpubase.so"))


def test_vector_fft_primitives_match_the_scalar_oracle():
    """oracle/cpu_baseline/fft_simd_host.c (what bench.py times as the CPU side of C3 / C4 / C5) against oracle/fft_oracle.c, the
    restatement the parity tests use: transforms both ways over the ranks the units use, and the fastconv trio (whose image
    format differs between the two: compared through what they compute)."""
    lib = _cpubase()
    fp = ctypes.POINTER(ctypes.c_float)
    f = lambda a: a.ctypes.data_as(fp)
    for fn in (lib.orc_packed_direct_fft, lib.orc_packed_reverse_fft):
        fn.argtypes = [fp, fp, ctypes.c_size_t]
    lib.orc_fastconv_parse.argtypes = [fp, fp, ctypes.c_size_t]
    lib.orc_fastconv_apply.argtypes = [fp, fp, fp, fp, ctypes.c_size_t]
    lib.orc_fastconv_parse_apply.argtypes = [fp, fp, fp, fp, ctypes.c_size_t]
    rng = np.random.default_rng(5)
    for rank in (2, 3, 4, 5, 8, 9, 12, 13, 14):
        n = 1 << rank
        z = rng.standard_normal(2 * n).astype(np.float32)
        out = np.empty_like(z)
        lib.orc_packed_direct_fft(f(out), f(z), rank)
        zc = z[0::2].astype(np.float64) + 1j * z[1::2].astype(np.float64)
        want = np.fft.fft(zc)
        got = out[0::2] + 1j * out[1::2]
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() * max(1, rank / 4), rank
        back = np.empty_like(z)
        lib.orc_packed_reverse_fft(f(back), f(out), rank)
        assert np.abs(back - z).max() <= 4e-6 * max(1, rank / 4), rank
        # linear convolution through the fastconv trio: n/2 samples with n/2 taps -> n - 1 outputs
        h, x = rng.standard_normal(n // 2).astype(np.float32), rng.standard_normal(n // 2).astype(np.float32)
        img_h, img_x = np.empty(2 * n, np.float32), np.empty(2 * n, np.float32)
        lib.orc_fastconv_parse(f(img_h), f(h), rank)
        lib.orc_fastconv_parse(f(img_x), f(x), rank)
        tmp = np.empty(2 * n, np.float32)
        acc = np.ones(n, np.float32)                          # (the result is ADDED to dst)
        lib.orc_fastconv_apply(f(acc), f(tmp), f(img_h), f(img_x), rank)
        want_c = np.convolve(h.astype(np.float64), x.astype(np.float64))
        want_c = np.concatenate([want_c, [0.0]]) + 1.0
        assert np.abs(acc - want_c).max() <= 1e-5 * np.abs(want_c).max(), rank
        acc2 = np.ones(n, np.float32)
        lib.orc_fastconv_parse_apply(f(acc2), f(tmp), f(img_h), f(x), rank)
        assert np.abs(acc2 - want_c).max() <= 1e-5 * np.abs(want_c).max(), rank


def test_convolver_bank_of_the_cpu_baseline_matches_the_oracle():
    """cpu_convolver_bank_* (the oracle's Convolver on the vectorised primitives, OpenMP over the channels) against
    oracle.Convolver (the same algorithm on the scalar ones) and exact linear convolution."""
    lib = _cpubase()
    fp = ctypes.POINTER(ctypes.c_float)
    f = lambda a: a.ctypes.data_as(fp)
    lib.cpu_convolver_bank_create.restype = ctypes.c_void_p
    lib.cpu_convolver_bank_create.argtypes = [fp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    lib.cpu_convolver_bank_run.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    lib.cpu_convolver_bank_destroy.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(9)
    C, rank, taps, frames = 3, 10, 3 * 512 + 77, 5
    frame = 1 << (rank - 1)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 600.0)).astype(np.float32)
    x = rng.standard_normal((frames, C, frame)).astype(np.float32)
    y = np.empty_like(x)
    bank = lib.cpu_convolver_bank_create(f(irs), taps, C, rank, 2)
    assert lib.cpu_convolver_bank_run(bank, f(y), f(x), frame, frames, frames, 2) == 2
    lib.cpu_convolver_bank_destroy(bank)
    for c in range(C):
        xs = x[:, c, :].reshape(-1)
        ref = oracle.Convolver(irs[c], rank).process_chunked(xs, frame)
        exact = np.convolve(xs.astype(np.float64), irs[c].astype(np.float64))[:xs.size]
        peak = np.abs(exact).max()
        got = y[:, c, :].reshape(-1)
        assert np.abs(got - exact).max() <= 1e-5 * peak, c
        assert np.abs(got - ref).max() <= 1e-5 * peak, c
