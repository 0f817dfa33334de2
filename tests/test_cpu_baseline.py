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
