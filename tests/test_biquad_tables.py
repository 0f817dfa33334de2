"""Host logic of the biquad kernel, checked without a GPU.

The kernel evaluates each TDF-II section chunk-parallel: zero-state end state of every
chunk by two dot products, an inclusive scan over chunks with powers of P = A^L, then the exact
recurrence from the scanned start state.  This test takes the REAL per-section tables the product
builds (mi_biquad_section_tables, host C++) and replays the kernel's three steps in numpy float32 with
the same lane/chunk index math, against the sequential oracle."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import filter_design as fd
from conftest import assert_iir_parity

import workloads as wl

F = np.float32


def product_tables(mi, q, variant):
    from importlib import import_module
    capi = import_module("lsp-dsp-units_amd.capi")
    chain = capi.BiquadX1(*[float(v) for v in q], 0.0, 0.0, 0.0)
    geo = (ctypes.c_uint32 * 4)()
    mi.check(mi.lib.mi_biquad_section_tables(ctypes.byref(chain), variant, None, geo))
    L, NT, NLEV, TAB = [int(v) for v in geo]
    row = np.zeros(TAB, np.float32)
    mi.check(mi.lib.mi_biquad_section_tables(ctypes.byref(chain), variant,
                                             row.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), geo))
    M = row[8:8 + 4 * NLEV].reshape(NLEV, 2, 2)
    p = row[8 + 4 * NLEV: 8 + 4 * NLEV + L]
    qq = row[8 + 4 * NLEV + L: 8 + 4 * NLEV + 2 * L]
    return L, NT, NLEV, row[:5], M, p, qq


def emulate_block(mi, x, coef, state, variant):
    """One kernel launch for one channel: x has at most L*NT samples."""
    cnt = len(x)
    first = product_tables(mi, coef[0], variant) if len(coef) else None
    L, NT = (first[0], first[1]) if first else ((32, 128) if variant == 0 else (8, 64))
    assert cnt <= L * NT
    X = np.zeros(L * NT, F)
    X[:cnt] = x
    X = X.reshape(NT, L).copy()
    t_last = (cnt - 1) // L
    m_last = cnt - t_last * L
    for s, q in enumerate(coef):
        L, NT, NLEV, c5, M, p, qq = product_tables(mi, q, variant)
        np.testing.assert_array_equal(c5, np.asarray(q, F))
        b0, b1, b2, a1, a2 = [F(v) for v in q]
        c0, c1 = F(state[s][0]), F(state[s][1])
        z0 = np.zeros(NT, F); z1 = np.zeros(NT, F); w0 = np.zeros(NT, F); w1 = np.zeros(NT, F)
        for k in range(0, L, 2):
            z0 = (p[k] * X[:, k] + z0).astype(F); w0 = (qq[k] * X[:, k] + w0).astype(F)
            z1 = (p[k + 1] * X[:, k + 1] + z1).astype(F); w1 = (qq[k + 1] * X[:, k + 1] + w1).astype(F)
        z = (z0 + z1).astype(F); w = (w0 + w1).astype(F)
        z[0] = F(M[0, 0, 0] * c0 + F(M[0, 0, 1] * c1 + z[0]))
        w[0] = F(M[0, 1, 0] * c0 + F(M[0, 1, 1] * c1 + w[0]))
        for j in range(NLEV):
            d = 1 << j
            zs = np.concatenate([np.zeros(d, F), z[:-d]]); ws = np.concatenate([np.zeros(d, F), w[:-d]])
            act = np.arange(NT) >= d
            zn = (M[j, 0, 0] * zs + (M[j, 0, 1] * ws + z).astype(F)).astype(F)
            wn = (M[j, 1, 0] * zs + (M[j, 1, 1] * ws + w).astype(F)).astype(F)
            z = np.where(act, zn, z); w = np.where(act, wn, w)
        d0 = np.concatenate([[c0], z[:-1]]).astype(F)
        d1 = np.concatenate([[c1], w[:-1]]).astype(F)
        f0 = d0.copy(); f1 = d1.copy()
        for k in range(L):
            xx = X[:, k]
            y = (b0 * xx + d0).astype(F)
            tt = (b1 * xx + d1).astype(F)
            d0 = (a1 * y + tt).astype(F)
            d1 = (a2 * y + (b2 * xx).astype(F)).astype(F)
            X[:, k] = y
            if k + 1 == m_last:
                f0 = d0.copy(); f1 = d1.copy()
        state[s][0] = f0[t_last]; state[s][1] = f1[t_last]
    return X.reshape(-1)[:cnt]


def emulate(mi, x, coef):
    st = [[F(0), F(0)] for _ in coef]
    out = np.empty_like(x)
    done = 0
    while done < len(x):
        left = len(x) - done
        if left > 512:
            step, variant = min(left, 4096), 0
        else:
            step, variant = left, 1
        out[done:done + step] = emulate_block(mi, x[done:done + step], coef, st, variant)
        done += step
    return out, np.array(st, F)


CASES = [
    ("lrx_lp_1k", fd.FLT_BT_LRX_LOPASS, 4, 1000.0, 1.0, 0.75),
    ("lrx_lp_200", fd.FLT_BT_LRX_LOPASS, 4, 200.0, 1.0, 0.75),
    ("lrx_lp_18k", fd.FLT_BT_LRX_LOPASS, 4, 18000.0, 1.0, 0.75),
    ("bell_20", fd.FLT_BT_RLC_BELL, 4, 20.0, 4.0, 2.0),
    ("hishelf", fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 2.0, 0.0),
    ("kweight", fd.FLT_K_WEIGHTED, 1, 0.0, 1.0, 0.0),
    ("notch", fd.FLT_BT_RLC_NOTCH, 1, 5000.0, 1.0, 3.0),
]


@pytest.mark.parametrize("name,ftype,slope,freq,gain,q", CASES)
@pytest.mark.parametrize("n", [4096, 5000, 300])
def test_chunked_form_matches_sequential(mi, name, ftype, slope, freq, gain, q, n):
    coef = wl.design(ftype, slope, freq, freq, gain, q)
    x = (np.random.default_rng(42).standard_normal(n) * 0.25).astype(F)
    y, st = emulate(mi, x, coef)
    y32, st32 = oracle.biquad_cascade(x, coef)
    y64 = oracle.biquad_cascade_f64(x, coef)
    assert_iir_parity(y, y32, y64, name)


def test_table_shapes(mi):
    L, NT, NLEV, c5, M, p, q = product_tables(mi, [1, 0, 0, 0.5, 0], 0)
    assert (L, NT, NLEV) == (32, 128, 7)
    # one-pole y = x + 0.5 y[-1]: d0' = 0.5 (x + d0); end-state weight of sample k is 0.5^(L-k)
    np.testing.assert_allclose(p, 0.5 ** (L - np.arange(L)), rtol=1e-6)
    np.testing.assert_allclose(M[0], [[0.5 ** L, 0.5 ** (L - 1)], [0, 0]], rtol=1e-6)
    L, NT, NLEV, *_ = product_tables(mi, [1, 0, 0, 0.5, 0], 1)
    assert (L, NT, NLEV) == (8, 64, 6)
