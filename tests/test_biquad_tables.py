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
    L, NT, NM, TAB = [int(v) for v in geo]
    row = np.zeros(TAB, np.float32)
    mi.check(mi.lib.mi_biquad_section_tables(ctypes.byref(chain), variant,
                                             row.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), geo))
    M = row[8:8 + 4 * NM].reshape(NM, 2, 2)          # M[i] = P^(i+1)
    p = row[8 + 4 * NM: 8 + 4 * NM + L]
    qq = row[8 + 4 * NM + L: 8 + 4 * NM + 2 * L]
    return L, NT, NM, row[:5], M, p, qq


def _mv(M, vx, vy, ax, ay):
    """(ax,ay) + M (vx,vy) with the kernel's fma nesting: fma(m00, vx, fma(m01, vy, ax))."""
    return ((M[0, 0] * vx + (M[0, 1] * vy + ax).astype(F)).astype(F),
            (M[1, 0] * vx + (M[1, 1] * vy + ay).astype(F)).astype(F))


def emulate_block(mi, x, coef, state, variant):
    """One kernel launch for one channel, lane for lane: x has at most L*NT samples."""
    cnt = len(x)
    L, NT = (32, 128) if variant == 0 else (8, 64)
    assert cnt <= L * NT
    X = np.zeros(L * NT, F)
    X[:cnt] = x
    X = X.reshape(NT, L).copy()
    t = np.arange(NT)
    l16, row, wave = t & 15, (t & 63) >> 4, t >> 6
    t_last = (cnt - 1) // L
    m_last = cnt - t_last * L
    for s, q in enumerate(coef):
        L_, NT_, NM, c5, M, p, qq = product_tables(mi, q, variant)
        assert (L_, NT_, NM) == (L, NT, 16)
        np.testing.assert_array_equal(c5, np.asarray(q, F))
        b0, b1, b2, a1, a2 = [F(v) for v in q]
        c0, c1 = F(state[s][0]), F(state[s][1])
        # 1. dot products (two accumulators each, as in the kernel)
        z0 = np.zeros(NT, F); z1 = np.zeros(NT, F); w0 = np.zeros(NT, F); w1 = np.zeros(NT, F)
        for k in range(0, L, 4):
            z0 = (p[k] * X[:, k] + z0).astype(F); w0 = (qq[k] * X[:, k] + w0).astype(F)
            z1 = (p[k + 1] * X[:, k + 1] + z1).astype(F); w1 = (qq[k + 1] * X[:, k + 1] + w1).astype(F)
            z0 = (p[k + 2] * X[:, k + 2] + z0).astype(F); w0 = (qq[k + 2] * X[:, k + 2] + w0).astype(F)
            z1 = (p[k + 3] * X[:, k + 3] + z1).astype(F); w1 = (qq[k + 3] * X[:, k + 3] + w1).astype(F)
        z = (z0 + z1).astype(F); w = (w0 + w1).astype(F)
        zc, wc = _mv(M[0], np.array([c0]), np.array([c1]), z[:1], w[:1])
        z[0], w[0] = zc[0], wc[0]
        # 2a. row scan with DPP row_shr d (lanes whose source falls out of the 16-lane row read 0)
        for d, Mi in ((1, M[0]), (2, M[1]), (4, M[3]), (8, M[7])):
            src = t - d
            ok = l16 >= d
            zs = np.where(ok, z[np.maximum(src, 0)], F(0)); ws = np.where(ok, w[np.maximum(src, 0)], F(0))
            z, w = _mv(Mi, zs, ws, z, w)
        # 2b. chain of row totals inside each wave, waves in order
        NW = NT // 64
        cr = np.zeros((NT, 2), F)
        cin = np.zeros(2, F)
        P16 = M[15]
        for wv in range(NW):
            base = 64 * wv
            c = [cin.copy()]
            for r in range(4):
                tx, ty = z[base + 16 * r + 15], w[base + 16 * r + 15]
                nx, ny = _mv(P16, np.array([c[-1][0]]), np.array([c[-1][1]]), np.array([tx]), np.array([ty]))
                c.append(np.array([nx[0], ny[0]], F))
            for r in range(4):
                cr[base + 16 * r: base + 16 * r + 16] = c[r]
            cin = c[4]
        # 2c. lane power
        ML = M[l16]                                   # (NT,2,2)
        zn = (ML[:, 0, 0] * cr[:, 0] + (ML[:, 0, 1] * cr[:, 1] + z).astype(F)).astype(F)
        wn = (ML[:, 1, 0] * cr[:, 0] + (ML[:, 1, 1] * cr[:, 1] + w).astype(F)).astype(F)
        z, w = zn, wn
        d0 = np.where(l16 == 0, cr[:, 0], z[np.maximum(t - 1, 0)]).astype(F)
        d1 = np.where(l16 == 0, cr[:, 1], w[np.maximum(t - 1, 0)]).astype(F)
        d0[0], d1[0] = c0, c1
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
    L, NT, NM, c5, M, p, q = product_tables(mi, [1, 0, 0, 0.5, 0], 0)
    assert (L, NT, NM) == (32, 128, 16)
    # one-pole y = x + 0.5 y[-1]: d0' = 0.5 (x + d0); end-state weight of sample k is 0.5^(L-k)
    np.testing.assert_allclose(p, 0.5 ** (L - np.arange(L)), rtol=1e-6)
    np.testing.assert_allclose(M[0], [[0.5 ** L, 0.5 ** (L - 1)], [0, 0]], rtol=1e-6)
    np.testing.assert_allclose(M[1][0, 0], 0.5 ** (2 * L), rtol=1e-6)
    L, NT, NM, *_ = product_tables(mi, [1, 0, 0, 0.5, 0], 1)
    assert (L, NT, NM) == (8, 64, 16)
