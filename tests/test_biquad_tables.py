"""Host logic of the biquad kernel, checked without a GPU.

The kernel evaluates each TDF-II section chunk-parallel: zero-state end state of every
chunk by two dot products, an inclusive scan over pairs of chunks with powers of P^2 (P = A^L), then the
exact recurrence from the scanned start states.  This test takes the REAL per-section tables the product
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
    P = row[8:12].reshape(2, 2)                      # A^L
    Q = row[12:12 + 4 * NM].reshape(NM, 2, 2)        # Q[i] = (P^2)^(i+1)
    Q64 = row[76:80].reshape(2, 2)                   # (P^2)^64
    p = row[80: 80 + L]
    qq = row[80 + L: 80 + 2 * L]
    return L, NT, NM, row[:5], P, Q, Q64, p, qq


def _mv(M, vx, vy, ax, ay):
    """(ax,ay) + M (vx,vy) with the kernel's fma nesting: fma(m00, vx, fma(m01, vy, ax))."""
    return ((M[0, 0] * vx + (M[0, 1] * vy + ax).astype(F)).astype(F),
            (M[1, 0] * vx + (M[1, 1] * vy + ay).astype(F)).astype(F))


def _mv1(M, v, a):
    x, y = _mv(M, np.array([v[0]], F), np.array([v[1]], F), np.array([a[0]], F), np.array([a[1]], F))
    return np.array([x[0], y[0]], F)


def emulate_super_block(mi, x, coef, state, L, NW):
    """One super-block of one channel, wave for wave and lane for lane: NW waves, each owning a sub-block of
    64 * 2L samples; every lane owns two adjacent chunks A and B of L samples."""
    cnt = len(x)
    W = 2 * L
    SB = 64 * W
    assert cnt <= NW * SB
    X = np.zeros(NW * SB, F)
    X[:cnt] = x
    X = X.reshape(NW, 64, 2, L).copy()               # [wave][lane][chunk][k]
    t = np.arange(64)
    l16, row = t & 15, t >> 4
    last = cnt - 1
    w_last = last // SB
    t_last = (last - w_last * SB) // W
    m_last = last - w_last * SB - t_last * W + 1
    variant = 0 if L == 16 else 1
    for s, q in enumerate(coef):
        L_, NT_, NM, c5, P, Q, Q64, p, qq = product_tables(mi, q, variant)
        assert (L_, NT_, NM) == (L, 64, 16)
        np.testing.assert_array_equal(c5, np.asarray(q, F))
        b0, b1, b2, a1, a2 = [F(v) for v in q]
        Q16 = Q[15]
        carried = np.array([state[s][0], state[s][1]], F)
        pre = []
        for wv in range(NW):
            Xw = X[wv]
            # 1. dot products (two accumulators each, both chunks at once as in the packed kernel)
            z0 = np.zeros((64, 2), F); z1 = np.zeros((64, 2), F); w0 = np.zeros((64, 2), F); w1 = np.zeros((64, 2), F)
            for k in range(0, L, 4):
                z0 = (p[k] * Xw[:, :, k] + z0).astype(F); w0 = (qq[k] * Xw[:, :, k] + w0).astype(F)
                z1 = (p[k + 1] * Xw[:, :, k + 1] + z1).astype(F); w1 = (qq[k + 1] * Xw[:, :, k + 1] + w1).astype(F)
                z0 = (p[k + 2] * Xw[:, :, k + 2] + z0).astype(F); w0 = (qq[k + 2] * Xw[:, :, k + 2] + w0).astype(F)
                z1 = (p[k + 3] * Xw[:, :, k + 3] + z1).astype(F); w1 = (qq[k + 3] * Xw[:, :, k + 3] + w1).astype(F)
            z = (z0 + z1).astype(F); w = (w0 + w1).astype(F)
            # 2. pair end state for a zero start
            ex, ey = _mv(P, z[:, 0], w[:, 0], z[:, 1], w[:, 1])
            # 2a. row scan with DPP row_shr d (lanes whose source falls out of the 16-lane row read 0)
            for d, Qi in ((1, Q[0]), (2, Q[1]), (4, Q[3]), (8, Q[7])):
                src = t - d
                ok = l16 >= d
                xs = np.where(ok, ex[np.maximum(src, 0)], F(0)); ys = np.where(ok, ey[np.maximum(src, 0)], F(0))
                ex, ey = _mv(Qi, xs, ys, ex, ey)
            tot = [np.array([ex[16 * r + 15], ey[16 * r + 15]], F) for r in range(4)]
            # zero-start end state of the whole sub-block (published to the later waves)
            u = tot[0]
            for r in (1, 2, 3):
                u = _mv1(Q16, u, tot[r])
            pre.append((z, w, ex, ey, tot, u))
        for wv in range(NW):
            z, w, ex, ey, tot, _ = pre[wv]
            Xw = X[wv]
            c0 = carried.copy()
            for v in range(wv):
                c0 = _mv1(Q64, c0, pre[v][5])
            c = [c0]
            for r in range(3):
                c.append(_mv1(Q16, c[-1], tot[r]))
            cr = np.array([c[r] for r in row], F)
            # 2c. lane power
            QL = Q[l16]
            exn = (QL[:, 0, 0] * cr[:, 0] + (QL[:, 0, 1] * cr[:, 1] + ex).astype(F)).astype(F)
            eyn = (QL[:, 1, 0] * cr[:, 0] + (QL[:, 1, 1] * cr[:, 1] + ey).astype(F)).astype(F)
            # 3. start states
            sx0 = np.where(l16 == 0, cr[:, 0], exn[np.maximum(t - 1, 0)]).astype(F)
            sy0 = np.where(l16 == 0, cr[:, 1], eyn[np.maximum(t - 1, 0)]).astype(F)
            bx, by = _mv(P, sx0, sy0, z[:, 0], w[:, 0])
            d0 = np.stack([sx0, bx], 1); d1 = np.stack([sy0, by], 1)
            f0 = d0.copy(); f1 = d1.copy()
            for k in range(L):
                xx = Xw[:, :, k]
                y = (b0 * xx + d0).astype(F)
                tt = (b1 * xx + d1).astype(F)
                d0 = (a1 * y + tt).astype(F)
                d1 = (a2 * y + (b2 * xx).astype(F)).astype(F)
                Xw[:, :, k] = y
                if k + 1 == m_last or k + 1 + L == m_last:
                    f0 = d0.copy(); f1 = d1.copy()
            if cnt == NW * SB:
                if wv == NW - 1:
                    state[s][0] = d0[63, 1]; state[s][1] = d1[63, 1]
            elif wv == w_last:
                h = 0 if m_last <= L else 1
                state[s][0] = f0[t_last, h]; state[s][1] = f1[t_last, h]
    return X.reshape(-1)[:cnt]


def emulate_tail(x, coef, state):
    """biquad_tail_kernel: the samples % L leftover samples, one after the other."""
    y = x.copy()
    for s, q in enumerate(coef):
        b0, b1, b2, a1, a2 = [F(v) for v in q]
        d0, d1 = F(state[s][0]), F(state[s][1])
        for k in range(len(y)):
            xx = y[k]
            out = F(b0 * xx + d0)
            tt = F(b1 * xx + d1)
            d0 = F(a1 * out + tt)
            d1 = F(a2 * out + F(b2 * xx))
            y[k] = out
        state[s][0], state[s][1] = d0, d1
    return y


def emulate(mi, x, coef):
    """Variant selection of mi_biquad_bank_process, then super-block after super-block, then the tail."""
    st = [[F(0), F(0)] for _ in coef]
    out = np.empty_like(x)
    if len(x) <= 1024:
        L, NW = 8, 1
    elif len(x) <= 2048:
        L, NW = 8, 2
    else:
        L, NW = 16, 2
    body = len(x) - len(x) % L
    sup = NW * 64 * 2 * L
    for done in range(0, body, sup):
        step = min(sup, body - done)
        out[done:done + step] = emulate_super_block(mi, x[done:done + step], coef, st, L, NW)
    if body < len(x):
        out[body:] = emulate_tail(x[body:], coef, st)
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
@pytest.mark.parametrize("n", [4096, 5000, 300, 1030, 2048, 9000, 5, 4111])
def test_chunked_form_matches_sequential(mi, name, ftype, slope, freq, gain, q, n):
    coef = wl.design(ftype, slope, freq, freq, gain, q)
    x = (np.random.default_rng(42).standard_normal(n) * 0.25).astype(F)
    y, st = emulate(mi, x, coef)
    y32, st32 = oracle.biquad_cascade(x, coef)
    y64 = oracle.biquad_cascade_f64(x, coef)
    assert_iir_parity(y, y32, y64, name)


def test_table_shapes(mi):
    L, NT, NM, c5, P, Q, Q64, p, q = product_tables(mi, [1, 0, 0, 0.5, 0], 0)
    assert (L, NT, NM) == (16, 64, 16)
    # one-pole y = x + 0.5 y[-1]: d0' = 0.5 (x + d0); end-state weight of sample k is 0.5^(L-k)
    np.testing.assert_allclose(p, 0.5 ** (L - np.arange(L)), rtol=1e-6)
    np.testing.assert_allclose(P, [[0.5 ** L, 0.5 ** (L - 1)], [0, 0]], rtol=1e-6)
    np.testing.assert_allclose(Q[0][0, 0], 0.5 ** (2 * L), rtol=1e-6)
    np.testing.assert_allclose(Q[1][0, 0], 0.5 ** (4 * L), rtol=1e-6)
    np.testing.assert_allclose(Q64[0, 0], 0.5 ** (128 * L), rtol=1e-5, atol=0)
    L, NT, NM, *_ = product_tables(mi, [1, 0, 0, 0.5, 0], 1)
    assert (L, NT, NM) == (8, 64, 16)
