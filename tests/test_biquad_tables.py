"""Host logic of the biquad kernel, checked without a GPU.

The kernel evaluates each TDF-II section chunk-parallel: zero-state end state of every
chunk by two dot products, an inclusive DPP scan over pairs of chunks with powers of P^2 (P = A^L), then the
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


def _cm(r):
    """column-major 2x2 (m00 m10 m01 m11) -> row-major array"""
    return np.array([[r[0], r[2]], [r[1], r[3]]], F)


def product_tables(mi, q, variant):
    from importlib import import_module
    capi = import_module("lsp-dsp-units_amd.capi")
    chain = capi.BiquadX1(*[float(v) for v in q], 0.0, 0.0, 0.0)
    geo = (ctypes.c_uint32 * 4)()
    mi.check(mi.lib.mi_biquad_section_tables(ctypes.byref(chain), variant, None, geo))
    L, NT, NM, TAB = [int(v) for v in geo]
    assert TAB == 96 + 2 * L
    row = np.zeros(TAB, np.float32)
    mi.check(mi.lib.mi_biquad_section_tables(ctypes.byref(chain), variant,
                                             row.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), geo))
    # PW[j] = P^(2^j), j = 0..5 : P, P^2, P^4, P^8, P^16, P^32   (P = A^L)
    PW = [_cm(row[8 + 4 * j: 12 + 4 * j]) for j in range(6)]
    pq = row[32:32 + 2 * L].reshape(L, 2)
    QL = np.array([_cm(row[32 + 2 * L + 4 * i: 36 + 2 * L + 4 * i]) for i in range(NM)], F)     # (P^2)^(i+1)
    return L, NT, NM, row[:5], PW, QL, pq[:, 0].copy(), pq[:, 1].copy()


def _mv(M, vx, vy, ax, ay):
    """(ax,ay) + M (vx,vy) with the kernel's fma nesting: fma(m00, vx, fma(m01, vy, ax))."""
    return ((M[0, 0] * vx + (M[0, 1] * vy + ax).astype(F)).astype(F),
            (M[1, 0] * vx + (M[1, 1] * vy + ay).astype(F)).astype(F))


def emulate_super_block(mi, x, coef, state, L, NW):
    """One super-block of one channel, wave for wave and lane for lane: NW waves, each owning a sub-block of
    64 * 2L samples; every lane owns two adjacent chunks A and B of L samples.  cnt is a multiple of L."""
    cnt = len(x)
    W = 2 * L
    SB = 64 * W
    assert cnt <= NW * SB and cnt % L == 0
    X = np.zeros(NW * SB, F)
    X[:cnt] = x
    X = X.reshape(NW, 64, 2, L).copy()               # [wave][lane][chunk][k]
    t = np.arange(64)
    l16, row = t & 15, t >> 4
    last = cnt - 1
    w_last = last // SB
    t_last = (last - w_last * SB) // W
    save_hi = (last - w_last * SB - t_last * W) >= L
    variant = 0 if L == 16 else 1
    zero = np.zeros(64, F)
    for s, q in enumerate(coef):
        L_, NT_, NM, c5, PW, QLT, p, qq = product_tables(mi, q, variant)
        assert (L_, NT_, NM) == (L, 64, 16)
        np.testing.assert_array_equal(c5, np.asarray(q, F))
        b0, b1, b2, a1, a2 = [F(v) for v in q]
        P, P2 = PW[0], PW[1]
        carry = np.array([state[s][0], state[s][1]], F)
        new_state = None
        for wv in range(NW):
            Xw = X[wv]
            # 1. dot products: accumulators for even and odd k, for the first (A) and second (B) chunk
            acc = np.zeros((2, 2, 64, 2), F)                     # [chunk][parity][lane][z|w]
            for k in range(L):
                for c in range(2):
                    acc[c, k & 1, :, 0] = (p[k] * Xw[:, c, k] + acc[c, k & 1, :, 0]).astype(F)
                    acc[c, k & 1, :, 1] = (qq[k] * Xw[:, c, k] + acc[c, k & 1, :, 1]).astype(F)
            zwA = (acc[0, 0] + acc[0, 1]).astype(F); zwB = (acc[1, 0] + acc[1, 1]).astype(F)
            # 2. pair end state for a zero start; the state entering the wave joins at lane 0
            ex, ey = _mv(P, zwA[:, 0], zwA[:, 1], zwB[:, 0], zwB[:, 1])
            cx = np.where(t == 0, carry[0], F(0)).astype(F); cy = np.where(t == 0, carry[1], F(0)).astype(F)
            ex, ey = _mv(P2, cx, cy, ex, ey)
            # 2a. row scan with DPP row_shr d (lanes whose source falls out of the 16-lane row read 0)
            for d, M in ((1, PW[1]), (2, PW[2]), (4, PW[3]), (8, PW[4])):
                ok = l16 >= d
                xs = np.where(ok, ex[np.maximum(t - d, 0)], F(0)); ys = np.where(ok, ey[np.maximum(t - d, 0)], F(0))
                ex, ey = _mv(M, xs, ys, ex, ey)
            # 2b. row_bcast:15 into rows 1 and 3, with the lane's own (P^2)^(l16+1)
            odd = (row & 1) == 1
            xs = np.where(odd, ex[np.maximum(16 * row - 1, 0)], F(0)); ys = np.where(odd, ey[np.maximum(16 * row - 1, 0)], F(0))
            QL = QLT[l16]
            exn = (QL[:, 0, 0] * xs + (QL[:, 0, 1] * ys + ex).astype(F)).astype(F)
            eyn = (QL[:, 1, 0] * xs + (QL[:, 1, 1] * ys + ey).astype(F)).astype(F)
            ex, ey = exn, eyn
            # 2c. row_bcast:31 into rows 2 and 3; row 3 through one more P^32
            hi = row >= 2
            xs = np.where(hi, ex[31], F(0)); ys = np.where(hi, ey[31], F(0))
            x2, y2 = _mv(PW[5], xs, ys, zero, zero)
            xs = np.where(row == 3, x2, xs); ys = np.where(row == 3, y2, ys)
            exn = (QL[:, 0, 0] * xs + (QL[:, 0, 1] * ys + ex).astype(F)).astype(F)
            eyn = (QL[:, 1, 0] * xs + (QL[:, 1, 1] * ys + ey).astype(F)).astype(F)
            ex, ey = exn, eyn
            # 3. start states (wave_shr:1, lane 0 keeps the carried state)
            sx0 = np.where(t == 0, carry[0], ex[np.maximum(t - 1, 0)]).astype(F)
            sy0 = np.where(t == 0, carry[1], ey[np.maximum(t - 1, 0)]).astype(F)
            bx, by = _mv(P, sx0, sy0, zwA[:, 0], zwA[:, 1])
            d0 = np.stack([sx0, bx], 1); d1 = np.stack([sy0, by], 1)
            for k in range(L):
                xx = Xw[:, :, k]
                y = (b0 * xx + d0).astype(F)
                tt = (b1 * xx + d1).astype(F)
                d0 = (a1 * y + tt).astype(F)
                d1 = (a2 * y + (b2 * xx).astype(F)).astype(F)
                Xw[:, :, k] = y
            if wv == w_last:
                h = 1 if save_hi else 0
                new_state = (d0[t_last, h], d1[t_last, h])
            carry = np.array([ex[63], ey[63]], F)                # what the next wave starts from
        state[s][0], state[s][1] = new_state
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
    L, NT, NM, c5, PW, QL, p, q = product_tables(mi, [1, 0, 0, 0.5, 0], 0)
    assert (L, NT, NM) == (16, 64, 16)
    # one-pole y = x + 0.5 y[-1]: d0' = 0.5 (x + d0); end-state weight of sample k is 0.5^(L-k)
    np.testing.assert_allclose(p, 0.5 ** (L - np.arange(L)), rtol=1e-6)
    np.testing.assert_allclose(PW[0], [[0.5 ** L, 0.5 ** (L - 1)], [0, 0]], rtol=1e-6)
    np.testing.assert_allclose(PW[1][0, 0], 0.5 ** (2 * L), rtol=1e-6)
    np.testing.assert_allclose(PW[2][0, 0], 0.5 ** (4 * L), rtol=1e-6)
    np.testing.assert_allclose(QL[0], PW[1], rtol=1e-6)
    np.testing.assert_allclose(QL[1][0, 0], 0.5 ** (4 * L), rtol=1e-6)
    L, NT, NM, *_ = product_tables(mi, [1, 0, 0, 0.5, 0], 1)
    assert (L, NT, NM) == (8, 64, 16)
