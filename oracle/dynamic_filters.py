"""
ORACLE -- test infrastructure only (see oracle/__init__.py).

numpy restatement of lsp::dspu::DynamicFilters (reference: src/main/filters/DynamicFilters.cpp,
include/lsp-plug.in/dsp-units/filters/DynamicFilters.h): filters whose GAIN changes from sample to sample
(dynamic equalisers, de-essers): for every sample the analog cascades are rebuilt from the gain value
(build_filter_bank, :625-1738), transformed to digital sections (dsp::bilinear_transform_x* /
dsp::matched_transform_x*, :265-303) and applied to that sample (dsp::dyn_biquad_process_x*).

Layout note.  The reference fills its cascade array diagonally (row = pipeline step, column = cascade, the values of
sample n reach column j at row n + j: :320-369, :542-556 and the `dst[(nc+1)*j]`, `c += nc` walks) because
lsp-dsp-lib's x8/x4/x2 kernels pipeline the sections over SIMD lanes; what cascade J of sample n contains does not depend
on that layout, and that is what `cascades()` returns: arrays indexed [J][n].  The grouping of cascades into 8/4/2/1
packs (quantify(), :191-202) is a property of the CPU kernels too and has no arithmetic effect.

The three lsp-dsp-lib primitives are un-vendored (modules.mk:29-33) and no reference test touches this unit, so they are
restated from their published algorithm and pinned by properties (tests/test_oracle_dynamic_filters.py):
  * bilinear_transform_x1: the formulas of Filter::bilinear_transform (Filter.cpp:2192-2267) in float;
  * matched_transform_x1: Filter::matched_transform (Filter.cpp:2291-2416), per sample;
  * dyn_biquad_process_x1: oracle/biquad_oracle.c::orc_dyn_biquad_cascade.
Parity unpinned by reference vectors (there are none for this unit): see DESIGN.md.

FLT_*_RLC_ENVELOPE is not restated: the reference's builder leaves t[2] / b[2] of its second half unset and keys one
branch on a group-relative index (:1022-1085, "TODO: test this"), so its result depends on stale memory.
"""
import numpy as np

from . import binding
from . import filter_design as fd

F = np.float32
PI = F(np.pi)
PI_2 = F(np.pi / 2)
TWO_PI = F(2 * np.pi)
FILTER_CHAINS_MAX = 0x80
BUF_SIZE = 0x400


def _f(x):
    return np.asarray(x, dtype=np.float32)


def _iroot(x, n):
    """dsp::irootf(x, n): the n-th root (lsp-dsp-lib; restated as exp(log(x) / n) in float)."""
    return np.exp(np.log(_f(x)) / F(n)).astype(np.float32)


def cascade_count(ftype, slope):
    """Number of cascades of a filter (the sum of build_filter_bank's nj over its calls)."""
    base = ftype if (ftype & 1) else ftype - 1
    n = {
        fd.FLT_BT_AMPLIFIER: 1,
        fd.FLT_BT_RLC_LOPASS: (slope >> 1) + (slope & 1), fd.FLT_BT_RLC_HIPASS: (slope >> 1) + (slope & 1),
        fd.FLT_BT_RLC_LOSHELF: slope, fd.FLT_BT_RLC_HISHELF: slope,
        fd.FLT_BT_RLC_LADDERPASS: 2 * slope, fd.FLT_BT_RLC_LADDERREJ: 2 * slope,
        fd.FLT_BT_RLC_BANDPASS: slope, fd.FLT_BT_RLC_BELL: slope, fd.FLT_BT_RLC_RESONANCE: slope, fd.FLT_BT_RLC_NOTCH: 1,
        fd.FLT_BT_BWC_LOPASS: (slope >> 1) + (slope & 1), fd.FLT_BT_BWC_HIPASS: (slope >> 1) + (slope & 1),
        fd.FLT_BT_BWC_LOSHELF: slope, fd.FLT_BT_BWC_HISHELF: slope,
        fd.FLT_BT_BWC_LADDERPASS: 2 * slope, fd.FLT_BT_BWC_LADDERREJ: 2 * slope,
        fd.FLT_BT_BWC_BELL: 2 * slope, fd.FLT_BT_BWC_BANDPASS: 2 * slope,
        fd.FLT_BT_LRX_LOPASS: 2 * slope, fd.FLT_BT_LRX_HIPASS: 2 * slope,
        fd.FLT_BT_LRX_LOSHELF: 2 * slope, fd.FLT_BT_LRX_HISHELF: 2 * slope,
        fd.FLT_BT_LRX_LADDERPASS: 4 * slope, fd.FLT_BT_LRX_LADDERREJ: 4 * slope,
        fd.FLT_BT_LRX_BELL: 4 * slope, fd.FLT_BT_LRX_BANDPASS: 4 * slope,
    }.get(base, 0)
    return min(n, FILTER_CHAINS_MAX)


def cascades(ftype, slope, freq2, quality, gains):
    """Analog cascades of every sample: (T, B), float32 [nc][n][3].  `freq2` is the TRANSFORMED second frequency of
    set_params (:170-178); `gains` the per-sample gain vector of process()."""
    g = _f(gains)
    n = g.size
    base = ftype if (ftype & 1) else ftype - 1
    nc = cascade_count(ftype, slope)
    T = np.zeros((nc, n, 3), np.float32)
    B = np.zeros((nc, n, 3), np.float32)
    one, zero = np.ones(n, np.float32), np.zeros(n, np.float32)
    Q = F(quality)
    xf = F(freq2)

    def put(J, t, b):
        for i in range(3):
            T[J, :, i] = _f(t[i]) * one if np.ndim(t[i]) == 0 else t[i]
            B[J, :, i] = _f(b[i]) * one if np.ndim(b[i]) == 0 else b[i]

    if base == fd.FLT_BT_AMPLIFIER:                                                    # :633-657
        put(0, (g, 0, 0), (1, 0, 0))

    elif base in (fd.FLT_BT_RLC_LOPASS, fd.FLT_BT_RLC_HIPASS):                         # :660-732
        lo = base == fd.FLT_BT_RLC_LOPASS
        for J in range(nc):
            if J == 0 and (slope & 1):
                put(J, (g, 0, 0) if lo else (0, g, 0), (1, 1, 0))
            else:
                k = F(2.0) / (F(1.0) + Q)
                t = [F(1) * one if lo else zero, zero, zero if lo else F(1) * one]
                if J == 0:                                                              # "Patch volume"
                    t = [v * g for v in t]
                put(J, t, (1, k, 1))

    elif base in (fd.FLT_BT_RLC_LOSHELF, fd.FLT_BT_RLC_HISHELF):                       # :734-785
        gs = np.sqrt(g)
        fg = np.exp(np.log(gs) / F(slope * 2)).astype(np.float32)
        k = F(2.0) / (F(1.0) + Q)
        a, b = (fg, k * one, F(1) / fg), (F(1) / fg, k * one, fg)
        for J in range(nc):
            t, bb = (a, b) if base == fd.FLT_BT_RLC_LOSHELF else (b, a)
            if J == 0:
                t = tuple(v * gs for v in t)
            put(J, t, bb)

    elif base in (fd.FLT_BT_RLC_LADDERPASS, fd.FLT_BT_RLC_LADDERREJ):                  # :790-876
        rej = base == fd.FLT_BT_RLC_LADDERREJ
        s2 = F(slope * 2)
        k = F(2.0) / (F(1.0) + Q)
        sq, isq = np.sqrt(g), np.sqrt(F(1.0) / g).astype(np.float32)
        for J in range(nc):
            if J & 1:                                                                   # second shelf, always a hi-shelf
                gain = sq if rej else isq
                fg = np.exp(np.log(gain) / s2).astype(np.float32)
                cb = (fg, F(2.0) * xf / (F(1.0) + Q) * one, xf * xf / fg)
                ct = (F(1) / fg, F(2.0) * xf / (F(1.0) + Q) * one, fg * xf * xf)
            else:
                gain1, gain2 = (isq, sq) if rej else (sq, isq)
                fg = np.exp(np.log(gain2 if rej else gain1) / s2).astype(np.float32)
                gain = gain2 if rej else gain1
                a, b = (fg, k * one, F(1) / fg), (F(1) / fg, k * one, fg)
                ct, cb = (a, b) if rej else (b, a)
            if (J >> 1) == 0:
                ct = tuple(v * gain for v in ct)
            put(J, ct, cb)

    elif base == fd.FLT_BT_RLC_BANDPASS:                                                # :878-911
        f2 = F(1.0) / xf
        k = (F(1.0) + f2) / (F(1.0) + Q)
        for J in range(nc):
            t1 = np.exp(F(slope) * np.log(k)).astype(np.float32) * g if J == 0 else one
            put(J, (0, t1, 0), (f2, k, 1))

    elif base in (fd.FLT_BT_RLC_BELL, fd.FLT_BT_RLC_RESONANCE):                         # :913-991
        fg = np.exp(np.log(g) / F(slope)).astype(np.float32)
        tsin = np.sin(np.arctan(fg)).astype(np.float32)
        tcos = np.sqrt(F(1.0) - tsin * tsin)
        if base == fd.FLT_BT_RLC_BELL:
            k = F(2.0) * (F(1.0) / fg + fg) / (F(1.0) + (F(2.0) * Q) / F(slope))
        else:
            k = F(2.0) / (F(1.0) + Q) * one
        for J in range(nc):
            put(J, (1, k * tsin, 1), (1, k * tcos, 1))

    elif base == fd.FLT_BT_RLC_NOTCH:                                                   # :993-1020
        put(0, (g, 0, g), (1, F(2.0) / (F(1.0) + Q), 1))

    elif base in (fd.FLT_BT_BWC_LOPASS, fd.FLT_BT_BWC_HIPASS, fd.FLT_BT_LRX_LOPASS, fd.FLT_BT_LRX_HIPASS):   # :1090-1170, :1509-1563
        lrx = base in (fd.FLT_BT_LRX_LOPASS, fd.FLT_BT_LRX_HIPASS)
        hi = base in (fd.FLT_BT_BWC_HIPASS, fd.FLT_BT_LRX_HIPASS)
        k = F(1.0) / (F(1.0) + Q)
        for J in range(nc):
            if not lrx and J == 0 and (slope & 1):
                put(J, (0, g, 0) if hi else (g, 0, 0), (1, 1, 0))
                continue
            if lrx:
                theta = F(((J & ~1) + 1) * PI_2) / F(slope * 2)
            else:
                theta = F((2 * (J - (slope & 1)) + 1) * PI_2) / F(slope)
            tsin = np.sin(F(theta), dtype=np.float32)
            tcos = np.sqrt(F(1.0) - tsin * tsin)
            kf1 = F(1.0) / (tsin * tsin + k * k * tcos * tcos)
            lead = g if J == 0 else one
            if hi:
                put(J, (0, 0, lead), (kf1, F(2.0) * k * tcos * kf1, 1))
            else:
                put(J, (lead, 0, 0), (1, F(2.0) * k * tcos * kf1, kf1))

    elif base in (fd.FLT_BT_BWC_HISHELF, fd.FLT_BT_BWC_LOSHELF):                        # :1172-1233
        gain = np.sqrt(g)
        fg = np.exp(np.log(gain) / F(2.0 * slope)).astype(np.float32)
        k = F(1.0) / (F(1.0) + Q * (F(1.0) - np.exp(F(2.0) - gain - F(1.0) / gain)))
        for J in range(nc):
            theta = F((2 * J + 1) * PI_2) / F(2 * slope)
            tsin = np.sin(F(theta), dtype=np.float32)
            tcos = np.sqrt(F(1.0) - tsin * tsin)
            kf = tsin * tsin + k * k * tcos * tcos
            a, b = (kf / fg, F(2.0) * k * tcos, fg), (fg, F(2.0) * k * tcos, kf / fg)
            t, bb = (a, b) if base == fd.FLT_BT_BWC_HISHELF else (b, a)
            if J == 0:
                t = tuple(v * gain for v in t)
            put(J, t, bb)

    elif base in (fd.FLT_BT_BWC_LADDERPASS, fd.FLT_BT_BWC_LADDERREJ):                   # :1235-1347
        passing = base == fd.FLT_BT_BWC_LADDERPASS
        rc = F(1.0) / F(slope * 2)
        for J in range(nc):
            theta = F(((J & ~1) + 1) * PI_2) * rc
            tcos = np.cos(F(theta), dtype=np.float32)
            tcos2 = tcos * tcos
            tsin2 = F(1.0) - tcos2
            if J & 1:                                                                   # second shelf, always a hi-shelf
                xf2 = xf * xf
                xtcos = F(2.0) * tcos * xf
                gain = np.sqrt(g) if passing else np.sqrt(F(1.0) / g).astype(np.float32)
                fg = np.exp(np.log(gain) * rc).astype(np.float32)
                k = F(1.0) / (F(1.0) + Q * (F(1.0) - np.exp(F(2.0) - gain - F(1.0) / gain)))
                kf = tsin2 + k * k * tcos2
                cb = (kf / fg, k * xtcos, fg * xf2)
                ct = (fg, k * xtcos, (kf / fg) * xf2)
                if not (J & ~1):
                    ct = tuple(v * (F(1.0) / gain) for v in ct)
            else:
                xtcos = F(2.0) * tcos
                gain = np.sqrt(g)
                k = F(1.0) / (F(1.0) + Q * (F(1.0) - np.exp(F(2.0) - gain - F(1.0) / gain)))
                fg = np.exp(np.log(gain) * rc).astype(np.float32)
                kf = tsin2 + k * k * tcos2
                a, b = (kf / fg, k * xtcos, fg), (fg, k * xtcos, kf / fg)
                ct, cb = (a, b) if passing else (b, a)
                if not (J & ~1):
                    ct = tuple(v * gain for v in ct)
            put(J, ct, cb)

    elif base in (fd.FLT_BT_BWC_BELL, fd.FLT_BT_LRX_BELL):                               # :1349-1442, :1573-1666
        lrx = base == fd.FLT_BT_LRX_BELL
        sl = slope * (4 if lrx else 2)
        k = F(1.0) / (F(1.0) + Q)
        fg = np.exp(np.log(g) / F(sl)).astype(np.float32)
        up = g >= F(1.0)
        for J in range(nc):
            theta = F((((J & ~3) + 2) if lrx else ((J & ~1) + 1)) * PI_2) / F(sl)
            tsin = np.sin(F(theta), dtype=np.float32)
            tcos = np.sqrt(F(1.0) - tsin * tsin)
            kf = tsin * tsin + k * k * tcos * tcos
            c2 = F(2.0) * k * tcos
            if J & 1:
                t = (one, np.where(up, c2 / fg, c2), np.where(up, kf / (fg * fg), kf))
                b = (one, np.where(up, c2, c2 * fg), np.where(up, kf, kf * fg * fg))
            else:
                t = (one, np.where(up, c2 * fg / kf, c2 / kf), np.where(up, F(1.0) * fg * fg / kf, F(1.0) / kf))
                b = (one, np.where(up, c2 / kf, c2 / (fg * kf)), np.where(up, F(1.0) / kf, F(1.0) / (fg * fg * kf)))
            put(J, tuple(_f(v) for v in t), tuple(_f(v) for v in b))

    elif base in (fd.FLT_BT_BWC_BANDPASS, fd.FLT_BT_LRX_BANDPASS):                       # :1444-1505, :1668-1730
        lrx = base == fd.FLT_BT_LRX_BANDPASS
        sl = slope * (4 if lrx else 2)
        k = F(1.0) / (F(1.0) + Q)
        for J in range(nc):
            theta = F((((J & ~3) + 2) if lrx else ((J & ~1) + 1)) * PI_2) / F(sl)
            tsin = np.sin(F(theta), dtype=np.float32)
            tcos = np.sqrt(F(1.0) - tsin * tsin)
            kf1 = F(1.0) / (tsin * tsin + k * k * tcos * tcos)
            if J & 1:                                                                   # hi-pass cascade
                put(J, (1, 0, 0), (1, F(2.0) * k * tcos * xf * kf1, xf * xf * kf1))
            else:
                put(J, (0, 0, g if J == 0 else one), (kf1, F(2.0) * k * tcos * kf1, 1))

    elif base in (fd.FLT_BT_LRX_HISHELF, fd.FLT_BT_LRX_LOSHELF):                         # build_lrx_shelf_filter_bank :509-623
        b3 = np.sqrt(g)
        gain = np.sqrt(b3)
        fg = _iroot(np.sqrt(gain), slope)
        k = F(1.0) / (F(1.0) + Q * (F(1.0) - np.exp(F(2.0) - gain - F(1.0) / gain)))
        for J in range(nc):
            theta = F(((J & ~1) + 1) * PI_2) / F(2 * slope)
            tcos = np.cos(F(theta), dtype=np.float32)
            tcos2 = tcos * tcos
            tsin2 = F(1.0) - tcos2
            kf = tsin2 + k * k * tcos2
            a, b = (kf * (F(1.0) / fg), k * (F(2.0) * tcos), fg), (fg, k * (F(2.0) * tcos), kf * (F(1.0) / fg))
            t, bb = (a, b) if base == fd.FLT_BT_LRX_HISHELF else (b, a)
            if J == 0:
                t = tuple(v * b3 for v in t)
            put(J, t, bb)

    elif base in (fd.FLT_BT_LRX_LADDERPASS, fd.FLT_BT_LRX_LADDERREJ):                    # :320-507
        passing = base == fd.FLT_BT_LRX_LADDERPASS
        sl = slope * 4
        gain = np.sqrt(g)
        igain = F(1.0) / gain
        fg = _iroot(gain, sl)
        ifg = F(1.0) / fg
        k = F(1.0) / (F(1.0) + Q * (F(1.0) - np.exp(F(2.0) - gain - igain)))
        xf2 = xf * xf
        for J in range(nc):
            theta = F(((J & ~3) + 2) * PI_2) / F(sl)
            tcos = np.cos(F(theta), dtype=np.float32)
            tcos2 = tcos * tcos
            tsin2 = F(1.0) - tcos2
            xtcos = F(2.0) * tcos
            xtcos_xf = F(2.0) * tcos * xf
            kf = tsin2 + k * k * tcos2
            if passing:
                if J & 1:
                    gn = igain
                    b0 = kf * ifg
                    b1 = k * xtcos_xf
                    ct, cb = (fg, b1, b0 * xf2), (b0, b1, fg * xf2)
                else:
                    gn = gain
                    t0 = kf * ifg
                    t1 = k * xtcos
                    ct, cb = (t0, t1, fg), (fg, t1, t0)
            else:
                gn = gain
                if J & 1:
                    b0 = kf * fg
                    b1 = k * xtcos_xf
                    ct, cb = (ifg, b1, b0 * xf2), (b0, b1, ifg * xf2)
                else:
                    b0 = kf * ifg
                    b1 = k * xtcos
                    ct, cb = (fg, b1, b0), (b0, b1, fg)
            if not (J & ~1):
                ct = tuple(v * gn for v in ct)
            put(J, ct, cb)
    return T, B


def bilinear(T, B, kf):
    """dsp::bilinear_transform_x1 for every cascade and sample (formulas of Filter.cpp:2225-2262 in float):
    returns coef [nc][n][5] = {b0, b1, b2, a1, a2}, denominator signs negated."""
    kf = F(kf)
    kf2 = kf * kf
    T0, T1, T2 = T[..., 0], T[..., 1] * kf, T[..., 2] * kf2
    B0, B1, B2 = B[..., 0], B[..., 1] * kf, B[..., 2] * kf2
    N = (F(1.0) / (B0 + B1 + B2)).astype(np.float32)
    out = np.empty(T.shape[:2] + (5,), np.float32)
    out[..., 0] = (T0 + T1 + T2) * N
    out[..., 1] = F(2.0) * (T0 - T2) * N
    out[..., 2] = (T0 - T1 + T2) * N
    out[..., 3] = F(2.0) * (B2 - B0) * N
    out[..., 4] = (B1 - B2 - B0) * N
    return out


def _matched_poly(p, f, td):
    """One side (numerator or denominator) of Filter::matched_transform, vectorised: p [..., 3] -> Q [..., 3]."""
    p0, p1, p2 = p[..., 0], p[..., 1], p[..., 2]
    Q = np.zeros(p.shape, np.float32)
    with np.errstate(all="ignore"):
        # first order (p2 == 0, p1 != 0)
        k1 = p1 / f
        R = -p0 / k1
        q_first = np.stack([k1, -k1 * np.exp(R * td), np.zeros_like(k1)], -1)
        # second order
        k = p2
        qa = F(1.0) / (f * f)
        qb = p1 / (f * p2)
        qc = p0 / p2
        D = qb * qb - F(4.0) * qa * qc
        Ds = np.sqrt(np.abs(D))
        R0 = (-qb - Ds) / (F(2.0) * qa)
        R1 = (-qb + Ds) / (F(2.0) * qa)
        q_real = np.stack([k, -k * (np.exp(R0 * td) + np.exp(R1 * td)), k * np.exp((R0 + R1) * td)], -1)
        Rc = -qb / (F(2.0) * qa)
        Kc = Ds / (F(2.0) * qa)
        q_cplx = np.stack([k, F(-2.0) * k * np.exp(Rc * td) * np.cos(Kc * td), k * np.exp(F(2.0) * Rc * td)], -1)
    zero_order = (p2 == 0) & (p1 == 0)
    first = (p2 == 0) & (p1 != 0)
    second = p2 != 0
    Q[zero_order, 0] = p0[zero_order]
    Q[first] = q_first[first]
    Q[second & (D >= 0)] = q_real[second & (D >= 0)]
    Q[second & (D < 0)] = q_cplx[second & (D < 0)]
    return Q.astype(np.float32)


def matched(T, B, f, td):
    """dsp::matched_transform_x1(bq, cascades, f, td) for every cascade and sample: Filter::matched_transform
    (Filter.cpp:2291-2416) with td = 2 pi / sample_rate; the amplitude reference point is w = 0.1 f td."""
    f = F(f)
    td = F(td)
    out = np.empty(T.shape[:2] + (5,), np.float32)
    P, A, I = [], [], []
    for p in (T, B):
        Q = _matched_poly(p, f, td)
        w = 0.1 * np.float64(f) * np.float64(td)
        Q64 = Q.astype(np.float64)
        re = Q64[..., 0] * np.cos(2.0 * w) + Q64[..., 1] * np.cos(w) + Q64[..., 2]
        im = Q64[..., 0] * np.sin(2.0 * w) + Q64[..., 1] * np.sin(w)
        A.append(np.sqrt(re * re + im * im))
        re = p[..., 0].astype(np.float64) - p[..., 2].astype(np.float64) * 0.01
        im = p[..., 1].astype(np.float64) * 0.1
        I.append(np.sqrt(re * re + im * im))
        P.append(Q)
    with np.errstate(all="ignore"):
        AN = (A[1] * I[0]) / (A[0] * I[1])
        N = 1.0 / P[1][..., 0].astype(np.float64)
    P0, P1 = P[0].astype(np.float64), P[1].astype(np.float64)
    out[..., 0] = P0[..., 0] * N * AN
    out[..., 1] = P0[..., 1] * N * AN
    out[..., 2] = P0[..., 2] * N * AN
    out[..., 3] = -P1[..., 1] * N
    out[..., 4] = -P1[..., 2] * N
    return out


class DynamicFilters:
    """lsp::dspu::DynamicFilters (one object: `filters` filters with their own memory)."""

    def __init__(self, filters):
        self.n = int(filters)
        self.sample_rate = 0
        self.params = [dict(nType=fd.FLT_NONE, nSlope=0, fFreq=F(0), fFreq2=F(0), fGain=F(0), fQuality=F(0)) for _ in range(self.n)]
        self.active = [False] * self.n
        self.memory = np.zeros((self.n, FILTER_CHAINS_MAX, 2), np.float32)
        self.memory64 = np.zeros((self.n, FILTER_CHAINS_MAX, 2), np.float64)
        self.clear_mem = False

    def set_sample_rate(self, sr):
        self.sample_rate = int(sr)

    def set_filter_active(self, fid, active):
        if fid >= self.n:
            return False
        self.active[fid] = True                                 # the reference sets true whatever is asked (.h:147-153)
        return True

    def set_params(self, fid, ntype, slope, freq, freq2, gain, quality):      # :127-181
        if fid >= self.n:
            return False
        if ntype != fd.FLT_NONE and cascade_count(ntype, max(int(slope), 1)) == 0:
            raise ValueError("filter type %d is not restated (see the module header)" % ntype)
        p = self.params[fid]
        if p["nType"] != ntype:
            self.clear_mem = True
        freq, freq2 = F(freq), F(freq2)
        base = ntype if (ntype & 1) else ntype - 1
        swap = (fd.FLT_BT_RLC_LADDERPASS, fd.FLT_BT_RLC_LADDERREJ, fd.FLT_BT_RLC_BANDPASS, fd.FLT_BT_BWC_LADDERPASS,
                fd.FLT_BT_BWC_LADDERREJ, fd.FLT_BT_BWC_BANDPASS, fd.FLT_BT_LRX_LADDERPASS, fd.FLT_BT_LRX_LADDERREJ,
                fd.FLT_BT_LRX_BANDPASS)
        if ntype != fd.FLT_NONE and base in swap and freq2 < freq:
            freq, freq2 = freq2, freq
        with np.errstate(all="ignore"):
            if ntype & 1:
                nf = F(PI / F(self.sample_rate)) if self.sample_rate else F(np.inf)
                f2 = F(np.tan(freq * nf, dtype=np.float32) / np.tan(freq2 * nf, dtype=np.float32))
            else:
                f2 = F(freq / freq2)
        p.update(nType=int(ntype), nSlope=int(slope), fFreq=freq, fFreq2=f2, fGain=F(gain), fQuality=F(quality))
        return True

    def get_params(self, fid):
        return dict(self.params[fid])

    def coefficients(self, fid, gains):
        """Digital sections of every sample for filter `fid`: [nc][n][5]."""
        p = self.params[fid]
        t = p["nType"]
        T, B = cascades(t, p["nSlope"], p["fFreq2"], p["fQuality"], gains)
        with np.errstate(all="ignore"):                      # kf of :223-228
            if t <= fd.FLT_MT_AMPLIFIER:
                kf = F(0.95)
            elif t & 1:
                kf = F(1.0) / np.tan(p["fFreq"] * PI / F(self.sample_rate), dtype=np.float32)
            else:
                kf = F(TWO_PI / F(self.sample_rate))
            return bilinear(T, B, kf) if (t & 1) else matched(T, B, p["fFreq"], kf)

    def bypassed(self, fid):
        p = self.params[fid] if fid < self.n else None
        return (p is None or not self.active[fid] or p["nType"] == fd.FLT_NONE or p["nSlope"] == 0
                or self.sample_rate == 0)

    def process(self, fid, x, gains, exact=False):
        """DynamicFilters::process (:204-318).  exact=True also returns the float64 run of the same coefficients."""
        x = _f(x)
        if self.bypassed(fid):
            return (x.copy(), x.astype(np.float64)) if exact else x.copy()
        if self.clear_mem:
            self.memory[:] = 0
            self.memory64[:] = 0
            self.clear_mem = False
        coef = self.coefficients(fid, gains)
        nc = coef.shape[0]
        y, st = binding.dyn_biquad_cascade(x, coef, self.memory[fid, :nc])
        self.memory[fid, :nc] = st
        if not exact:
            return y
        y64, st64 = binding.dyn_biquad_cascade_f64(x, coef, self.memory64[fid, :nc])
        self.memory64[fid, :nc] = st64
        return y, y64

    def freq_chart(self, fid, freqs, gain):
        """DynamicFilters::freq_chart (:1774-1873): H(f) of filter `fid` at a fixed gain, complex64."""
        f = _f(freqs)
        if fid >= self.n:
            return None
        p = self.params[fid]
        t = p["nType"]
        if t == fd.FLT_NONE:
            return np.ones(f.size, np.complex64)
        if t in (fd.FLT_BT_AMPLIFIER, fd.FLT_MT_AMPLIFIER):
            return np.full(f.size, F(gain), np.complex64)
        if t & 1:
            nf = F(PI / F(self.sample_rate))
            kf = F(1.0) / np.tan(p["fFreq"] * nf, dtype=np.float32)
            lf = F(self.sample_rate * F(0.499))
            w = (np.tan(np.minimum(f, lf) * nf, dtype=np.float32) * kf).astype(np.float32)
        else:
            w = (f * (F(1.0) / p["fFreq"])).astype(np.float32)
        T, B = cascades(t, p["nSlope"], p["fFreq2"], p["fQuality"], np.array([gain], np.float32))
        h = np.ones(f.size, np.complex64)
        w2 = (w * w).astype(np.float32)
        for J in range(T.shape[0]):                           # dsp::filter_transfer_calc/apply: (t0 - t2 w^2 + j t1 w) / (b0 - b2 w^2 + j b1 w)
            tt, bb = T[J, 0], B[J, 0]
            t_re = (tt[0] - tt[2] * w2).astype(np.float32); t_im = (tt[1] * w).astype(np.float32)
            b_re = (bb[0] - bb[2] * w2).astype(np.float32); b_im = (bb[1] * w).astype(np.float32)
            nrm = (F(1.0) / (b_re * b_re + b_im * b_im)).astype(np.float32)
            re = ((t_re * b_re + t_im * b_im) * nrm).astype(np.float32)
            im = ((t_im * b_re - t_re * b_im) * nrm).astype(np.float32)
            h = (h * (re + 1j * im)).astype(np.complex64)
        return h
