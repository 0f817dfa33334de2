"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's filter designer: filter_params_t -> analog
cascades -> digital biquads, i.e. what lsp::dspu::Filter::rebuild() leaves in a
FilterBank.  Restated from /root/reference/src/main/filters/Filter.cpp:

  rebuild dispatch ........ Filter.cpp:208-403
  RLC prototypes .......... Filter.cpp:722-1082
  Butterworth/Chebyshev ... Filter.cpp:1084-1395
  Linkwitz-Riley .......... Filter.cpp:1397-1487
  APO (RBJ) biquads ....... Filter.cpp:1489-1647
  normalize ............... Filter.cpp:1649-1676
  A/B/C/D/K weighting ..... Filter.cpp:1678-2190
  bilinear transform ...... Filter.cpp:2225-2267
  matched Z transform ..... Filter.cpp:2291-2416
  limit ................... Filter.cpp:161-167

The reference mixes float and double arithmetic; every expression here keeps
the C++ evaluation type (np.float32 for `float`, Python float for `double`),
and the float libm entry points (sinf, cosf, ...) are taken from the same glibc
the product's host C++ links, so the two designers agree to the last bit on
this machine.

Parity pin: the reference holds no unit test for coefficients; the anchors are
the ITU-R BS.1770 table quoted in Filter.cpp:2103-2111 (K-weighting @48 kHz),
the sign convention of Filter.cpp:2261-2262 and the analytic frequency-response
identities checked in tests/test_oracle_filters.py.
"""
import ctypes
import ctypes.util
import math

import numpy as np

F = np.float32

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")


def _f1(name):
    fn = getattr(_libm, name)
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_float]
    return lambda x: F(fn(ctypes.c_float(float(x))))


sinf, cosf, tanf, expf, logf, sqrtf, atanf = (
    _f1(n) for n in ("sinf", "cosf", "tanf", "expf", "logf", "sqrtf", "atanf"))

C_PI = F(math.pi)
C_PI_MUL_2 = F(math.pi * 2.0)
C_PI_DIV_2 = F(math.pi / 2.0)

FILTER_CHAINS_MAX = 0x80
MIN_APO_Q = F(0.1)

# filters/common.h:38-135
FILTER_TYPES = """FLT_NONE FLT_BT_AMPLIFIER FLT_MT_AMPLIFIER
FLT_BT_RLC_LOPASS FLT_MT_RLC_LOPASS FLT_BT_RLC_HIPASS FLT_MT_RLC_HIPASS
FLT_BT_RLC_LOSHELF FLT_MT_RLC_LOSHELF FLT_BT_RLC_HISHELF FLT_MT_RLC_HISHELF
FLT_BT_RLC_BELL FLT_MT_RLC_BELL FLT_BT_RLC_RESONANCE FLT_MT_RLC_RESONANCE
FLT_BT_RLC_NOTCH FLT_MT_RLC_NOTCH FLT_BT_RLC_ALLPASS FLT_MT_RLC_ALLPASS
FLT_BT_RLC_ALLPASS2 FLT_MT_RLC_ALLPASS2 FLT_BT_RLC_LADDERPASS FLT_MT_RLC_LADDERPASS
FLT_BT_RLC_LADDERREJ FLT_MT_RLC_LADDERREJ FLT_BT_RLC_BANDPASS FLT_MT_RLC_BANDPASS
FLT_BT_RLC_ENVELOPE FLT_MT_RLC_ENVELOPE
FLT_BT_BWC_LOPASS FLT_MT_BWC_LOPASS FLT_BT_BWC_HIPASS FLT_MT_BWC_HIPASS
FLT_BT_BWC_LOSHELF FLT_MT_BWC_LOSHELF FLT_BT_BWC_HISHELF FLT_MT_BWC_HISHELF
FLT_BT_BWC_BELL FLT_MT_BWC_BELL FLT_BT_BWC_LADDERPASS FLT_MT_BWC_LADDERPASS
FLT_BT_BWC_LADDERREJ FLT_MT_BWC_LADDERREJ FLT_BT_BWC_BANDPASS FLT_MT_BWC_BANDPASS
FLT_BT_BWC_ALLPASS FLT_MT_BWC_ALLPASS
FLT_BT_LRX_LOPASS FLT_MT_LRX_LOPASS FLT_BT_LRX_HIPASS FLT_MT_LRX_HIPASS
FLT_BT_LRX_LOSHELF FLT_MT_LRX_LOSHELF FLT_BT_LRX_HISHELF FLT_MT_LRX_HISHELF
FLT_BT_LRX_BELL FLT_MT_LRX_BELL FLT_BT_LRX_LADDERPASS FLT_MT_LRX_LADDERPASS
FLT_BT_LRX_LADDERREJ FLT_MT_LRX_LADDERREJ FLT_BT_LRX_BANDPASS FLT_MT_LRX_BANDPASS
FLT_BT_LRX_ALLPASS FLT_MT_LRX_ALLPASS
FLT_DR_APO_LOPASS FLT_DR_APO_HIPASS FLT_DR_APO_BANDPASS FLT_DR_APO_NOTCH
FLT_DR_APO_ALLPASS FLT_DR_APO_ALLPASS2 FLT_DR_APO_PEAKING FLT_DR_APO_LOSHELF
FLT_DR_APO_HISHELF FLT_DR_APO_LADDERPASS FLT_DR_APO_LADDERREJ
FLT_A_WEIGHTED FLT_B_WEIGHTED FLT_C_WEIGHTED FLT_D_WEIGHTED FLT_K_WEIGHTED""".split()
T = {n: i for i, n in enumerate(FILTER_TYPES)}
globals().update(T)

FM_BYPASS, FM_BILINEAR, FM_MATCHED, FM_APO = range(4)


class Params:
    """filter_params_t (filters/common.h:137-145)."""

    def __init__(self, ftype, slope=1, freq=1000.0, freq2=1000.0, gain=1.0, quality=0.0):
        self.nType = int(ftype)
        self.nSlope = int(slope)
        self.fFreq = F(freq)
        self.fFreq2 = F(freq2)
        self.fGain = F(gain)
        self.fQuality = F(quality)

    def copy(self):
        return Params(self.nType, self.nSlope, self.fFreq, self.fFreq2, self.fGain, self.fQuality)


def D(x):
    return float(x)


class _Designer:
    def __init__(self, params, sample_rate):
        self.sr = int(sample_rate)
        self.p = params.copy()
        # Filter::limit (Filter.cpp:161-167)
        max_freq = F(F(0.49) * F(self.sr))
        self.p.nSlope = min(max(self.p.nSlope, 1), FILTER_CHAINS_MAX)
        self.p.fFreq = min(max(self.p.fFreq, F(0.0)), max_freq)
        self.p.fFreq2 = min(max(self.p.fFreq2, F(0.0)), max_freq)
        self.mode = FM_BYPASS
        self.cascades = []   # analog (or plot) cascades: (t[3], b[3]) float32
        self.biquads = []    # digital sections (b0,b1,b2,a1,a2), a* sign-negated

    # -- helpers -----------------------------------------------------------
    def _cascade(self):
        # Filter::add_cascade (Filter.cpp:177-197): beyond the limit the last slot is reused
        c = {"t": [F(0)] * 3, "b": [F(0)] * 3}
        if len(self.cascades) >= FILTER_CHAINS_MAX:
            self.cascades[-1] = c
        else:
            self.cascades.append(c)
        return c

    def _chain(self, b0, b1, b2, a1, a2):
        q = [F(b0), F(b1), F(b2), F(a1), F(a2)]
        self.biquads.append(q)
        return q

    def _plot_cascade(self, q):
        c = self._cascade()
        c["t"] = [q[0], q[1], q[2]]
        c["b"] = [F(1.0), F(-q[3]), F(-q[4])]

    def _bilinear_relative(self, f1, f2):
        nf = F(C_PI / F(self.sr))
        with np.errstate(divide="ignore", invalid="ignore"):       # f2 == 0 divides by zero exactly as the C++ does
            return F(tanf(F(f1 * nf)) / tanf(F(f2 * nf)))

    # -- dispatch (Filter.cpp:208-403) --------------------------------------
    def run(self):
        t = self.p.nType
        fp = self.p.copy()
        name = FILTER_TYPES[t] if 0 <= t < len(FILTER_TYPES) else "FLT_NONE"
        if name.startswith(("FLT_BT_", "FLT_MT_")):
            matched = name.startswith("FLT_MT_")
            base = t - 1 if matched else t
            if matched:
                with np.errstate(divide="ignore", invalid="ignore"):    # fFreq2 == 0: inf, as the float division in C
                    fp.fFreq2 = F(fp.fFreq / fp.fFreq2)
            else:
                fp.fFreq2 = self._bilinear_relative(fp.fFreq, fp.fFreq2)
            bname = FILTER_TYPES[base]
            if "_RLC_" in bname or bname == "FLT_BT_AMPLIFIER":
                self.mode = FM_BILINEAR          # calc_rlc_filter sets it first (Filter.cpp:725)
                self._rlc(base, fp)
            elif "_BWC_" in bname:
                self._bwc(base, fp)
            else:
                self._lrx(base, fp)
            # the dispatcher overrides the mode after the calc_* call (Filter.cpp:238-239 ...)
            self.mode = FM_MATCHED if matched else FM_BILINEAR
        elif name in ("FLT_DR_APO_LOPASS", "FLT_DR_APO_HIPASS", "FLT_DR_APO_BANDPASS", "FLT_DR_APO_NOTCH",
                      "FLT_DR_APO_ALLPASS", "FLT_DR_APO_PEAKING", "FLT_DR_APO_LOSHELF", "FLT_DR_APO_HISHELF"):
            self._apo(t, fp)
            self.mode = FM_APO
        elif name == "FLT_DR_APO_ALLPASS2":
            self._apo(FLT_DR_APO_ALLPASS, fp)
            fp.fFreq = self.p.fFreq2
            fp.fGain = F(1.0)
            self._apo(FLT_DR_APO_ALLPASS, fp)
            self.mode = FM_APO
        elif name == "FLT_DR_APO_LADDERPASS":
            self._apo(FLT_DR_APO_HISHELF, fp)
            fp.fFreq = self.p.fFreq2
            fp.fGain = F(F(1.0) / self.p.fGain)
            self._apo(FLT_DR_APO_HISHELF, fp)
            self.mode = FM_APO
        elif name == "FLT_DR_APO_LADDERREJ":
            self._apo(FLT_DR_APO_LOSHELF, fp)
            fp.fFreq = self.p.fFreq2
            self._apo(FLT_DR_APO_HISHELF, fp)
            self.mode = FM_APO
        elif name.endswith("_WEIGHTED"):
            self._weighted(t)
        else:
            self.mode = FM_BYPASS

        if self.mode == FM_BILINEAR:
            self._bilinear()
        elif self.mode == FM_MATCHED:
            self._matched()
        return self

    # -- RLC prototypes (Filter.cpp:722-1082) --------------------------------
    def _rlc(self, type_, fp):
        q = fp.fQuality
        g = fp.fGain
        n = fp.nSlope
        if type_ == FLT_BT_AMPLIFIER:
            c = self._cascade()
            c["t"][0] = g
            c["b"][0] = F(1.0)
        elif type_ in (FLT_BT_RLC_LOPASS, FLT_BT_RLC_HIPASS):
            lo = type_ == FLT_BT_RLC_LOPASS
            k = F(2.0 / (1.0 + D(q)))
            i = n & 1
            if i:
                c = self._cascade()
                c["b"][0] = F(1.0)
                c["b"][1] = F(1.0)
                c["t"][0 if lo else 1] = g
            for j in range(i, n, 2):
                c = self._cascade()
                c["b"] = [F(1.0), k, F(1.0)]
                c["t"][0 if lo else 2] = g if j == 0 else F(1.0)
        elif type_ in (FLT_BT_RLC_LOSHELF, FLT_BT_RLC_HISHELF):
            slope = n * 2
            gain = sqrtf(g)
            fg = expf(F(logf(gain) / F(slope)))
            for j in range(n):
                c = self._cascade()
                t = [fg, F(2.0 / (1.0 + D(q))), F(1.0 / D(fg))]
                b = [F(1.0 / D(fg)), F(2.0 / (1.0 + D(q))), fg]
                if type_ == FLT_BT_RLC_LOSHELF:
                    c["t"], c["b"] = t, b
                else:
                    c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain) for v in c["t"]]
        elif type_ in (FLT_BT_RLC_LADDERPASS, FLT_BT_RLC_LADDERREJ):
            rej = type_ == FLT_BT_RLC_LADDERREJ
            slope = n * 2
            gain1 = sqrtf(F(1.0 / D(g))) if rej else sqrtf(g)
            gain2 = sqrtf(g) if rej else sqrtf(F(1.0 / D(g)))
            fg1 = expf(F(logf(gain1) / F(slope)))
            fg2 = expf(F(logf(gain2) / F(slope)))
            kf = fp.fFreq2
            for j in range(n):
                c = self._cascade()
                fg = fg2 if rej else fg1
                gain = gain2 if rej else gain1
                t = [fg, F(2.0 / (1.0 + D(q))), F(1.0 / D(fg))]
                b = [F(1.0 / D(fg)), F(2.0 / (1.0 + D(q))), fg]
                if rej:
                    c["t"], c["b"] = t, b
                else:
                    c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain) for v in c["t"]]
                # second shelf, always hi-shelf
                c = self._cascade()
                t = [fg2, F(2.0 * D(kf) / (1.0 + D(q))), F(F(kf * kf) / fg2)]
                b = [F(1.0 / D(fg2)), F(2.0 * D(kf) / (1.0 + D(q))), F(F(fg2 * kf) * kf)]
                c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain2) for v in c["t"]]
        elif type_ == FLT_BT_RLC_BANDPASS:
            kf = fp.fFreq2
            kf2 = F(kf * kf)
            k = F(F(2.0) / F(F(1.0) + q))
            i = n & 1
            if i:
                c = self._cascade()
                c["t"][1] = F(g * g)
                c["b"] = [F(1.0), F(F(1.0) + kf), kf]
            for j in range(i, n, 2):
                c = self._cascade()
                c["b"] = [F(1.0), k, F(1.0)]
                c["t"][0] = g if j == 0 else F(1.0)
                c = self._cascade()
                c["b"] = [F(1.0), F(k * kf), kf2]
                c["t"][2] = g if j == 0 else F(1.0)
        elif type_ == FLT_BT_RLC_BELL:
            fg = expf(F(logf(g) / F(n)))
            angle = atanf(fg)
            k = F(2.0 * (1.0 / D(fg) + D(fg)) / (1.0 + (2.0 * D(q)) / n))
            kt = F(k * sinf(angle))
            kb = F(k * cosf(angle))
            for j in range(n):
                c = self._cascade()
                c["t"] = [F(1.0), kt, F(1.0)]
                c["b"] = [F(1.0), kb, F(1.0)]
        elif type_ == FLT_BT_RLC_RESONANCE:
            angle = atanf(expf(F(logf(g) / F(n))))
            k = F(2.0 / (1.0 + D(q)))
            kt = F(k * sinf(angle))
            kb = F(k * cosf(angle))
            for j in range(n):
                c = self._cascade()
                c["t"] = [F(1.0), kt, F(1.0)]
                c["b"] = [F(1.0), kb, F(1.0)]
        elif type_ == FLT_BT_RLC_NOTCH:
            c = self._cascade()
            c["t"] = [g, F(0), g]
            c["b"] = [F(1.0), F(2.0 / (1.0 + D(q))), F(1.0)]
        elif type_ == FLT_BT_RLC_ALLPASS:
            k = F(F(2.0) / F(F(1.0) + q))
            c = None
            for j in range(n):
                c = self._cascade()
                c["t"] = [F(1.0), F(-k), F(1.0)]
                c["b"] = [F(1.0), k, F(1.0)]
            if c is not None:
                c["t"] = [F(v * g) for v in c["t"]]
        elif type_ == FLT_BT_RLC_ALLPASS2:
            kf = fp.fFreq2
            kfp1 = F(1.0 + D(kf))
            c = None
            for j in range(n):
                c = self._cascade()
                c["t"] = [F(1.0), F(-kfp1), kf]
                c["b"] = [F(1.0), kfp1, kf]
            if c is not None:
                c["t"] = [F(v * g) for v in c["t"]]
        elif type_ == FLT_BT_RLC_ENVELOPE:
            slope = n
            cj = 0
            if slope & 1:
                k = F(1.0)
                for _ in range(3):
                    c = self._cascade()
                    kk = F(k * k)
                    c["t"] = [F(1.0), F(F(F(1.0) + F(0.25)) * k), F(F(F(0.25) * k) * k)]
                    c["b"] = [F(1.0), F(F(F(0.5) + F(0.125)) * k), F(F(F(F(0.5) * F(0.125)) * k) * k)]
                    k = F(k * F(0.0625))
                    if cj == 0:
                        c["t"] = [F(v * g) for v in c["t"]]
                    cj += 1
            slope >>= 1
            for j in range(slope):
                c = self._cascade()
                c["t"][0] = g if cj == 0 else F(1.0)
                c["t"][1] = g if cj == 0 else F(1.0)
                c["b"][0] = F(1.0)
                c["b"][1] = F(0.0005)
                cj += 1
        else:
            self.mode = FM_BYPASS

    # -- Butterworth-Chebyshev (Filter.cpp:1084-1395) ------------------------
    @staticmethod
    def _pole(theta, k):
        tsin = sinf(theta)
        tcos = sqrtf(F(1.0 - D(F(tsin * tsin))))
        kf = F(F(tsin * tsin) + F(F(F(k * k) * tcos) * tcos))
        return tsin, tcos, kf

    def _bwc(self, type_, fp):
        q = fp.fQuality
        g = fp.fGain
        n = fp.nSlope
        if type_ in (FLT_BT_BWC_LOPASS, FLT_BT_BWC_HIPASS):
            lo = type_ == FLT_BT_BWC_LOPASS
            k = F(F(1.0) / F(F(1.0) + q))
            i = n & 1
            if i:
                c = self._cascade()
                c["b"][0] = F(1.0)
                c["b"][1] = F(1.0)
                c["t"][0 if lo else 1] = g
            for j in range(i, n, 2):
                theta = F(F(F(j - i + 1) * C_PI_DIV_2) / F(n))
                tsin, tcos, kf = self._pole(theta, k)
                c = self._cascade()
                if not lo:
                    c["t"][2] = g if j == 0 else F(1.0)
                    c["b"] = [F(1.0 / D(kf)), F(2.0 * D(k) * D(tcos) / D(kf)), F(1.0)]
                else:
                    c["t"][0] = g if j == 0 else F(1.0)
                    c["b"] = [F(1.0), F(2.0 * D(k) * D(tcos) / D(kf)), F(1.0 / D(kf))]
        elif type_ == FLT_BT_BWC_ALLPASS:
            k = F(F(1.0) / F(F(1.0) + q))
            i = n & 1
            if i:
                c = self._cascade()
                c["t"] = [F(-g), g, F(0.0)]
                c["b"] = [F(1.0), F(1.0), F(0.0)]
            for j in range(i, n, 2):
                theta = F(F(F(j - i + 1) * C_PI_DIV_2) / F(n))
                tsin, tcos, kf = self._pole(theta, k)
                c = self._cascade()
                c["t"] = [F(1.0), F(-2.0 * D(tcos)), F(1.0)]
                c["b"] = [F(1.0 / D(kf)), F(2.0 * D(k) * D(tcos) / D(kf)), F(1.0)]
                if j == 0:
                    c["t"] = [F(v * g) for v in c["t"]]
        elif type_ in (FLT_BT_BWC_HISHELF, FLT_BT_BWC_LOSHELF):
            gain = sqrtf(g)
            fg = expf(F(D(logf(gain)) / (2.0 * n)))
            k = F(D(F(1.0)) / (1.0 + D(q) * (1.0 - D(expf(F(2.0 - D(gain) - 1.0 / D(gain)))))))
            for j in range(n):
                theta = F(F(F(2 * j + 1) * C_PI_DIV_2) / F(2 * n))
                tsin, tcos, kf = self._pole(theta, k)
                c = self._cascade()
                t = [F(kf / fg), F(2.0 * D(k) * D(tcos)), fg]
                b = [fg, F(2.0 * D(k) * D(tcos)), F(kf / fg)]
                if type_ == FLT_BT_BWC_HISHELF:
                    c["t"], c["b"] = t, b
                else:
                    c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain) for v in c["t"]]
        elif type_ in (FLT_BT_BWC_LADDERPASS, FLT_BT_BWC_LADDERREJ):
            lp = type_ == FLT_BT_BWC_LADDERPASS
            slope = n * 2
            gain1 = sqrtf(g) if lp else sqrtf(F(1.0 / D(g)))
            gain2 = sqrtf(F(1.0 / D(g))) if lp else sqrtf(g)
            fg1 = expf(F(D(logf(gain1)) / (2.0 * n)))
            fg2 = expf(F(D(logf(gain2)) / (2.0 * n)))

            def kq(gn):
                e = expf(F(F(F(2.0) - gn) - F(F(1.0) / gn)))
                return F(F(1.0) / F(F(1.0) + F(q * F(F(1.0) - e))))
            k1 = kq(gain1)
            k2 = kq(gain2)
            xf = fp.fFreq2
            xf2 = F(xf * xf)
            for j in range(n):
                theta = F(F(F(2 * j + 1) * C_PI_DIV_2) / F(slope))
                tsin = sinf(theta)
                tcos = sqrtf(F(F(1.0) - F(tsin * tsin)))
                k = k1 if lp else k2
                fg = fg1 if lp else fg2
                gain = gain1 if lp else gain2
                kf = F(F(tsin * tsin) + F(F(F(k * k) * tcos) * tcos))
                c = self._cascade()
                t0 = F(kf / fg)
                t1 = F(F(F(2.0) * k) * tcos)
                t = [t0, t1, fg]
                b = [fg, t1, t0]
                if lp:
                    c["t"], c["b"] = t, b
                else:
                    c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain) for v in c["t"]]
                # second shelf, always hi-shelf
                kf = F(F(tsin * tsin) + F(F(F(k1 * k1) * tcos) * tcos))
                c = self._cascade()
                t0 = F(kf / fg1)
                t1 = F(F(F(F(2.0) * k1) * xf) * tcos)
                t = [t0, t1, F(fg1 * xf2)]
                b = [fg1, t1, F(t0 * xf2)]
                c["b"], c["t"] = t, b
                if j == 0:
                    c["t"] = [F(v * gain2) for v in c["t"]]
        elif type_ == FLT_BT_BWC_BELL:
            fg = expf(F(logf(g) / F(2 * n)))
            k = F(D(F(1.0)) / (1.0 + D(q)))
            for j in range(n):
                theta = F(F(F(2 * j + 1) * C_PI_DIV_2) / F(2 * n))
                tsin, tcos, kf = self._pole(theta, k)
                two_k_tcos = F(F(F(2.0) * k) * tcos)
                if D(g) >= 1.0:
                    c = self._cascade()
                    c["t"] = [F(1.0), F(F(two_k_tcos * fg) / kf), F(F(F(F(1.0) * fg) * fg) / kf)]
                    c["b"] = [F(1.0), F(two_k_tcos / kf), F(F(1.0) / kf)]
                    c = self._cascade()
                    c["t"] = [F(1.0), F(two_k_tcos / fg), F(kf / F(fg * fg))]
                    c["b"] = [F(1.0), two_k_tcos, kf]
                else:
                    c = self._cascade()
                    c["t"] = [F(1.0), F(two_k_tcos / kf), F(F(1.0) / kf)]
                    c["b"] = [F(1.0), F(two_k_tcos / F(fg * kf)), F(F(1.0) / F(F(fg * fg) * kf))]
                    c = self._cascade()
                    c["t"] = [F(1.0), two_k_tcos, kf]
                    c["b"] = [F(1.0), F(two_k_tcos * fg), F(F(kf * fg) * fg)]
        elif type_ == FLT_BT_BWC_BANDPASS:
            f2 = fp.fFreq2
            k = F(F(1.0) / F(F(1.0) + q))
            for j in range(n):
                theta = F(F(F(2 * j + 1) * C_PI_DIV_2) / F(2 * n))
                tsin, tcos, kf = self._pole(theta, k)
                two_k_tcos = F(F(F(2.0) * k) * tcos)
                c = self._cascade()
                c["t"][2] = g if j == 0 else F(1.0)
                c["b"] = [F(F(1.0) / kf), F(two_k_tcos / kf), F(1.0)]
                c = self._cascade()
                c["t"][0] = F(1.0)
                c["b"] = [F(1.0), F(F(two_k_tcos * f2) / kf), F(F(f2 * f2) / kf)]
        else:
            self.mode = FM_BYPASS

    # -- Linkwitz-Riley = two Butterworth passes (Filter.cpp:1397-1487) -------
    def _lrx(self, type_, fp):
        remap = {
            FLT_BT_LRX_LOPASS: FLT_BT_BWC_LOPASS, FLT_BT_LRX_HIPASS: FLT_BT_BWC_HIPASS,
            FLT_BT_LRX_LOSHELF: FLT_BT_BWC_LOSHELF, FLT_BT_LRX_HISHELF: FLT_BT_BWC_HISHELF,
            FLT_BT_LRX_BELL: FLT_BT_BWC_BELL, FLT_BT_LRX_BANDPASS: FLT_BT_BWC_BANDPASS,
            FLT_BT_LRX_LADDERPASS: FLT_BT_BWC_LADDERPASS, FLT_BT_LRX_LADDERREJ: FLT_BT_BWC_LADDERREJ,
        }
        if type_ == FLT_BT_LRX_ALLPASS:
            k = F(F(1.0) / F(F(1.0) + fp.fQuality))
            i = self.p.nSlope * 2
            for j in range(0, i, 2):
                theta = F(F(F(j + 1) * C_PI_DIV_2) / F(i))
                tsin = sinf(theta)
                tcos = sqrtf(F(F(1.0) - F(tsin * tsin)))
                kf = F(F(tsin * tsin) + F(F(F(k * k) * tcos) * tcos))
                c1 = self._cascade()
                c2 = self._cascade()
                xeta = F(F(F(F(j) + F(0.5)) * C_PI) / F(i))
                c1["t"] = [F(1.0), F(F(-2.0) * cosf(xeta)), F(1.0)]
                xeta = F(F(F(F(j) + F(1.5)) * C_PI) / F(i))
                c2["t"] = [F(1.0), F(F(-2.0) * cosf(xeta)), F(1.0)]
                c1["b"] = [F(F(1.0) / kf), F(F(F(F(2.0) * k) * tcos) / kf), F(1.0)]
                c2["b"] = list(c1["b"])
                if j == 0:
                    c1["t"] = [F(v * fp.fGain) for v in c1["t"]]
            return
        if type_ not in remap:
            self.mode = FM_BYPASS
            return
        bfp = fp.copy()
        bfp.nSlope = self.p.nSlope * 2
        bfp.fGain = sqrtf(bfp.fGain)
        self._bwc(remap[type_], bfp)
        self._bwc(remap[type_], bfp)

    # -- APO / RBJ direct digital design (Filter.cpp:1489-1647) ---------------
    def _apo(self, type_, fp):
        omega = F(F(C_PI_MUL_2 * fp.fFreq) / F(self.sr))
        cs = sinf(omega)
        cc = cosf(omega)
        Q = fp.fQuality if fp.fQuality > MIN_APO_Q else MIN_APO_Q
        alpha = F(F(F(0.5) * cs) / Q)
        one = F(1.0)
        two = F(2.0)
        if type_ == FLT_DR_APO_LOPASS:
            A = fp.fGain
            a0 = F(F(A * F(0.5)) * F(one - cc)); a1 = F(A * F(one - cc)); a2 = a0
            b0 = F(one + alpha); b1 = F(F(-2.0) * cc); b2 = F(one - alpha)
        elif type_ == FLT_DR_APO_HIPASS:
            A = fp.fGain
            a0 = F(F(A * F(0.5)) * F(one + cc)); a1 = F(A * F(F(-1.0) - cc)); a2 = a0
            b0 = F(one + alpha); b1 = F(F(-2.0) * cc); b2 = F(one - alpha)
        elif type_ == FLT_DR_APO_BANDPASS:
            A = fp.fGain
            a0 = F(A * alpha); a1 = F(0.0); a2 = F(A * F(-alpha))
            b0 = F(one + alpha); b1 = F(F(-2.0) * cc); b2 = F(one - alpha)
        elif type_ == FLT_DR_APO_NOTCH:
            A = fp.fGain
            a0 = A; a1 = F(F(A * F(-2.0)) * cc); a2 = a0
            b0 = F(one + alpha); b1 = F(F(-2.0) * cc); b2 = F(one - alpha)
        elif type_ == FLT_DR_APO_ALLPASS:
            A = fp.fGain
            a0 = F(A * F(one - alpha)); a1 = F(F(A * F(-2.0)) * cc); a2 = F(A * F(one + alpha))
            b0 = a2; b1 = a1; b2 = a0
        elif type_ == FLT_DR_APO_PEAKING:
            A = sqrtf(fp.fGain)
            a0 = F(one + F(alpha * A)); a1 = F(F(-2.0) * cc); a2 = F(one - F(alpha * A))
            b0 = F(one + F(alpha / A)); b1 = a1; b2 = F(one - F(alpha / A))
        elif type_ == FLT_DR_APO_LOSHELF:
            A = sqrtf(fp.fGain)
            beta = F(F(two * alpha) * sqrtf(A))
            ap1 = F(A + one); am1 = F(A - one)
            a0 = F(A * F(F(ap1 - F(am1 * cc)) + beta))
            a1 = F(F(two * A) * F(am1 - F(ap1 * cc)))
            a2 = F(A * F(F(ap1 - F(am1 * cc)) - beta))
            b0 = F(F(ap1 + F(am1 * cc)) + beta)
            b1 = F(F(-2.0) * F(am1 + F(ap1 * cc)))
            b2 = F(F(ap1 + F(am1 * cc)) - beta)
        elif type_ == FLT_DR_APO_HISHELF:
            A = sqrtf(fp.fGain)
            beta = F(2.0 * D(alpha) * D(sqrtf(A)))
            ap1 = F(A + one); am1 = F(A - one)
            a0 = F(A * F(F(ap1 + F(am1 * cc)) + beta))
            a1 = F(F(F(-2.0) * A) * F(am1 + F(ap1 * cc)))
            a2 = F(A * F(F(ap1 + F(am1 * cc)) - beta))
            b0 = F(F(ap1 - F(am1 * cc)) + beta)
            b1 = F(two * F(am1 - F(ap1 * cc)))
            b2 = F(F(ap1 - F(am1 * cc)) - beta)
        else:
            return
        rb0 = F(one / b0)
        qd = self._chain(F(a0 * rb0), F(a1 * rb0), F(a2 * rb0), F(F(-b1) * rb0), F(F(-b2) * rb0))
        self._plot_cascade(qd)

    # -- normalisation of a digital section (Filter.cpp:1649-1676) ------------
    def _normalize(self, q, frequency, gain):
        fr = min(F(frequency), F(F(self.sr) * F(0.5)))
        xf = F(F(C_PI_MUL_2 * fr) / F(self.sr))
        cw = cosf(xf); sw = sinf(xf)
        c2w = F(F(cw * cw) - F(sw * sw))
        s2w = F(F(F(2.0) * sw) * cw)
        b0, b1, b2, a1, a2 = q
        alpha = F(F(b0 + F(b1 * cw)) + F(b2 * c2w))
        beta = F(F(b1 * sw) + F(b2 * s2w))
        gamma = F(F(F(1.0) - F(a1 * cw)) - F(a2 * c2w))
        delta = F(F(F(-a1) * sw) - F(a2 * s2w))
        mag = F(F(gamma * gamma) + F(delta * delta))
        w_re = F(F(alpha * gamma) - F(beta * delta))
        w_im = F(F(alpha * delta) + F(beta * gamma))
        egain = F(F(F(gain) * mag) / sqrtf(F(F(w_re * w_re) + F(w_im * w_im))))
        q[0] = F(b0 * egain); q[1] = F(b1 * egain); q[2] = F(b2 * egain)

    # -- weighting filters (Filter.cpp:1678-2190) ----------------------------
    def _weighted(self, type_):
        Tp = F(F(1.0) / F(self.sr))
        one = F(1.0)

        def pole_pair_hp(p0):            # zeros 0,0 ; double pole at -p0
            ww = F(F(p0) * Tp); ws = sinf(ww); wc = cosf(ww)
            ka0 = F(one / F(one + ws))
            b0 = F(F(F(0.5) * F(one + wc)) * ka0)
            q = self._chain(b0, F(F(F(-1.0) - wc) * ka0), b0,
                            F(F(F(2.0) * wc) * ka0), F(F(ws - one) * ka0))
            self._normalize(q, 1000.0, 1.0); self._plot_cascade(q)

        def pole_pair_lp(p0):            # no zeros ; double pole at -p0
            ww = F(F(p0) * Tp); ws = sinf(ww); wc = cosf(ww)
            ka0 = F(one / F(one + ws))
            b0 = F(F(F(0.5) * F(one - wc)) * ka0)
            q = self._chain(b0, F(F(one - wc) * ka0), b0,
                            F(F(F(-2.0) * wc) * ka0), F(F(one - ws) * ka0))
            self._normalize(q, 1000.0, 1.0); self._plot_cascade(q)

        def two_real_poles(p0, p1, kind):
            ww0 = F(F(p0) * Tp); ww1 = F(F(p1) * Tp)
            ws0 = sinf(ww0); wc0 = cosf(ww0); ws1 = sinf(ww1); wc1 = cosf(ww1)
            kx0 = F(one / F(F(one + ws0) - wc0)); kx1 = F(one / F(F(one + ws1) - wc1))
            ka0 = F(kx0 * kx1)
            ky0 = F(F(one - wc0) - ws0); ky1 = F(F(one - wc1) - ws1)
            a1 = F(-F(F(ky0 * kx0) + F(ky1 * kx1)))
            a2 = F(F(F(-ky0) * ky1) * ka0)
            if kind == "A":               # zeros 0,0
                b0 = F(F(ws0 * ws1) * ka0)
                q = self._chain(b0, F(F(-2.0) * b0), b0, a1, a2)
            else:                         # "D": one zero at 0
                b0 = F(F(ws0 * F(one - wc1)) * ka0)
                q = self._chain(b0, F(0.0), F(-b0), a1, a2)
            self._normalize(q, 1000.0, 1.0); self._plot_cascade(q)

        if type_ == FLT_A_WEIGHTED:
            pole_pair_hp(129.4); two_real_poles(676.7, 4636.0, "A"); pole_pair_lp(76655.0)
            self.mode = FM_APO
        elif type_ == FLT_B_WEIGHTED:
            pole_pair_hp(129.4)
            ww = F(F(995.9) * Tp); ws = sinf(ww); wc = cosf(ww)
            ka0 = F(one / F(F(one + ws) - wc))
            b0 = F(ws * ka0)
            q = self._chain(b0, F(-b0), F(0.0), F(F(F(ws + wc) - one) * ka0), F(0.0))
            self._normalize(q, 1000.0, 1.0); self._plot_cascade(q)
            pole_pair_lp(76655.0)
            self.mode = FM_APO
        elif type_ == FLT_C_WEIGHTED:
            pole_pair_hp(129.4); pole_pair_lp(76655.0)
            self.mode = FM_APO
        elif type_ == FLT_D_WEIGHTED:
            two_real_poles(1776.3, 7288.5, "D")
            p0, p1, r0, r1 = F(6401.17), F(19706.85), F(1.02), F(1.092)
            ww0 = F(F(p0 * Tp) * F(0.5)); ww1 = F(F(p1 * Tp) * F(0.5))
            wt0 = F(one / tanf(ww0)); wt1 = F(one / tanf(ww1))
            ka0 = F(one / F(one + F(wt1 * F(wt1 + r1))))
            q = self._chain(F(F(one + F(wt0 * F(wt0 + r0))) * ka0),
                            F(F(F(2.0) * F(one - F(wt0 * wt0))) * ka0),
                            F(F(one + F(wt0 * F(wt0 - r0))) * ka0),
                            F(F(F(-2.0) * F(one - F(wt1 * wt1))) * ka0),
                            F(F(-F(one + F(wt1 * F(wt1 - r1)))) * ka0))
            self._normalize(q, 1000.0, 1.0); self._plot_cascade(q)
            self.mode = FM_APO
        elif type_ == FLT_K_WEIGHTED:
            Vh = F(1.58486470113); Vb = F(1.25872093023)
            f0 = F(1681.974450955533); Q = F(0.7071752369554196)
            K = tanf(F(F(C_PI * f0) * Tp)); K2 = F(K * K); KQ = F(K / Q)
            ka0 = F(one / F(F(one + KQ) + K2))
            q = self._chain(F(F(F(Vh + F(Vb * KQ)) + K2) * ka0),
                            F(F(F(2.0) * F(K2 - Vh)) * ka0),
                            F(F(F(Vh - F(Vb * KQ)) + K2) * ka0),
                            F(F(F(-2.0) * F(K2 - one)) * ka0),
                            F(F(-F(F(one - KQ) + K2)) * ka0))
            self._plot_cascade(q)
            f0 = F(38.13547087602444); Q = F(0.5003270373238773)
            K = tanf(F(F(C_PI * f0) * Tp)); K2 = F(K * K); KQ = F(K / Q)
            ka0 = F(one / F(F(one + KQ) + K2))
            q = self._chain(F(1.0), F(-2.0), F(1.0),
                            F(F(F(-2.0) * F(K2 - one)) * ka0),
                            F(F(-F(F(one - KQ) + K2)) * ka0))
            self._plot_cascade(q)
            self.mode = FM_APO

    # -- analog -> digital (Filter.cpp:2225-2267) ----------------------------
    def _bilinear(self):
        kf = D(F(F(1.0) / tanf(F(F(self.p.fFreq * C_PI) / F(self.sr)))))
        kf2 = kf * kf
        for idx, c in enumerate(self.cascades):
            if idx >= FILTER_CHAINS_MAX:
                break
            t, b = c["t"], c["b"]
            T0, T1, T2 = D(t[0]), D(t[1]) * kf, D(t[2]) * kf2
            B0, B1, B2 = D(b[0]), D(b[1]) * kf, D(b[2]) * kf2
            N = 1.0 / (B0 + B1 + B2)
            self._chain((T0 + T1 + T2) * N, 2.0 * (T0 - T2) * N, (T0 - T1 + T2) * N,
                        2.0 * (B2 - B0) * N, (B1 - B2 - B0) * N)

    # -- matched Z transform (Filter.cpp:2291-2416) --------------------------
    def _matched(self):
        f = self.p.fFreq
        TD = F(C_PI_MUL_2 / F(self.sr))
        for idx, c in enumerate(self.cascades):
            if idx >= FILTER_CHAINS_MAX:
                break
            PP = []
            AI = []
            for p in (c["t"], c["b"]):
                P = [F(0)] * 3
                if p[2] == 0.0:
                    if p[1] == 0.0:
                        P[0] = p[0]
                    else:
                        k = F(p[1] / f)
                        R = F(F(-p[0]) / k)
                        P[0] = k
                        P[1] = F(F(-k) * expf(F(R * TD)))
                else:
                    k = p[2]
                    a = F(F(1.0) / F(f * f))
                    b = F(p[1] / F(f * p[2]))
                    cc = F(p[0] / p[2])
                    Dd = F(F(b * b) - F(F(F(4.0) * a) * cc))
                    if Dd >= 0:
                        Dd = sqrtf(Dd)
                        R0 = F(F(F(-b) - Dd) / F(F(2.0) * a))
                        R1 = F(F(F(-b) + Dd) / F(F(2.0) * a))
                        P[0] = k
                        P[1] = F(F(-k) * F(expf(F(R0 * TD)) + expf(F(R1 * TD))))
                        P[2] = F(k * expf(F(F(R0 + R1) * TD)))
                    else:
                        Dd = sqrtf(F(-Dd))
                        R = F(F(-b) / F(F(2.0) * a))
                        K = F(Dd / F(F(2.0) * a))
                        P[0] = k
                        P[1] = F(F(F(F(-2.0) * k) * expf(F(R * TD))) * cosf(F(K * TD)))
                        P[2] = F(k * expf(F(F(F(2.0) * R) * TD)))
                # amplitude of the discrete part at f/10 and of the analog part at 0.1
                w = D(F(F(F(C_PI * F(0.2)) * self.p.fFreq) / F(self.sr)))
                re = D(P[0]) * math.cos(2.0 * w) + D(P[1]) * math.cos(w) + D(P[2])
                im = D(P[0]) * math.sin(2.0 * w) + D(P[1]) * math.sin(w)
                A = F(math.sqrt(re * re + im * im))
                w = 0.1
                re = D(p[0]) - D(p[2]) * w * w
                im = D(p[1]) * w
                I = F(math.sqrt(re * re + im * im))
                PP.append(P)
                AI.append((A, I))
            Tt, Bb = PP
            (A0, I0), (A1, I1) = AI
            AN = D(F(F(A1 * I0) / F(A0 * I1)))            # float division, then widened (Filter.cpp:2395)
            N = 1.0 / D(Bb[0])
            self._chain(D(Tt[0]) * N * AN, D(Tt[1]) * N * AN, D(Tt[2]) * N * AN,
                        -D(Bb[1]) * N, -D(Bb[2]) * N)


def design(params, sample_rate):
    """Returns (mode, cascades, biquads): biquads is an (n,5) float32 array of
    (b0,b1,b2,a1,a2) with a1,a2 sign-negated as FilterBank::add_chain receives them."""
    d = _Designer(params, sample_rate).run()
    bq = np.array(d.biquads, dtype=np.float32).reshape(-1, 5)
    return d.mode, d.cascades, bq


def freq_response(biquads, freqs, sample_rate):
    """H(e^{jw}) of a digital cascade in float64 (reference convention: y = b.x + a.y)."""
    w = 2.0 * np.pi * np.asarray(freqs, dtype=np.float64) / float(sample_rate)
    z1 = np.exp(-1j * w)
    z2 = z1 * z1
    h = np.ones_like(z1)
    for b0, b1, b2, a1, a2 in np.asarray(biquads, dtype=np.float64):
        h = h * (b0 + b1 * z1 + b2 * z2) / (1.0 - a1 * z1 - a2 * z2)
    return h


def freq_chart(params, sample_rate, freqs):
    """Filter::freq_chart(c, f, count) (Filter.cpp:602-696): complex H at the given frequencies, float32 math.
    The analog cascade response is lsp-dsp-lib's filter_transfer_calc/apply_pc:
    H = (t0 - t2 w^2 + j t1 w) / (b0 - b2 w^2 + j b1 w) at the normalised frequency w."""
    d = _Designer(params, sample_rate).run()
    f = np.asarray(freqs, dtype=np.float32)
    mode = d.mode if d.cascades else FM_BYPASS
    sr = F(d.sr)
    h = np.ones(f.size, np.complex64)
    if mode in (FM_BILINEAR, FM_MATCHED):
        if mode == FM_BILINEAR:
            nf = F(C_PI / sr)
            kf = F(F(1.0) / tanf(F(d.p.fFreq * nf)))
            lf = F(F(d.sr) * F(0.499))
            w = np.array([F(tanf(F(min(x, lf) * nf)) * kf) for x in f], np.float32)
        else:
            w = (f * F(F(1.0) / d.p.fFreq)).astype(np.float32)
        w2 = (w * w).astype(np.float32)
        for c in d.cascades:
            t, b = c["t"], c["b"]
            t_re = (t[0] - t[2] * w2).astype(np.float32); t_im = (t[1] * w).astype(np.float32)
            b_re = (b[0] - b[2] * w2).astype(np.float32); b_im = (b[1] * w).astype(np.float32)
            n = (F(1.0) / (b_re * b_re + b_im * b_im)).astype(np.float32)
            re = ((t_re * b_re + t_im * b_im) * n).astype(np.float32)
            im = ((t_im * b_re - t_re * b_im) * n).astype(np.float32)
            h = (h * (re + 1j * im)).astype(np.complex64)
    elif mode == FM_APO:
        kf = F(C_PI_MUL_2 / sr); lf = F(F(d.sr) * F(0.5))
        a = [F(min(x, lf) * kf) for x in f]
        cw = np.array([cosf(x) for x in a], np.float32); sw = np.array([sinf(x) for x in a], np.float32)
        c2w = (cw * cw - sw * sw).astype(np.float32); s2w = (F(2.0) * sw * cw).astype(np.float32)
        for c in d.cascades:
            t, b = c["t"], c["b"]
            alpha = (t[0] + t[1] * cw + t[2] * c2w).astype(np.float32); beta = (t[1] * sw + t[2] * s2w).astype(np.float32)
            gamma = (b[0] + b[1] * cw + b[2] * c2w).astype(np.float32); delta = (b[1] * sw + b[2] * s2w).astype(np.float32)
            mag = (F(1.0) / (gamma * gamma + delta * delta)).astype(np.float32)
            re = (mag * (alpha * gamma - beta * delta)).astype(np.float32)
            im = (mag * (alpha * delta + beta * gamma)).astype(np.float32)
            h = (h * (re + 1j * im)).astype(np.complex64)
    return h, mode
