"""Drives every host-only entry of the C-ABI (filter designer, windows, envelopes, FFT-crossover curves) in a library built
with AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py builds it and runs this file in a child
process with the sanitizer runtime preloaded).  Buffers are exactly as long as the header says they must be, so a write or
read past an end is a report; parameters sweep every filter type, slope, the frequency and gain limits, empty and one-point
outputs.   usage: driver.py <library>"""
import ctypes
import sys

import numpy as np

lib = ctypes.CDLL(sys.argv[1])
F, U32, SZ, I = ctypes.c_float, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int
FP = ctypes.POINTER(ctypes.c_float)


class Params(ctypes.Structure):
    _fields_ = [("nType", U32), ("nSlope", U32), ("fFreq", F), ("fFreq2", F), ("fGain", F), ("fQuality", F)]


class X1(ctypes.Structure):
    _fields_ = [(n, F) for n in ("b0", "b1", "b2", "a1", "a2", "p0", "p1", "p2")]


class Cascade(ctypes.Structure):
    _fields_ = [("t", F * 4), ("b", F * 4)]


lib.mi_filter_design.argtypes = [ctypes.POINTER(Params), U32, ctypes.POINTER(X1), U32, ctypes.POINTER(U32),
                                 ctypes.POINTER(Cascade), U32, ctypes.POINTER(U32), ctypes.POINTER(I)]
lib.mi_filter_limit.argtypes = [ctypes.POINTER(Params), U32]
lib.mi_filter_freq_chart.argtypes = [ctypes.POINTER(Params), U32, FP, FP, SZ]
lib.mi_window.argtypes = [FP, SZ, I]
lib.mi_window_general.argtypes = [FP, SZ, I, FP, U32]
for n in ("mi_envelope_noise_lin", "mi_envelope_reverse_noise_lin"):
    getattr(lib, n).argtypes = [FP, F, F, F, SZ, I]
lib.mi_envelope_noise_log.argtypes = [FP, F, F, F, SZ, I, I]
lib.mi_envelope_noise_list.argtypes = [FP, FP, F, SZ, I, I]
for n in ("mi_crossover_hipass", "mi_crossover_lopass"):
    getattr(lib, n).argtypes = [F, F, F]
    getattr(lib, n).restype = F
for n in ("hipass_set", "hipass_apply", "lopass_set", "lopass_apply"):
    f = getattr(lib, "mi_crossover_" + n); f.argtypes = [FP, FP, F, F, SZ]; f.restype = None
    f = getattr(lib, "mi_crossover_" + n.replace("_", "_fft_")); f.argtypes = [FP, F, F, F, SZ]; f.restype = None




libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [SZ]
libc.free.argtypes = [ctypes.c_void_p]
_live = []


def buf(n):
    """exactly n floats from malloc() -- the preloaded sanitizer's malloc, so its red zone starts right behind them
    (Python's own small-object allocator would hand out unguarded arena memory)"""
    p = libc.malloc(4 * max(n, 0))
    _live.append(p)
    if len(_live) > 64:
        libc.free(_live.pop(0))
    return ctypes.cast(p, FP)


def rec(kind, n):
    """n records of a ctypes structure, guarded the same way"""
    p = libc.malloc(ctypes.sizeof(kind) * max(n, 0))
    _live.append(p)
    return ctypes.cast(p, ctypes.POINTER(kind))


if len(sys.argv) > 2 and sys.argv[2] == "--selftest":       # a deliberate overrun: the harness must see a report
    lib.mi_window(buf(8), 64, 1)
    print("selftest: the overrun went unnoticed")
    sys.exit(0)

rng = np.random.default_rng(4)
calls = 0
# ---- the designer: every type, the slope range and beyond, frequencies from 0 to above Nyquist, zero and huge gains ------
for ftype in range(0, 82):                                   # 0 .. 80 are filter_type_t, 81 is out of range
    for slope in (0, 1, 2, 3, 4, 7, 16, 63, 64, 127, 128, 129, 1000):
        for sr in (8000, 44100, 48000, 192000):
            p = Params(ftype, slope, float(rng.choice([0.0, 1.0, 20.0, 997.0, 0.49 * sr, 0.5 * sr, 1e6])),
                       float(rng.choice([0.0, 50.0, 4000.0, 1e6])), float(rng.choice([0.0, 1e-6, 0.25, 1.0, 4.0, 1e6])),
                       float(rng.choice([0.0, 0.1, 0.7071, 10.0, 100.0])))
            nch, ncs, mode = U32(0), U32(0), I(0)
            lib.mi_filter_design(ctypes.byref(p), sr, None, 0, ctypes.byref(nch), None, 0, ctypes.byref(ncs), ctypes.byref(mode))
            assert nch.value <= 128 and ncs.value <= 128, (ftype, slope, nch.value, ncs.value)
            chains, cascades = rec(X1, nch.value), rec(Cascade, ncs.value)          # exactly what the query asked for
            lib.mi_filter_design(ctypes.byref(p), sr, chains, nch.value, ctypes.byref(nch), cascades, ncs.value,
                                 ctypes.byref(ncs), ctypes.byref(mode))
            short = rec(X1, max(nch.value - 1, 0))                                   # one short: must not be overrun
            lib.mi_filter_design(ctypes.byref(p), sr, short, max(nch.value - 1, 0), ctypes.byref(nch), None, 0,
                                 ctypes.byref(ncs), ctypes.byref(mode))
            q = Params(p.nType, p.nSlope, p.fFreq, p.fFreq2, p.fGain, p.fQuality)
            lib.mi_filter_limit(ctypes.byref(q), sr)
            for count in (0, 1, 255, 256, 257):
                f = buf(count)
                for i in range(count):
                    f[i] = 10.0 + i * (0.5 * sr / max(count, 1))
                c = buf(2 * count)
                lib.mi_filter_freq_chart(ctypes.byref(p), sr, c, f, count)
            calls += 8

# ---- windows: all named types and the families, lengths 0, 1, 2, odd, even -------------------------------------------
for wtype in range(-1, 24):
    for n in (0, 1, 2, 3, 16, 17, 4096):
        lib.mi_window(buf(n), n, wtype)
        for count in (0, 1, 2, 3, 4, 5, 6):
            pr = buf(count)
            for i in range(count):
                pr[i] = 0.1 * (i + 1)
            lib.mi_window_general(buf(n), n, wtype, pr, count)
        calls += 8

# ---- envelopes ---------------------------------------------------------------------------------------------------------
for etype in range(-1, 9):
    for n in (0, 1, 2, 257):
        lib.mi_envelope_noise_lin(buf(n), 10.0, 24000.0, 1000.0, n, etype)
        lib.mi_envelope_reverse_noise_lin(buf(n), 0.0, 24000.0, 1000.0, n, etype)
        for rev in (0, 1):
            lib.mi_envelope_noise_log(buf(n), 10.0, 24000.0, 1000.0, n, etype, rev)
            fr = buf(n)
            for i in range(n):
                fr[i] = 10.0 * (i + 1)
            lib.mi_envelope_noise_list(buf(n), fr, 1000.0, n, etype, rev)
        calls += 6

# ---- FFT-crossover curves ----------------------------------------------------------------------------------------------
for slope in (-96.0, -24.0, -3.0, -2.9, 0.0, 6.0):
    for f0 in (0.0, 10.0, 1000.0, 30000.0):
        lib.mi_crossover_hipass(500.0, f0, slope); lib.mi_crossover_lopass(500.0, f0, slope)
        for count in (0, 1, 100):
            fr = buf(count)
            for i in range(count):
                fr[i] = 20.0 * (i + 1)
            for name in ("hipass_set", "hipass_apply", "lopass_set", "lopass_apply"):
                getattr(lib, "mi_crossover_" + name)(buf(count), fr, f0, slope, count)
        for rank in (0, 1, 5, 12):
            for name in ("hipass_fft_set", "hipass_fft_apply", "lopass_fft_set", "lopass_fft_apply"):
                getattr(lib, "mi_crossover_" + name)(buf(1 << rank), f0, slope, 48000.0, rank)
        calls += 30
print("host sanitizer driver: %d calls, no report" % calls)
