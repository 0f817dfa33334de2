"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in the reference's float32 arithmetic and control flow, of
  SpectralSplitter           /root/reference/src/main/util/SpectralSplitter.cpp:62-361
  crossover::hipass/lopass*  src/main/misc/fft_crossover.cpp:33-400
  FFTCrossover               src/main/util/FFTCrossover.cpp:63-540
on top of the C primitives of fft_oracle.c (packed_direct_fft / packed_reverse_fft) and the window of oracle/spectral.py.

The reference has only manual tests for these (src/test/mtest/util/{spectral_splitter,fft_crossover}.cpp, no expected
values).  Pins used instead (tests/test_oracle_splitter.py): with handlers that pass the spectrum through, every sink
reproduces the input delayed by latency() (sqr_cosine windows at 50 % overlap sum to 1); complementary brick-wall bands of
the manual test's spectral_splitter_func sum to the delayed input; the FFT crossover's masks equal hipass()/lopass()
evaluated bin by bin.
"""
import math

import numpy as np

from . import binding as B
from .filter_design import F, expf, logf
from .spectral import window

BUFFER_MULTIPLIER = 4
XOVER_LEVEL = F(0.5)
SLOPE_SCALE = F((float(F(0.05)) * math.log(10.0)) / math.log(2.0))       # (0.05f * M_LN10) / M_LN2
SLOPE_SCALE_M6 = F((float(F(-0.3)) * math.log(10.0)) / math.log(2.0))


class SpectralSplitter:
    """func(out_spectrum <- in_spectrum (2*2^rank interleaved floats), rank) returns the new spectrum or None;
    sink(samples, first, count)."""

    def __init__(self, max_rank, handlers):
        assert max_rank >= 5
        self.rank = self.max_rank = max_rank
        self.user_chunk_rank = 0; self.chunk_rank = 0
        self.phase = F(0.0)
        bins = 1 << max_rank
        self.inbuf = np.zeros(bins * BUFFER_MULTIPLIER, np.float32)
        self.fft = np.zeros(bins * 2, np.float32)
        self.wnd = np.zeros(bins, np.float32)
        self.h = [dict(func=None, sink=None, out=np.zeros(bins * BUFFER_MULTIPLIER, np.float32)) for _ in range(handlers)]
        self.frame_size = 0; self.in_offset = 0
        self.update = True
        self.bindings = 0

    def bind(self, i, func, sink):
        h = self.h[i]
        if h["func"] is None and h["sink"] is None:
            self.bindings += 1
        h["func"], h["sink"] = func, sink
        h["out"][:(1 << self.rank) * BUFFER_MULTIPLIER] = 0

    def unbind(self, i):
        h = self.h[i]
        if h["func"] is None and h["sink"] is None:
            return False
        h["func"] = h["sink"] = None
        self.bindings -= 1
        return True

    def bound(self, i):
        return self.h[i]["func"] is not None or self.h[i]["sink"] is not None

    def set_phase(self, p):
        self.phase = F(min(max(p, 0.0), 1.0)); self.update = True

    def set_rank(self, r):
        if r == self.rank or r > self.max_rank:
            return
        self.rank = r; self.update = True

    def set_chunk_rank(self, r):
        if r == self.user_chunk_rank:
            return
        self.user_chunk_rank = r; self.update = True

    def latency(self):
        if not self.update:
            return 1 << self.chunk_rank
        rank = min(self.rank, self.max_rank)
        return 1 << (min(max(self.user_chunk_rank, 5), rank) if self.user_chunk_rank > 0 else self.rank)

    def clear(self):
        n = 1 << self.rank
        self.inbuf[:n * BUFFER_MULTIPLIER] = 0
        self.fft[:n * 2] = 0
        for h in self.h:
            if h["sink"] is not None:
                h["out"][:n * BUFFER_MULTIPLIER] = 0

    def update_settings(self):
        if not self.update:
            return
        self.rank = min(self.rank, self.max_rank)
        self.chunk_rank = min(max(self.user_chunk_rank, 5), self.rank) if self.user_chunk_rank > 0 else self.rank
        frame = 1 << (self.chunk_rank - 1)
        self.wnd[:frame * 2] = window(frame * 2, "sqr_cosine")
        self.clear()
        self.frame_size = int(F(F(frame) * F(self.phase * F(0.5))))
        self.in_offset = 0
        self.update = False

    def process(self, src, count):
        self.update_settings()
        if self.bindings <= 0:
            return
        n = 1 << self.rank
        max_buf = n * BUFFER_MULTIPLIER
        frame = 1 << (self.chunk_rank - 1)
        gap = n - frame
        max_in = max_buf - gap
        w = self.wnd[:frame * 2]
        off = 0
        while off < count:
            if self.frame_size >= frame:
                new_off = self.in_offset + frame
                x = self.inbuf[self.in_offset:self.in_offset + n]
                z = np.zeros(2 * n, np.float32); z[0::2] = x
                self.fft = B.packed_direct_fft(z, self.rank)
                for h in self.h:
                    if h["func"] is not None:
                        tmp = np.asarray(h["func"](self.fft.copy(), self.rank), np.float32)
                        tmp = B.packed_reverse_fft(tmp, self.rank)
                        y = tmp[2 * n - 4 * frame::2].copy()             # real parts of the last 2*frame samples
                    else:
                        y = self.inbuf[self.in_offset:self.in_offset + 2 * frame].copy()
                    if h["sink"] is not None:
                        o = h["out"]
                        if new_off >= max_in:
                            o[:frame] = o[new_off:new_off + frame].copy()
                            o[frame:frame + max_in] = 0
                            o[:2 * frame] = (o[:2 * frame] + (y * w).astype(np.float32)).astype(np.float32)
                        else:
                            o[new_off:new_off + 2 * frame] = (o[new_off:new_off + 2 * frame] + (y * w).astype(np.float32)).astype(np.float32)
                if new_off >= max_in:
                    self.inbuf[:gap] = self.inbuf[new_off:new_off + gap].copy()
                    self.in_offset = 0
                else:
                    self.in_offset = new_off
                self.frame_size = 0
            todo = min(frame - self.frame_size, count - off)
            p = self.in_offset + self.frame_size + gap
            if src is not None:
                self.inbuf[p:p + todo] = src[off:off + todo]
            else:
                self.inbuf[p:p + todo] = 0
            for h in self.h:
                if h["sink"] is not None:
                    q = self.in_offset + self.frame_size
                    h["sink"](h["out"][q:q + todo].copy(), off, todo)
            self.frame_size += todo
            off += todo


# ---- misc/fft_crossover.cpp ------------------------------------------------------------------------------------------
def hipass(f, f0, slope):
    f, f0, slope = F(f), F(f0), F(slope)
    if slope > F(-3.0):
        if f <= f0:
            return XOVER_LEVEL
        if f >= F(f0 * F(2.0)):
            return F(1.0)
        return F(expf(F(SLOPE_SCALE_M6 * logf(F(f0 / f)))) * XOVER_LEVEL)
    k = F(slope * SLOPE_SCALE)
    if f >= f0:
        return F(F(1.0) - F(expf(F(k * logf(F(f / f0)))) * XOVER_LEVEL))
    return F(expf(F(k * logf(F(f0 / f)))) * XOVER_LEVEL)


def lopass(f, f0, slope):
    f, f0, slope = F(f), F(f0), F(slope)
    if slope > F(-3.0):
        if f >= f0:
            return XOVER_LEVEL
        if f <= F(f0 * F(0.5)):
            return F(1.0)
        return F(expf(F(SLOPE_SCALE_M6 * logf(F(f / f0)))) * XOVER_LEVEL)
    k = F(slope * SLOPE_SCALE)
    if f >= f0:
        return F(expf(F(k * logf(F(f / f0)))) * XOVER_LEVEL)
    return F(F(1.0) - F(expf(F(k * logf(F(f0 / f)))) * XOVER_LEVEL))


# The *_apply variants leave the gain alone exactly where the *_set variants write 1 (the slope > -3 branches), so
# "gain *= value of the function" restates them.
def _fft_freqs(sample_rate, rank):
    n = 1 << rank
    kf = F(F(sample_rate) / F(n))
    return [None] + [F(F(i) * kf) for i in range(1, n // 2 + 1)] + [F(F(n - i) * kf) for i in range(n // 2 + 1, n)]


def hipass_fft_set(f0, slope, sample_rate, rank):
    fr = _fft_freqs(sample_rate, rank)
    g = np.empty(1 << rank, np.float32)
    g[0] = 0.0
    for i in range(1, 1 << rank):
        g[i] = hipass(fr[i], f0, slope)
    return g


def lopass_fft_set(f0, slope, sample_rate, rank):
    fr = _fft_freqs(sample_rate, rank)
    g = np.empty(1 << rank, np.float32)
    g[0] = 1.0
    for i in range(1, 1 << rank):
        g[i] = lopass(fr[i], f0, slope)
    return g


def hipass_fft_apply(g, f0, slope, sample_rate, rank):
    fr = _fft_freqs(sample_rate, rank)
    g = np.array(g, np.float32)
    g[0] = 0.0
    for i in range(1, 1 << rank):
        g[i] = F(g[i] * hipass(fr[i], f0, slope))
    return g


def lopass_fft_apply(g, f0, slope, sample_rate, rank):
    fr = _fft_freqs(sample_rate, rank)
    g = np.array(g, np.float32)                               # gain[0] *= 1: untouched (fft_crossover.cpp lopass_fft_apply)
    for i in range(1, 1 << rank):
        g[i] = F(g[i] * lopass(fr[i], f0, slope))
    return g


def hipass_set(vf, f0, slope):
    return np.array([hipass(f, f0, slope) for f in vf], np.float32)


def lopass_set(vf, f0, slope):
    return np.array([lopass(f, f0, slope) for f in vf], np.float32)


def hipass_apply(g, vf, f0, slope):
    return np.array([F(F(a) * hipass(f, f0, slope)) for a, f in zip(g, vf)], np.float32)


def lopass_apply(g, vf, f0, slope):
    return np.array([F(F(a) * lopass(f, f0, slope)) for a, f in zip(g, vf)], np.float32)


# ---- FFTCrossover ----------------------------------------------------------------------------------------------------
class FFTCrossover:
    def __init__(self, max_rank, bands):
        self.split = SpectralSplitter(max_rank, bands)
        self.sr = 0
        self.b = [dict(hpf_freq=F(100.0), lpf_freq=F(1000.0), hpf_slope=F(-24.0), lpf_slope=F(-24.0), gain=F(1.0),
                       flatten=F(1.0), lpf=False, hpf=False, enabled=False, update=True, func=None,
                       fft=np.zeros(1 << max_rank, np.float32)) for _ in range(bands)]

    # setters keep the reference's update-flag rules (FFTCrossover.cpp:155-345)
    def set_lpf(self, i, freq, slope, enabled=True):
        b = self.b[i]; freq, slope = F(freq), F(slope)
        if not b["update"]:
            b["update"] = bool(enabled) and (b["lpf_freq"] != freq or b["lpf_slope"] != slope or b["lpf"] != enabled)
        b["lpf_freq"], b["lpf_slope"], b["lpf"] = freq, slope, bool(enabled)

    def set_hpf(self, i, freq, slope, enabled=True):
        b = self.b[i]; freq, slope = F(freq), F(slope)
        if not b["update"]:
            b["update"] = bool(enabled) and (b["hpf_freq"] != freq or b["hpf_slope"] != slope or b["hpf"] != enabled)
        b["hpf_freq"], b["hpf_slope"], b["hpf"] = freq, slope, bool(enabled)

    def set_gain(self, i, g):
        b = self.b[i]
        if b["gain"] != F(g):
            b["gain"] = F(g); b["update"] = True

    def set_flatten(self, i, a):
        b = self.b[i]
        if b["flatten"] != F(a):
            b["flatten"] = F(a); b["update"] = True

    def _sync(self, i):
        b = self.b[i]
        bound = self.split.bound(i)
        if b["enabled"] and b["func"] is not None:
            if not bound:
                self.split.bind(i, lambda spec, rank, b=b: self._spectral(b, spec, rank),
                                lambda s, first, count, b=b, i=i: b["func"](i, s, first, count))
        elif bound:
            self.split.unbind(i)

    def enable_band(self, i, enable=True):
        if self.b[i]["enabled"] != bool(enable):
            self.b[i]["enabled"] = bool(enable); self._sync(i)

    def set_handler(self, i, func):
        self.b[i]["func"] = func; self._sync(i)

    def set_sample_rate(self, sr):
        if sr != self.sr:
            self.sr = sr
            for b in self.b:
                b["update"] = True

    def set_rank(self, rank):
        rank = min(max(rank, 0), self.split.max_rank)
        if rank != self.split.rank:
            self.split.set_rank(rank)
            for b in self.b:
                b["update"] = True

    def set_phase(self, p):
        self.split.set_phase(p)

    def latency(self):
        return self.split.latency()

    def update_band(self, b):
        if not b["update"]:
            return
        rank = self.split.rank
        if b["hpf"]:
            g = hipass_fft_set(b["hpf_freq"], b["hpf_slope"], F(self.sr), rank)
            if b["lpf"]:
                g = lopass_fft_apply(g, b["lpf_freq"], b["lpf_slope"], F(self.sr), rank)
            g = (np.clip(g, F(0.0), b["flatten"]) * b["gain"]).astype(np.float32)
        elif b["lpf"]:
            g = lopass_fft_set(b["lpf_freq"], b["lpf_slope"], F(self.sr), rank)
            g = (np.clip(g, F(0.0), b["flatten"]) * b["gain"]).astype(np.float32)
        else:
            g = np.full(1 << rank, F(b["flatten"] * b["gain"]), np.float32)
        b["fft"] = g
        b["update"] = False

    def _spectral(self, b, spec, rank):
        self.update_band(b)
        out = spec.copy()
        out[0::2] = out[0::2] * b["fft"]; out[1::2] = out[1::2] * b["fft"]      # pcomplex_r2c_mul2
        return out

    def freq_chart(self, i, f):
        b = self.b[i]
        if b["hpf"]:
            m = hipass_set(f, b["hpf_freq"], b["hpf_slope"])
            if b["lpf"]:
                m = lopass_apply(m, f, b["lpf_freq"], b["lpf_slope"])
            return (np.clip(m, F(0.0), b["flatten"]) * b["gain"]).astype(np.float32)
        if b["lpf"]:
            m = lopass_set(f, b["lpf_freq"], b["lpf_slope"])
            return (np.clip(m, F(0.0), b["flatten"]) * b["gain"]).astype(np.float32)
        return np.full(len(f), F(b["flatten"] * b["gain"]), np.float32)

    def process(self, x, count):
        self.split.process(x, count)
