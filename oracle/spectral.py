"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in the reference's own float32 arithmetic and control flow, of
  windows::*                 /root/reference/src/main/misc/windows.cpp:62-401
  envelope::reverse_noise_lin  src/main/misc/envelope.cpp:40-123
  SpectralProcessor          src/main/util/SpectralProcessor.cpp:59-266
  MultiSpectralProcessor     src/main/util/MultiSpectralProcessor.cpp:288-393
  Analyzer                   src/main/util/Analyzer.cpp:83-152,251-409,443-456
on top of the C primitives of fft_oracle.c.

Pinned by the reference's tests: src/test/utest/util/spectral_proc.cpp:37-67 (identity through the unbound
processor, latency 2^rank, abs 1e-5) replayed in tests/test_oracle_spectral.py; the Analyzer has no reference
test -- its pins are analytic (a sine's peak bin and magnitude).
"""
import math

import numpy as np

from . import binding as B
from .filter_design import F, cosf, expf, logf, sinf

M_PI = math.pi


def _idx(n):
    return [F(i) for i in range(n)]


# ---- windows (misc/windows.cpp) ------------------------------------------------------------------------------
def _cos_sum(n, a):
    """a0 - a1 cos(f i) + a2 cos(2 f i) - a3 cos(3 f i) (+ a4 cos(4 f i)), f = float(2 pi / (n-1))."""
    f1 = F(2.0 * M_PI / (n - 1)) if n > 1 else F(np.inf)
    out = np.empty(n, np.float32)
    fs = [None, f1, F(f1 * F(2.0)), F(f1 * F(3.0)), F(f1 * F(4.0))]
    for i in range(n):
        fi = F(i)
        if len(a) == 2:
            out[i] = F(a[0] - F(a[1] * cosf(F(fi * fs[1]))))
        elif len(a) == 3:       # blackman_general: `a0 - 0.5 * cosf(..) + a2 * cosf(..)` is evaluated in double
            out[i] = F(float(a[0]) - 0.5 * float(cosf(F(fi * fs[1]))) + float(F(a[2] * cosf(F(fi * fs[2])))))
        elif len(a) == 4:
            out[i] = F(F(F(a[0] - F(a[1] * cosf(F(fi * fs[1])))) + F(a[2] * cosf(F(fi * fs[2])))) - F(a[3] * cosf(F(fi * fs[3]))))
    return out


def window(n, name):
    n = int(n)
    if n == 0:
        return np.empty(0, np.float32)
    if name == "hann":
        return _cos_sum(n, [F(0.5), F(0.5)])
    if name == "hamming":
        return _cos_sum(n, [F(0.54), F(0.46)])
    if name == "blackman":
        a2 = F(F(0.16) * F(0.5))
        return _cos_sum(n, [F(F(0.5) - a2), F(0.5), a2])
    if name == "nuttall":
        return _cos_sum(n, [F(0.355768), F(0.487396), F(0.144232), F(0.012604)])
    if name == "blackman_nuttall":
        return _cos_sum(n, [F(0.3635819), F(0.4891775), F(0.1365995), F(0.0106411)])
    if name == "blackman_harris":
        return _cos_sum(n, [F(0.35875), F(0.48829), F(0.14128), F(0.01168)])
    if name == "rectangular":
        return np.ones(n, np.float32)
    if name in ("cosine", "sqr_cosine"):
        f = F(M_PI / n)
        s = np.array([sinf(F(f * F(i))) for i in range(n)], np.float32)
        return s if name == "cosine" else (s * s).astype(np.float32)
    if name in ("triangular", "bartlett_fejer"):
        l = F(n - 1) if name == "bartlett_fejer" else F(n)
        if l == 0.0:
            return np.zeros(1, np.float32)
        l = F(F(2.0) / l)
        c = F((n - 1) * 0.5)
        return np.array([F(F(1.0) - abs(F(F(F(i) - c) * l))) for i in range(n)], np.float32)
    if name == "welch":
        with np.errstate(divide="ignore", invalid="ignore"):       # n == 1: 1 / 0 and 0 * inf, as in the C++
            c = F(F(n - 1) * F(0.5)); mc = F(F(1.0) / c)
            t = [F(F(F(i) - c) * mc) for i in range(n)]
        return np.array([F(F(1.0) - F(x * x)) for x in t], np.float32)
    raise KeyError(name)


WINDOW_IDS = {"hann": 0, "hamming": 1, "blackman": 2, "welch": 8, "nuttall": 9, "blackman_nuttall": 10,
              "blackman_harris": 11, "bartlett_fejer": 14, "triangular": 15, "rectangular": 16, "cosine": 18,
              "sqr_cosine": 19}


def noise_log(first, last, center, n, k):
    """basic_noise_log with colour exponent k (envelope.cpp:152-169): the grid first * exp(i ln(last / first) / (n - 1))."""
    if n <= 1:
        return np.ones(n, np.float32)
    kf = F(F(1.0) / F(center))
    first = F(F(first) * kf); last = F(F(last) * kf)
    df = F(np.log(F(last / first), dtype=np.float32) / F(n - 1))
    d = np.array([F(np.exp(F(df * F(i)), dtype=np.float32) * first) for i in range(n)], np.float32)
    return np.power(d, F(k)).astype(np.float32)


def noise_list(freqs, center, k):
    """basic_noise_list (envelope.cpp:272-280): (freqs / center)^k."""
    d = (np.asarray(freqs, np.float32) * F(F(1.0) / F(center))).astype(np.float32)
    return np.power(d, F(k)).astype(np.float32)


def reverse_noise_lin(first, last, center, n, k):
    """basic_noise_lin with the reversed colour exponent k (envelope.cpp:40-62,95-123)."""
    if n <= 1:
        return np.ones(n, np.float32)
    kf = F(F(1.0) / F(center))
    first = F(F(first) * kf); last = F(F(last) * kf)
    df = F(F(last - first) / F(n - 1))
    d = np.array([F(first + F(df * F(i))) for i in range(n)], np.float32)
    if d[0] <= 0:
        d[0] = d[1]
    return np.power(d, F(k)).astype(np.float32)


# ---- SpectralProcessor ------------------------------------------------------------------------------------------
class SpectralProcessor:
    def __init__(self, max_rank):
        self.rank = self.max_rank = max_rank
        self.phase = F(0.0)
        self.update = True
        self.func = None

    def set_rank(self, rank):
        if rank == self.rank or rank > self.max_rank:
            return
        self.rank = rank
        self.update = True

    def set_phase(self, phase):
        self.phase = F(min(max(phase, 0.0), 1.0))
        self.update = True

    def bind(self, func):
        self.func = func

    def latency(self):
        return 1 << self.rank

    def _apply(self):
        n = 1 << self.rank
        self.wnd = window(n, "cosine")
        self.out_buf = np.zeros(n, np.float32)
        self.in_buf = np.zeros(n, np.float32)
        self.offset = int(F(n) * F(self.phase * F(0.5)))
        self.update = False

    def process(self, src):
        src = np.asarray(src, np.float32)
        if self.update:
            self._apply()
        n = 1 << self.rank
        frame = n >> 1
        dst = np.empty_like(src)
        pos, count = 0, src.size
        while count > 0:
            if self.offset >= frame:
                if self.func is not None:
                    re = (self.in_buf * self.wnd).astype(np.float32)
                    c = np.zeros(2 * n, np.float32); c[0::2] = re
                    spec = B.packed_direct_fft(c, self.rank)
                    spec = np.ascontiguousarray(self.func(spec, self.rank), np.float32)
                    fft = B.packed_reverse_fft(spec, self.rank)[0::2].copy()
                else:
                    fft = (self.in_buf * self.wnd).astype(np.float32)
                self.out_buf[:frame] = self.out_buf[frame:]
                self.out_buf[frame:] = 0.0
                self.out_buf += (fft * self.wnd).astype(np.float32)          # fmadd3
                self.in_buf[:frame] = self.in_buf[frame:]
                self.offset = 0
            todo = min(frame - self.offset, count)
            self.in_buf[frame + self.offset: frame + self.offset + todo] = src[pos:pos + todo]
            dst[pos:pos + todo] = self.out_buf[self.offset:self.offset + todo]
            self.offset += todo; pos += todo; count -= todo
        return dst

    def remaining(self):
        return (1 << (self.rank - 1)) - self.offset

    def reset(self):                                        # SpectralProcessor.cpp:257-266: pOutBuf and pInBuf, not nOffset
        if self.update:
            return
        self.out_buf[:] = 0
        self.in_buf[:] = 0

    def analyze(self, src):
        """process(src, count), SpectralProcessor.cpp:201-249: the function sees every frame, nothing comes back."""
        src = np.asarray(src, np.float32)
        if self.update:
            self._apply()
        n = 1 << self.rank
        frame = n >> 1
        pos, count = 0, src.size
        while count > 0:
            if self.offset >= frame:
                if self.func is not None:
                    re = (self.in_buf * self.wnd).astype(np.float32)
                    c = np.zeros(2 * n, np.float32); c[0::2] = re
                    self.func(B.packed_direct_fft(c, self.rank), self.rank)
                self.out_buf[:frame] = self.out_buf[frame:]
                self.out_buf[frame:] = 0.0
                self.in_buf[:frame] = self.in_buf[frame:]
                self.offset = 0
            todo = min(frame - self.offset, count)
            self.in_buf[frame + self.offset: frame + self.offset + todo] = src[pos:pos + todo]
            self.offset += todo; pos += todo; count -= todo


# ---- Analyzer -----------------------------------------------------------------------------------------------------
R_ALL = frozenset(("envelope", "window", "analysis", "tau", "counters"))
ENV_VIOLET_NOISE, ENV_BLUE_NOISE, ENV_WHITE_NOISE, ENV_PINK_NOISE, ENV_BROWN_NOISE, ENV_MINUS_4_5_DB, ENV_PLUS_4_5_DB = range(7)
_P45 = float(np.float32(4.5 / (20.0 * float(np.float32(math.log10(2.0))))))
# exponent of envelope::reverse_noise_lin (envelope.cpp:95-123): the opposite colour's slope; white is all ones
ENV_REVERSE_SLOPE = {ENV_VIOLET_NOISE: -1.0, ENV_BLUE_NOISE: -0.5, ENV_WHITE_NOISE: None, ENV_PINK_NOISE: 0.5, ENV_BROWN_NOISE: 1.0,
                     ENV_MINUS_4_5_DB: _P45, ENV_PLUS_4_5_DB: -_P45}


class Analyzer:
    """Restated with the reference's staggered schedule (one channel analysed every nStep samples)."""

    def __init__(self, channels, max_rank, max_sr, min_rate, max_delay):
        self.channels, self.rank, self.max_rank = channels, max_rank, max_rank
        fft_items = 1 << max_rank
        bs = fft_items + int(F(max_sr * 2) / F(min_rate)) + max_delay + 0x40
        self.buf_size = (bs + 0x3f) & ~0x3f
        self.buffer = np.zeros((channels, self.buf_size), np.float32)
        csize = (fft_items >> 1) + 1
        self.amp = np.zeros((channels, csize), np.float32)
        self.data = np.zeros((channels, csize), np.float32)
        self.delay = [0] * channels
        self.user_delay = [0] * channels
        self.sample_rate = 0; self.rate = F(1.0); self.reactivity = F(0.0); self.tau = F(1.0)
        self.max_sr = max_sr; self.min_rate = F(int(min_rate))          # fMinRate = uint32_t(min_rate), Analyzer.cpp:120
        self.window_name = "hann"; self.envelope_type = ENV_PINK_NOISE; self.shift = F(1.0)
        self.counter = 0; self.head = 0
        self.flags = set(R_ALL)                         # nReconfigure
        self.active = True                              # Analyzer::set_activity (Analyzer.h)
        self.ch_active = [True] * channels              # enable_channel  (Analyzer.cpp:213-249)
        self.ch_freeze = [False] * channels             # freeze_channel
        self.step = self.period = 0
        self.envelope = np.zeros(csize, np.float32)
        self.wnd = None

    # the setters raise the reference's reconfiguration flags, with its no-change tests (Analyzer.cpp:154-250)
    def configure(self, sample_rate=None, rate=None, rank=None, window_name=None, reactivity=None, shift=None, envelope=None):
        if sample_rate is not None:
            sr = min(int(sample_rate), self.max_sr)
            if sr != self.sample_rate:
                self.sample_rate = sr; self.flags |= R_ALL
        if rate is not None:
            r = F(max(float(self.min_rate), float(F(rate))))
            if r != self.rate:
                self.rate = r; self.flags.add("counters")
        if rank is not None and 2 <= rank <= self.max_rank and rank != self.rank:
            self.rank = rank; self.flags |= R_ALL
        if window_name is not None and window_name != self.window_name:
            self.window_name = window_name; self.flags.add("window")
        if reactivity is not None and F(reactivity) != self.reactivity:
            self.reactivity = F(reactivity); self.flags.add("tau")
        if shift is not None and F(shift) != self.shift:
            self.shift = F(shift); self.flags.add("envelope")
        if envelope is not None and envelope != self.envelope_type:
            self.envelope_type = envelope; self.flags.add("envelope")

    def enable_channel(self, ch, enable):
        if self.ch_active[ch] == bool(enable):
            return False
        self.ch_active[ch] = bool(enable); self.flags.add("counters")
        return True

    def _reconfigure(self):                             # Analyzer.cpp:251-297
        if not self.flags:
            return
        fft_size = 1 << self.rank
        self.csize = (fft_size >> 1) + 1
        period = int(F(self.sample_rate) / self.rate)
        self.step = period // self.channels
        self.period = self.step * self.channels
        if "envelope" in self.flags:
            k = ENV_REVERSE_SLOPE[self.envelope_type]
            env = np.ones(self.csize, np.float32) if k is None else reverse_noise_lin(0.0, F(F(self.sample_rate) * F(0.5)), 100.0, self.csize, k)
            self.envelope = np.zeros(self.amp.shape[1], np.float32)
            self.envelope[:self.csize] = (env * F(self.shift / F(fft_size))).astype(np.float32)
        if "analysis" in self.flags:
            self.amp[:, :self.csize] = 0; self.data[:, :self.csize] = 0
        if "window" in self.flags:
            self.wnd = window(fft_size, self.window_name)
        if "tau" in self.flags:
            with np.errstate(divide="ignore", invalid="ignore"):
                self.tau = F(F(1.0) - expf(F(logf(F(F(1.0) - F(math.sqrt(0.5)))) / F(self.reactivity * self.rate))))
        if "counters" in self.flags:
            self.delay = [i * self.step for i in range(self.channels)]
        self.flags = set()

    def process(self, x):
        """x: [channels][n]."""
        self._reconfigure()
        x = np.asarray(x, np.float32)
        n = x.shape[1]
        fft_size = 1 << self.rank
        off = 0
        while off < n:
            ch, o = divmod(self.counter, self.step)
            if o == 0:
                if self.counter == 0:
                    self.data[:, :self.csize] = self.amp[:, :self.csize]
                if self.ch_freeze[ch]:                  # Analyzer.cpp:334: a frozen channel keeps vAmp
                    pass
                elif not (self.active and self.ch_active[ch]):
                    self.amp[ch, :self.csize] = 0       # Analyzer.cpp:363-364
                doff = self.head - (fft_size + self.delay[ch] + self.user_delay[ch])
                if doff < 0:
                    doff += self.buf_size
                idx = (doff + np.arange(fft_size)) % self.buf_size
                sig = (self.buffer[ch, idx] * self.wnd).astype(np.float32)
                c = np.zeros(2 * fft_size, np.float32); c[0::2] = sig
                spec = B.packed_direct_fft(c, self.rank)
                re, im = spec[0:2 * self.csize:2], spec[1:2 * self.csize:2]
                mod = np.sqrt((re * re + im * im).astype(np.float32)).astype(np.float32)
                if not self.ch_freeze[ch] and self.active and self.ch_active[ch]:
                    self.amp[ch, :self.csize] = (self.amp[ch, :self.csize] * F(F(1.0) - self.tau) + mod * self.tau).astype(np.float32)
            todo = min(n - off, self.step - o)
            idx = (self.head + np.arange(todo)) % self.buf_size
            self.buffer[:, idx] = x[:, off:off + todo]
            off += todo
            self.counter += todo
            if self.counter >= self.period:
                self.counter -= self.period
            self.head = (self.head + todo) % self.buf_size

    def get_spectrum(self, idx):
        idx = np.asarray(idx)
        return (self.data[:, idx] * self.envelope[idx]).astype(np.float32)
