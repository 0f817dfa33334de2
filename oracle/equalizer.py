"""
ORACLE (test infrastructure only).  Restatement of lsp::dspu::Equalizer
(/root/reference/src/main/filters/Equalizer.cpp:67-160 init, :243-358 reconfigure, :460-571 process) on top of
the oracle designer (filter_design.py), the biquad oracle and the FFT primitives of fft_oracle.c, with the
reference's buffers (vInBuffer, vOutBuffer, nBufSize) and block logic.

Pinned by the reference's src/test/utest/filters/equalizer.cpp:35-92 (impulse peak index == get_latency() in FIR,
FFT and SPM modes), replayed in tests/test_oracle_equalizer.py.
"""
import numpy as np

from . import binding as B
from . import filter_design as fd
from . import spectral as sp

BYPASS, IIR, FIR, FFT, SPM = range(5)


def _r2c(x):
    c = np.zeros(2 * x.size, np.float32)
    c[0::2] = x
    return c


class Equalizer:
    def __init__(self, filters, fir_rank):
        self.nf = filters
        self.rank = fir_rank
        self.n = (1 << fir_rank) if fir_rank else 0
        self.params = [fd.Params(fd.FLT_NONE, 1, 1000.0, 1000.0, 1.0, 0.0) for _ in range(filters)]
        self.sr = 0
        self.mode = BYPASS
        self.rebuild = self.clear = True
        self.smooth = False                     # EF_SMOOTH (Equalizer.cpp:618-626)
        self.xfade = False                      # EF_XFADE
        self.latency = 0
        self.bufsize = 0
        # hook: how the bank's impulse response is taken (tests swap in a float64 one to measure the float32 noise)
        self.ir_func = B.biquad_impulse_response
        if self.n:
            self.inb = np.zeros(2 * self.n, np.float32)
            self.outb = np.zeros(2 * self.n, np.float32)
            # vConv / vNewConv start zeroed (Equalizer.cpp:112): a smooth first configuration fades in from silence
            self.conv = B.fastconv_parse(np.zeros(self.n, np.float32), self.rank + 1)
            self.newconv = self.conv.copy()

    def set_smooth(self, smooth):
        self.smooth = bool(smooth)

    def set_mode(self, mode):
        if mode != self.mode:
            self.mode = mode
            self.rebuild = self.clear = True

    def set_sample_rate(self, sr):
        if sr != self.sr:
            self.sr = sr
            self.rebuild = self.clear = True

    def set_params(self, i, p):
        self.params[i] = p.copy()
        self.rebuild = True

    def get_latency(self):
        self._reconfigure()
        return self.latency

    def reset(self):                                            # Equalizer.cpp:573-597
        self.clear = False
        if self.mode == BYPASS:
            return
        if self.mode == IIR:
            if getattr(self, "state", None) is not None:
                self.state[:] = 0
            return
        self.inb[:] = 0; self.outb[:] = 0; self.bufsize = 0

    def _reconfigure(self):
        if not (self.rebuild or self.clear):
            return
        if self.mode == BYPASS:
            self.rebuild = self.clear = self.xfade = False      # Equalizer.cpp:250
            self.latency = 0
            return
        designs = [fd.design(p, self.sr) for p in self.params]
        coef = np.concatenate([d[2] for d in designs]) if designs else np.zeros((0, 5), np.float32)
        if self.clear or getattr(self, "state", None) is None or getattr(self, "coef", np.zeros((0, 5))).shape[0] != coef.shape[0]:
            self.state = np.zeros((max(coef.shape[0], 1), 2), np.float32)       # FilterBank::end(clear) / count change
        self.coef = coef.astype(np.float32)
        if self.mode == IIR:
            self.rebuild = self.clear = self.xfade = False      # Equalizer.cpp:264
            self.latency = 0
            return
        n, half = self.n, self.n >> 1
        if self.clear:
            self.inb[:] = 0; self.outb[:] = 0; self.bufsize = 0
        if self.mode == FIR:
            w2 = sp.window(2 * n, "blackman_nuttall")
            ir = self.ir_func(n, self.coef, self.state)
            tmp = (ir * w2[n:]).astype(np.float32)
            spec = B.packed_direct_fft(_r2c(tmp), self.rank)
            mag = np.sqrt((spec[0::2] * spec[0::2] + spec[1::2] * spec[1::2]).astype(np.float32)).astype(np.float32)
        else:
            fs = half + 1
            f = (np.arange(fs, dtype=np.float32) * np.float32((np.float32(0.5) * np.float32(self.sr)) / np.float32(half))).astype(np.float32)
            mag = np.ones(n, np.float32)
            act = 0
            for p in self.params:
                h, mode = fd.freq_chart(p, self.sr, f)
                if mode == fd.FM_BYPASS:
                    continue
                m = np.abs(h).astype(np.float32)
                mag[:fs] = m if act == 0 else (mag[:fs] * m).astype(np.float32)
                act += 1
            if act > 0:
                mag[fs:fs + half - 1] = mag[1:half][::-1]
            else:
                mag[:] = 1.0
        self.mag = mag
        if self.mode != SPM:
            re = B.packed_reverse_fft(_r2c(mag), self.rank)[0::2]
            h = np.concatenate([re[half:], re[:half]]).astype(np.float32)
            h = (h * sp.window(n, "blackman_nuttall")).astype(np.float32)
            self.fir = h
            if self.smooth:                                     # Equalizer.cpp:339-345
                self.xfade = True
                self.newconv = B.fastconv_parse(h, self.rank + 1)
            else:
                self.conv = B.fastconv_parse(h, self.rank + 1)
            self.latency = n + half
        else:
            self.wnd = sp.window(n, "sqr_cosine")
            self.latency = n
            self.xfade = False                                  # Equalizer.cpp:356
        self.rebuild = self.clear = False

    def process(self, x):
        x = np.asarray(x, np.float32)
        self._reconfigure()
        if self.mode == BYPASS:
            return x.copy()
        if self.mode == IIR:
            y, self.state = B.biquad_cascade(x, self.coef, self.state)
            return y
        out = np.empty_like(x)
        n, half = self.n, self.n >> 1
        pos, left = 0, x.size
        if self.mode in (FIR, FFT):
            while left > 0:
                if self.bufsize >= n:
                    self.outb[:n] = self.outb[n:]
                    self.outb[n:] = 0
                    B.fastconv_parse_apply(self.outb, self.conv, self.inb[:n], self.rank + 1)
                    if self.xfade:                              # Equalizer.cpp:486-501
                        vfft = np.zeros(2 * n, np.float32)
                        self.conv = self.newconv.copy()
                        B.fastconv_parse_apply(vfft, self.conv, self.inb[:n], self.rank + 1)
                        # lramp1(dst, 1, 0, n): dst[i] *= 1 + (0-1)/n * i ; lramp_add2(dst, src, 0, 1, n): dst[i] += src[i] * (i/n)
                        # (lsp-dsp-lib generic semantics: value = v1 + (v2 - v1)/count * i; not pinned by a reference test)
                        i = np.arange(n, dtype=np.float32)
                        delta = np.float32(1.0) / np.float32(n)
                        down = (np.float32(1.0) - delta * i).astype(np.float32)
                        up = (delta * i).astype(np.float32)
                        seg = slice(half, half + n)
                        self.outb[seg] = ((self.outb[seg] * down).astype(np.float32) + (vfft[seg] * up).astype(np.float32)).astype(np.float32)
                        self.outb[n + half:] = vfft[n + half:]
                        self.xfade = False
                    self.bufsize = 0
                k = min(left, n - self.bufsize)
                self.inb[self.bufsize:self.bufsize + k] = x[pos:pos + k]
                out[pos:pos + k] = self.outb[self.bufsize:self.bufsize + k]
                self.bufsize += k; pos += k; left -= k
        else:
            while left > 0:
                if self.bufsize >= half:
                    self.outb[:half] = self.outb[half:n]
                    self.outb[half:n] = 0
                    spec = B.packed_direct_fft(_r2c(self.inb[:n]), self.rank)
                    m = self.mag
                    spec[0::2] *= m; spec[1::2] *= m
                    y = B.packed_reverse_fft(spec, self.rank)[0::2]
                    self.outb[:n] += (y * self.wnd).astype(np.float32)
                    self.inb[:half] = self.inb[half:n]
                    self.bufsize = 0
                k = min(left, half - self.bufsize)
                self.inb[half + self.bufsize:half + self.bufsize + k] = x[pos:pos + k]
                out[pos:pos + k] = self.outb[self.bufsize:self.bufsize + k]
                self.bufsize += k; pos += k; left -= k
        return out
