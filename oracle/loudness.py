"""
ORACLE (test infrastructure only).  Restatement of lsp::dspu::LoudnessMeter
(/root/reference/src/main/meters/LoudnessMeter.cpp:85-185 init, :297-321 set_sample_rate, :328-379 update_settings,
:381-407 refresh_rms, :409-466 process_channels, :468-560 process) with the reference's own chunking (BUFFER_SIZE 0x400),
running-sum update order and refresh schedule, float32 throughout.

The reference has no unit test for it.  Pinned instead by the standard it implements: a 0 dBFS 997 Hz sine reads
-3.01 LKFS (ITU-R BS.1770-4, and the -0.691 dB of misc/broadcast.h:96), see tests/test_oracle_loudness.py.
lsp-dsp-lib primitives restated: sqr2 (x*x), h_sum (sequential float sum), ssqrt1 (sqrt of the non-negative part),
mul_k3 / fmadd_k3 / mix_copy2 (one multiply-add per element).
"""
import numpy as np

from . import binding as B
from . import filter_design as fd

F = np.float32
BUFFER_SIZE = 0x400
WEIGHT_NONE, WEIGHT_A, WEIGHT_B, WEIGHT_C, WEIGHT_D, WEIGHT_K = range(6)
CHANNEL_NONE, CHANNEL_CENTER, CHANNEL_LEFT, CHANNEL_RIGHT = 0, 1, 4, 5
CHANNEL_LFE1, CHANNEL_LFE2 = 32, 33
_TYPES = [fd.FLT_NONE, fd.FLT_A_WEIGHTED, fd.FLT_B_WEIGHTED, fd.FLT_C_WEIGHTED, fd.FLT_D_WEIGHTED, fd.FLT_K_WEIGHTED]


def channel_weighting(designation):                          # misc/broadcast.cpp:32-55
    if 6 <= designation <= 11:
        return F(1.41)
    if designation in (CHANNEL_LFE1, CHANNEL_LFE2):
        return F(0.0)
    return F(1.0)


def _hsum(v):
    s = F(0.0)
    for x in np.asarray(v, np.float32):
        s = F(s + x)
    return s


class LoudnessMeter:
    def __init__(self, channels, max_period=400.0):
        self.nch = channels
        self.ch = [dict(weight=F(0.0), link=F(1.0), desig=CHANNEL_NONE, enabled=True, bound=True, ms=F(0.0), data=None,
                        coef=None, state=None) for _ in range(channels)]
        if channels == 1:
            self.ch[0]["desig"] = CHANNEL_CENTER
        elif channels == 2:
            self.ch[0]["desig"], self.ch[1]["desig"] = CHANNEL_LEFT, CHANNEL_RIGHT
        for c in self.ch:
            if c["desig"] != CHANNEL_NONE:
                c["weight"] = channel_weighting(c["desig"])
        self.period_ms = F(min(max_period, 400.0)); self.max_period = F(max_period)
        self.avg = F(1.0); self.loud = F(0.0)
        self.period = 0; self.refresh = 0; self.sr = 0
        self.weighting = WEIGHT_K
        self.upd_filters = self.upd_time = True
        self.head = 0; self.size = 0

    def set_designation(self, i, d):
        self.ch[i]["desig"] = d; self.ch[i]["weight"] = channel_weighting(d)

    def set_link(self, i, link):
        self.ch[i]["link"] = F(min(max(link, 0.0), 1.0))

    def set_active(self, i, active=True):
        c = self.ch[i]
        if c["enabled"] == bool(active):
            return
        c["enabled"] = bool(active)
        if active and c["data"] is not None:
            c["data"][:] = 0; c["ms"] = F(0.0)

    def set_bound(self, i, bound=True):                       # bind(id, out, in) with in == NULL or not
        self.ch[i]["bound"] = bool(bound)

    def set_weighting(self, w):
        if w != self.weighting:
            self.weighting = w; self.upd_filters = True

    def set_period(self, ms):
        ms = F(min(max(ms, 0.0), float(self.max_period)))
        if ms != self.period_ms:
            self.period_ms = ms; self.upd_time = True

    def clear(self):
        self.loud = F(0.0)
        for c in self.ch:
            if c["state"] is not None:
                c["state"][:] = 0
            if c["enabled"] and c["data"] is not None:
                c["data"][:] = 0; c["ms"] = F(0.0)

    def set_sample_rate(self, sr):
        if sr == self.sr:
            return
        n = int(F(F(self.max_period * F(0.001)) * F(sr))) + BUFFER_SIZE
        size = 1
        while size < n:
            size <<= 1
        for c in self.ch:
            c["data"] = np.zeros(size, np.float32)
        self.sr, self.size, self.head = sr, size, 0
        self.upd_filters = self.upd_time = True
        self.clear()

    def latency(self):
        return int(F(F(self.period_ms * F(0.001)) * F(self.sr)))

    def _update(self):
        if self.upd_time:
            self.period = max(int(F(F(self.period_ms * F(0.001)) * F(self.sr))), 1)
            self.avg = F(F(1.0) / F(self.period)); self.refresh = 0
            self.upd_time = False
        if self.upd_filters:
            coef = fd.design(fd.Params(_TYPES[self.weighting], 0, 0.0, 0.0, 1.0, 0.0), self.sr)[2]
            coef = np.asarray(coef, np.float32).reshape(-1, 5)[:4]              # sBank.init(4)
            for c in self.ch:
                c["coef"] = coef
                c["state"] = np.zeros((max(coef.shape[0], 1), 2), np.float32)   # sBank.end(true)
            self.upd_filters = False

    def _refresh(self):                                      # LoudnessMeter.cpp:381-407
        if self.refresh > 0:
            return
        tail = (self.head + self.size - self.period) & (self.size - 1)
        for c in self.ch:
            if not c["enabled"]:
                continue
            if tail < self.head:
                c["ms"] = _hsum(c["data"][tail:self.head])
            else:
                c["ms"] = F(_hsum(c["data"][:self.head]) + _hsum(c["data"][tail:]))
        self.refresh = max(BUFFER_SIZE << 2, self.period >> 2)

    def process(self, x, gain=None):
        """x: [channels][n] -> (loudness[n], per-channel outputs [channels][n])."""
        self._update()
        x = np.asarray(x, np.float32)
        n = x.shape[1]
        out = np.zeros(n, np.float32)
        cho = np.zeros((self.nch, n), np.float32)
        mask = self.size - 1
        g = F(1.0) if gain is None else F(gain)
        off = 0
        while off < n:
            self._refresh()
            todo = min(n - off, self.refresh, BUFFER_SIZE)
            buf = np.zeros(todo, np.float32)
            vms = {}
            mixed = 0
            for i, c in enumerate(self.ch):
                if not c["enabled"] or not c["bound"]:           # vIn == NULL: left out of the block (:421-422)
                    continue
                y, c["state"] = B.biquad_cascade(x[i, off:off + todo], c["coef"], c["state"])
                idx = (self.head + np.arange(todo)) & mask
                c["data"][idx] = (y * y).astype(np.float32)
                tail = (self.head + self.size - self.period) & mask
                tidx = (tail + np.arange(todo)) & mask
                ms = c["ms"]
                v = np.empty(todo, np.float32)
                for j in range(todo):                        # ms += new - old, sample by sample (:447-453)
                    ms = F(ms + F(c["data"][idx[j]] - c["data"][tidx[j]]))
                    v[j] = F(self.avg * ms)
                c["ms"] = ms
                vms[i] = v
                buf = (v * c["weight"]).astype(np.float32) if mixed == 0 else (buf + (v * c["weight"]).astype(np.float32)).astype(np.float32)
                mixed += 1
            buf = np.sqrt(np.maximum(buf, F(0.0))).astype(np.float32)
            out[off:off + todo] = (buf * g).astype(np.float32) if gain is not None else buf
            if gain is None:                                    # only process(out, count) records fLoudness (:485)
                self.loud = buf[-1]
            for i, c in enumerate(self.ch):
                if not c["enabled"] or not c["bound"]:           # (an unbound channel's output is a stale buffer in the reference)
                    continue
                r = np.sqrt(np.maximum(vms[i], F(0.0))).astype(np.float32)
                if c["link"] <= 0:
                    o = r * g
                elif c["link"] >= 1:
                    o = buf * g
                else:
                    o = (buf * F(c["link"] * g)).astype(np.float32) + (r * F(F(F(1.0) - c["link"]) * g)).astype(np.float32)
                cho[i, off:off + todo] = o.astype(np.float32)
            self.head = (self.head + todo) & mask
            self.refresh -= todo
            off += todo
        return out, cho
