"""
ORACLE (test infrastructure only).  Restatement of lsp::dspu::ILUFSMeter
(/root/reference/src/main/meters/ILUFSMeter.cpp:113-211 init, :264-289 set_integration_period, :291-322 set_sample_rate,
:324-353 gated / infinite loudness, :355-470 process, :472-520 update_settings, :522-560 clear), float32 throughout, with
the reference's chunking (BUFFER_SIZE 0x400) and these behaviours of the reference kept as they are:
  * compute_gated_loudness(threshold) compares with GATING_ABS_THRESH whatever `threshold` is (:333), so the relative
    (-10 LU) stage returns the value of the absolute stage;
  * update_settings() ends with nFlags = 0 (:519) and F_BLK_FULL is one of nFlags' bits (ILUFSMeter.h:62-66): each
    process() call starts with the flag cleared, and gating blocks are evaluated only after the quarter counter wraps
    within the same call.

The reference has no unit test for it (src/test/mtest/meters/ilufs.cpp is a manual test without expected values).  Pinned
instead by the standard: a 0 dBFS 997 Hz sine integrates to -3.01 LKFS (ITU-R BS.1770-4), tests/test_oracle_ilufs.py.
lsp-dsp-lib primitives restated: h_sqr_sum (sequential float sum of x*x), fill, mul_k2.
"""
import numpy as np

from . import binding as B
from . import filter_design as fd
from .loudness import (CHANNEL_CENTER, CHANNEL_LEFT, CHANNEL_NONE, CHANNEL_RIGHT, WEIGHT_K, _TYPES,  # noqa: F401
                       channel_weighting)

F = np.float32
BUFFER_SIZE = 0x400
GATING_ABS_THRESH = F(1.17246530458e-07)                     # ILUFSMeter.cpp:39
GATING_REL_THRESH = F(0.1)
MIN_GATING_BLOCKS = 64
DBFS_TO_LUFS_SHIFT_GAIN = F(0.923527857225)                  # misc/broadcast.h: -0.691 dB


def _h_sqr_sum(v):
    v = np.asarray(v, np.float32)
    return np.cumsum(v * v, dtype=np.float32)[-1] if v.size else F(0.0)    # sequential float32 accumulation


class ILUFSMeter:
    def __init__(self, channels, max_int_time=60.0, block_period=400.0):
        self.nch = channels
        self.ch = [dict(weight=F(0.0), enabled=True, block=np.zeros(4, np.float32), coef=None, state=None)
                   for _ in range(channels)]
        if channels == 1:
            self.ch[0]["weight"] = channel_weighting(CHANNEL_CENTER)
        elif channels == 2:
            self.ch[0]["weight"] = channel_weighting(CHANNEL_LEFT)
            self.ch[1]["weight"] = channel_weighting(CHANNEL_RIGHT)
        self.block_period = F(block_period)
        self.int_time = F(max_int_time); self.max_int_time = F(max_int_time)
        self.avg = F(1.0); self.loud = F(0.0)
        self.block_size = 0; self.block_offset = 0; self.block_part = 0
        self.ms_size = 0; self.ms_head = 0; self.ms_int = 0; self.ms_count = 0
        self.sr = 0
        self.upd_filters = self.upd_time = True; self.blk_full = False
        self.weighting = WEIGHT_K
        self.hist = None
        # diagnostic for the tests, not part of the restated state: how close (relative) any gating block evaluated since
        # the last clear() came to the absolute gate -- a block the gate DROPS leaves no other trace when the mean runs on
        self.gate_margin = float("inf")
        self.call_gate_margin = float("inf")                 # the same over the blocks evaluated by the last process() call

    def set_designation(self, i, d):
        self.ch[i]["weight"] = channel_weighting(d)

    def set_active(self, i, active=True):
        self.ch[i]["enabled"] = bool(active)

    def set_weighting(self, w):
        if w != self.weighting:
            self.weighting = w; self.upd_filters = True

    def _clear_blocks(self):
        for c in self.ch:
            c["block"][:] = 0
        if self.hist is not None:
            self.hist[:] = 0
        self.blk_full = False

    def set_integration_period(self, period):
        lo = F(self.block_period * F(0.001))
        period = F(period)
        period = lo if period < lo else (self.max_int_time if period > self.max_int_time else period)
        if period == self.int_time:
            return
        if self.int_time <= 0:
            self.ms_count = 0
            self._clear_blocks()
        elif period <= 0:
            self._clear_blocks()
        self.int_time = period; self.upd_time = True

    def clear(self):
        for c in self.ch:
            if c["state"] is not None:
                c["state"][:] = 0
        self._clear_blocks()
        self.loud = F(0.0)
        self.block_offset = self.block_part = 0
        self.ms_head = self.ms_count = 0
        self.gate_margin = float("inf")

    def _blk(self):
        return int(F(F(F(self.block_period * F(0.25)) * F(0.001)) * F(self.sr)))

    def set_sample_rate(self, sr):
        if sr == self.sr:
            return
        self.sr = sr
        blk = self._blk()
        int_count = (int(F(self.max_int_time * F(sr))) + blk - 1) // blk
        blocks = (max(int_count, MIN_GATING_BLOCKS) + 3) & ~3
        self.hist = np.zeros(blocks, np.float32)
        self.avg = F(F(0.25) / F(blk))
        self.block_size = blk; self.ms_size = blocks
        self.upd_filters = self.upd_time = True; self.blk_full = False
        self.clear()

    def _update(self):
        if self.upd_time:
            t = min(self.int_time, self.max_int_time)
            blk = self._blk()
            if t > 0:
                v = ((int(F(t * F(self.sr))) - blk * 2 - 1) % (1 << 64)) // blk      # size_t arithmetic
                self.ms_int = max(v, 1) & 0xffffffff
            else:
                self.ms_int = 0
            self.ms_count = min(self.ms_count, self.ms_int)
            self.upd_time = False
        if self.upd_filters:
            coef = fd.design(fd.Params(_TYPES[self.weighting], 0, 0.0, 0.0, 1.0, 0.0), self.sr)[2]
            coef = np.asarray(coef, np.float32).reshape(-1, 5)[:4]
            for c in self.ch:
                c["coef"] = coef
                c["state"] = np.zeros((max(coef.shape[0], 1), 2), np.float32)
            self.upd_filters = False
        self.blk_full = False                                # nFlags = 0, ILUFSMeter.cpp:519

    def _gated(self):
        s = F(0.0); blocks = 0
        tail = (self.ms_head + self.ms_size - self.ms_count) % self.ms_size
        for _ in range(self.ms_count):
            lj = self.hist[tail]
            tail = (tail + 1) % self.ms_size
            if lj <= GATING_ABS_THRESH:
                continue
            blocks += 1
            s = F(s + lj)
        return F(s / F(blocks)) if blocks else F(0.0)

    def _infinite(self):
        mult = F(1.0 / float(F(self.ms_count)))
        s = F(0.0)
        for lj in self.hist:
            s = F(s + F(mult * lj))
        return s

    def process(self, x, gain=DBFS_TO_LUFS_SHIFT_GAIN):
        """x: [channels][n] -> integrated loudness per sample [n] (as a gain)."""
        # update_settings() returns early when no flag is set; F_BLK_FULL counts as a flag and is wiped with the rest
        self._update()
        self.call_gate_margin = float("inf")
        x = np.asarray(x, np.float32)
        n = x.shape[1]
        out = np.zeros(n, np.float32)
        g = F(gain)
        off = 0
        while off < n:
            todo = min(n - off, self.block_size - self.block_offset, BUFFER_SIZE)
            if todo > 0:
                for i, c in enumerate(self.ch):
                    if not c["enabled"]:
                        continue
                    y, c["state"] = B.biquad_cascade(x[i, off:off + todo], c["coef"], c["state"])
                    c["block"][self.block_part] = F(c["block"][self.block_part] + _h_sqr_sum(y))
                self.block_offset += todo
            out[off:off + todo] = F(self.loud * g)
            if self.block_offset >= self.block_size:
                self.block_offset = 0
                self.block_part += 1
                if self.block_part >= 4:
                    self.block_part = 0
                    self.blk_full = True
                if self.blk_full:
                    loud = F(0.0)
                    for c in self.ch:
                        b = c["block"]
                        s = F(F(F(F(b[0] + b[1]) + b[2]) + b[3]) * self.avg)
                        loud = F(loud + F(c["weight"] * s))
                    margin = abs(float(loud) - float(GATING_ABS_THRESH)) / float(GATING_ABS_THRESH)
                    self.gate_margin = min(self.gate_margin, margin)
                    self.call_gate_margin = min(self.call_gate_margin, margin)
                    if self.ms_int > 0:
                        self.ms_count = min(self.ms_count + 1, self.ms_int)
                        self.hist[self.ms_head] = loud
                        self.ms_head = (self.ms_head + 1) % self.ms_size
                        loud = self._gated()
                        if F(loud * GATING_REL_THRESH) > GATING_ABS_THRESH:
                            loud = self._gated()
                    else:
                        if loud > GATING_ABS_THRESH:
                            if self.ms_count >= 0x100:
                                self.hist *= F(0.5)
                                self.ms_count >>= 1
                            self.ms_count += 1
                            self.hist[self.ms_head] = F(self.hist[self.ms_head] + loud)
                            self.ms_head = (self.ms_head + 1) % self.ms_size
                        loud = self._infinite() if self.ms_count > 0 else F(0.0)
                    self.loud = F(np.sqrt(loud))
                for c in self.ch:
                    c["block"][self.block_part] = 0
            off += todo
        return out
