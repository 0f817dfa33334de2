"""
ORACLE (test infrastructure only).  Restatement of lsp::dspu::Crossover
(/root/reference/src/main/util/Crossover.cpp:71-160 init, :162-198 select_filter/select_slope, :342-449 reconfigure,
:451-498 process, :500-590 freq_chart) on top of the oracle designer (filter_design.py) and the biquad oracle.

A split point owns an Equalizer in IIR mode (low-pass + the all-pass filters of the split points above it; all
sections in one bank) and a Filter (high-pass).  The reference has no unit test for Crossover: parity unpinned by
reference vectors; the Linkwitz-Riley property (the bands sum to an all-pass) is checked in tests/test_oracle_crossover.py.
"""
import math

import numpy as np

from . import binding as B
from . import filter_design as fd

F = np.float32
MODE_BT, MODE_MT = 0, 1
SPEC_FREQ_MIN, SPEC_FREQ_MAX = F(10.0), F(24000.0)          # const.h:30-31
LPF, HPF, APF = range(3)


def select_filter(kind, mode, slope):                        # Crossover.cpp:162-190
    bt = (mode == MODE_BT)
    if slope == 1:
        return {LPF: fd.FLT_BT_RLC_LOPASS if bt else fd.FLT_MT_RLC_LOPASS,
                HPF: fd.FLT_BT_RLC_HIPASS if bt else fd.FLT_MT_RLC_HIPASS,
                APF: fd.FLT_BT_RLC_ALLPASS if bt else fd.FLT_MT_RLC_ALLPASS}[kind]
    return {LPF: fd.FLT_BT_LRX_LOPASS if bt else fd.FLT_MT_LRX_LOPASS,
            HPF: fd.FLT_BT_LRX_HIPASS if bt else fd.FLT_MT_LRX_HIPASS,
            APF: fd.FLT_BT_LRX_ALLPASS if bt else fd.FLT_MT_LRX_ALLPASS}[kind]


def select_slope(kind, slope):                               # Crossover.cpp:192-198
    if slope == 1:
        return 1 if kind == APF else 2
    return slope - 1


class Crossover:
    def __init__(self, bands):
        self.nsplits = bands - 1
        self.sr = 48000
        step = F(math.log(float(SPEC_FREQ_MAX / SPEC_FREQ_MIN))) / F(bands)
        self.split = [dict(band=i + 1, slope=0, freq=F(SPEC_FREQ_MIN * F(math.exp(float(F(i + 1) * step)))), mode=MODE_BT,
                           lpf_coef=None, lpf_state=None, lpf_params=None, hpf_coef=None, hpf_state=None, hpf_params=None,
                           hpf_key=None)
                      for i in range(self.nsplits)]
        # bands start out on the default split frequencies (Crossover.cpp:142-150); a band that is never enabled keeps them
        self.band = [dict(gain=F(1.0), start=(SPEC_FREQ_MIN if i == 0 else self.split[i - 1]["freq"]),
                          end=(self.split[i]["freq"] if i < self.nsplits else F(self.sr >> 1)),
                          enabled=False, p_start=None, p_end=None) for i in range(bands)]
        self.plan = []
        self.dirty = self.clear = True

    # ---- setters (Crossover.cpp:200-254, 327-341) ------------------------------------------------------------
    def set_sample_rate(self, sr):
        if sr != self.sr:
            self.sr = sr
            self.band[self.nsplits]["end"] = F(sr >> 1)        # Crossover.cpp:323
            self.dirty = self.clear = True

    def set_slope(self, sp, slope):
        if sp < self.nsplits and slope != self.split[sp]["slope"]:
            self.split[sp]["slope"] = slope; self.dirty = True

    def set_frequency(self, sp, freq):
        if sp < self.nsplits and F(freq) != self.split[sp]["freq"]:
            self.split[sp]["freq"] = F(freq); self.dirty = True

    def set_mode(self, sp, mode):
        if sp < self.nsplits and mode != self.split[sp]["mode"]:
            self.split[sp]["mode"] = mode; self.dirty = True

    def set_gain(self, band, gain):
        if band <= self.nsplits and F(gain) != self.band[band]["gain"]:
            self.band[band]["gain"] = F(gain); self.dirty = True

    # ---- Crossover::reconfigure --------------------------------------------------------------------------------
    def reconfigure(self):
        if not self.dirty:
            return
        plan = [i for i in range(self.nsplits) if self.split[i]["slope"] != 0]
        for b in self.band:
            b["enabled"] = False
        for si in range(len(plan) - 1):                      # the reference's exchange sort (:357-361)
            for sj in range(si + 1, len(plan)):
                if self.split[plan[sj]]["freq"] < self.split[plan[si]]["freq"]:
                    plan[si], plan[sj] = plan[sj], plan[si]
        self.plan = plan
        left = self.band[0]
        left["start"] = SPEC_FREQ_MIN; left["enabled"] = True; left["p_start"] = None
        for i, pi in enumerate(plan):
            sp = self.split[pi]
            right = self.band[sp["band"]]
            left["end"] = sp["freq"]; left["p_end"] = i
            right["start"] = sp["freq"]; right["p_start"] = i; right["enabled"] = True
            params = [fd.Params(select_filter(LPF, sp["mode"], sp["slope"]), select_slope(LPF, sp["slope"]),
                                sp["freq"], sp["freq"], left["gain"], 0.0)]
            for pj in plan[i + 1:]:
                x = self.split[pj]
                params.append(fd.Params(select_filter(APF, x["mode"], x["slope"]), select_slope(APF, x["slope"]),
                                        x["freq"], x["freq"], 1.0, 0.0))
            coef = np.concatenate([fd.design(p, self.sr)[2] for p in params]).astype(np.float32).reshape(-1, 5)
            # FilterBank::end(clear): the state survives a retune unless the number of sections changed (or EF_CLEAR)
            if self.clear or sp["lpf_coef"] is None or sp["lpf_coef"].shape[0] != coef.shape[0]:
                sp["lpf_state"] = np.zeros((max(coef.shape[0], 1), 2), np.float32)
            sp["lpf_coef"], sp["lpf_params"] = coef, params
            g = F(1.0) if i + 1 < len(plan) else right["gain"]
            if sp["slope"] == 1:
                g = F(-g)
            hp = fd.Params(select_filter(HPF, sp["mode"], sp["slope"]), select_slope(HPF, sp["slope"]), sp["freq"], sp["freq"], g, 0.0)
            hcoef = fd.design(hp, self.sr)[2].astype(np.float32).reshape(-1, 5)
            key = (hp.nType, hp.nSlope)                      # Filter::update: FF_CLEAR when type or slope changed (Filter.cpp:157-158)
            if self.clear or sp["hpf_coef"] is None or key != sp["hpf_key"] or sp["hpf_coef"].shape[0] != hcoef.shape[0]:
                sp["hpf_state"] = np.zeros((max(hcoef.shape[0], 1), 2), np.float32)
            sp["hpf_coef"], sp["hpf_params"], sp["hpf_key"] = hcoef, hp, key
            left = right
        left["end"] = F(F(self.sr) * F(0.5)); left["p_end"] = None
        self.dirty = self.clear = False

    # ---- Crossover::process: returns {band: output} for the bands in `handlers` --------------------------------
    def process(self, x, handlers=None):
        self.reconfigure()
        x = np.asarray(x, np.float32)
        nb = self.nsplits + 1
        handlers = set(range(nb)) if handlers is None else set(handlers)
        out = {}
        if not self.plan:
            if 0 in handlers:
                out[0] = (x * self.band[0]["gain"]).astype(np.float32)
            return out
        src, left = x, 0
        for i, pi in enumerate(self.plan):
            sp = self.split[pi]
            if left in handlers:
                out[left], sp["lpf_state"] = B.biquad_cascade(src, sp["lpf_coef"], sp["lpf_state"])
            src, sp["hpf_state"] = B.biquad_cascade(src, sp["hpf_coef"], sp["hpf_state"])
            left = sp["band"]
        if left in handlers:
            out[left] = src
        return out

    def band_info(self, band):
        self.reconfigure()
        b = self.band[band]
        return dict(gain=float(b["gain"]), start=float(b["start"]), end=float(b["end"]),
                    active=True if band == 0 else bool(b["enabled"]))

    # ---- Crossover::freq_chart (packed complex form) ------------------------------------------------------------
    def freq_chart(self, band, freqs):
        self.reconfigure()
        f = np.asarray(freqs, np.float32)
        b = self.band[band]
        if not b["enabled"]:
            return np.zeros(f.size, np.complex64)
        if not self.plan:
            return np.ones(f.size, np.complex64)

        def chart(p):
            return fd.freq_chart(p, self.sr, f)[0].astype(np.complex64)
        if b["p_end"] is None:
            return chart(self.split[self.plan[b["p_start"]]]["hpf_params"])
        if b["p_start"] is None:
            h = np.ones(f.size, np.complex64)
            for p in self.split[self.plan[b["p_end"]]]["lpf_params"]:
                h = (h * chart(p)).astype(np.complex64)
            return h
        return (chart(self.split[self.plan[b["p_start"]]]["hpf_params"]) *
                chart(self.split[self.plan[b["p_end"]]]["lpf_params"][0])).astype(np.complex64)
