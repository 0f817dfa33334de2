"""Which banks give the same samples when an output buffer is the input buffer?  (audit, round 4)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np
mi = importlib.import_module("lsp-dsp-units_amd")
rng = np.random.default_rng(1)

def compare(name, make, call, shape, calls=3):
    res = []
    xs = [(rng.standard_normal(shape) * 0.25).astype(np.float32) for _ in range(calls)]
    for inplace in (False, True):
        b = make()
        ys = []
        for x in xs:
            d = mi.DeviceBuffer.from_host(x)
            o = d if inplace else mi.DeviceBuffer(shape)
            call(b, o, d)
            ys.append(o.download())
        res.append(np.stack(ys)); b.close()
    print("%-28s in place == apart: %s (max diff %.3g)" % (name, np.array_equal(res[0], res[1]), float(np.abs(res[0] - res[1]).max())))

C, n = 3, 4096
def mk_split():
    s = mi.SplitterBank(C, 10, 2)
    s.bind_copy(0); s.bind_mask(1, np.linspace(1.0, 0.0, 1024).astype(np.float32))
    return s
compare("splitter (band 0 = input)", mk_split, lambda b, o, d: b.process([o, mi.DeviceBuffer((C, n))], d, n), (C, n))
def mk_dyn():
    df = mi.DynFilterBank(C, 1); df.set_sample_rate(48000); df.set_params(0, 11, 2, 1000.0, 1000.0, 1.0, 2.0); df.set_filter_active(0, True)
    return df
curve = mi.DeviceBuffer.from_host((1.0 + 0.5 * np.sin(np.arange(n) / 100.0))[None, :].repeat(C, 0).astype(np.float32))
compare("dynamic filters", mk_dyn, lambda b, o, d: b.process(0, o, d, curve, n), (C, n))
def mk_delay():
    dl = mi.DelayBank(C, 3000)
    for c in range(C): dl.set_delay(100 + 50 * c, channel=c)
    return dl
compare("delay", mk_delay, lambda b, o, d: b.process(o, d, n), (C, n))
def mk_lm():
    lm = mi.LoudnessBank(1, C, 400.0); lm.set_sample_rate(48000); return lm
