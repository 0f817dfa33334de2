"""Small seeded scenarios shared by the golden-vector generator (tests/golden/make_vectors.py), the CPU test that pins the
oracle to the committed vectors and the GPU test that runs the banks against them.

The vectors are OUTPUTS OF THE ORACLE (the reference cannot be built here): they pin the oracle against
unintended change and give the GPU path a fixed target that does not depend on the oracle code at test time.  Inputs are
regenerated from their seeds.  Every case: oracle() -> {name: array}, gpu(mi) -> {name: array}, tol (relative to the peak
of the expected array; 0 = bit-exact)."""
import numpy as np

import oracle
from oracle import crossover as oc
from oracle import delay as od
from oracle import equalizer as oe
from oracle import filter_design as fd
from oracle import ilufs as oi
from oracle import loudness as ol
from oracle import spectral as osp
from oracle import splitter as ospl

SR = 48000


def _noise(seed, shape, scale=0.25):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


# ---- biquad -------------------------------------------------------------------------------------------------------
def _bq_coefs():
    c1 = fd.design(fd.Params(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, float(10 ** (6.0 / 20.0)), 0.0), SR)[2]
    c2 = fd.design(fd.Params(fd.FLT_BT_LRX_LOPASS, 4, 2500.0, 2500.0, 1.0, 0.75), SR)[2]
    return c1, c2


def biquad_oracle():
    c1, c2 = _bq_coefs()
    x = _noise(1, (2, 600))
    y1, _ = oracle.biquad_cascade(x[0], c1)
    ya, st = oracle.biquad_cascade(x[1, :333], c2)
    yb, _ = oracle.biquad_cascade(x[1, 333:], c2, st)
    return {"c1": y1, "c2": np.concatenate([ya, yb])}


def biquad_gpu(mi):
    c1, c2 = _bq_coefs()
    x = _noise(1, (2, 600))
    bank = mi.BiquadBank(2, 8)
    bank.set_chains(0, c1); bank.set_chains(1, c2)
    out = np.empty_like(x)
    for a, b in ((0, 333), (333, 600)):
        o = mi.DeviceBuffer((2, b - a))
        bank.process(o, mi.DeviceBuffer.from_host(x[:, a:b]), b - a)
        out[:, a:b] = o.download()
    bank.close()
    return {"c1": out[0], "c2": out[1]}


# ---- convolver ------------------------------------------------------------------------------------------------------
def _conv_data():
    ir = (_noise(2, (1, 300), 1.0)[0] * np.exp(-np.arange(300) / 80.0)).astype(np.float32)
    return ir, _noise(3, (1, 528), 1.0)


def convolver_oracle():
    ir, x = _conv_data()
    c = oracle.Convolver(ir, 8)
    return {"y": np.concatenate([c.process(x[0, a:b]) for a, b in ((0, 100), (100, 228), (228, 528))])}


def convolver_gpu(mi):
    ir, x = _conv_data()
    bank = mi.ConvolverBank(ir, 8)
    ys = []
    for a, b in ((0, 100), (100, 228), (228, 528)):
        o = mi.DeviceBuffer((1, b - a))
        bank.process(o, mi.DeviceBuffer.from_host(x[:, a:b]), b - a)
        ys.append(o.download()[0])
    bank.close()
    return {"y": np.concatenate(ys)}


# ---- equalizer ------------------------------------------------------------------------------------------------------
_EQ = [(fd.FLT_BT_RLC_BELL, 1, 2000.0, 2000.0, 2.0, 1.5), (fd.FLT_BT_RLC_HISHELF, 1, 8000.0, 8000.0, 0.5, 0.0)]


def equalizer_oracle():
    out = {}
    x = _noise(4, (1, 700))
    for name, mode in (("iir", oe.IIR), ("fir", oe.FIR), ("fft", oe.FFT), ("spm", oe.SPM)):
        e = oe.Equalizer(2, 6); e.set_sample_rate(SR); e.set_mode(mode)
        for i, p in enumerate(_EQ):
            e.set_params(i, fd.Params(*p))
        out[name] = np.concatenate([e.process(x[0, :257]), e.process(x[0, 257:])])
    return out


def equalizer_gpu(mi):
    out = {}
    x = _noise(4, (1, 700))
    for name, mode in (("iir", oe.IIR), ("fir", oe.FIR), ("fft", oe.FFT), ("spm", oe.SPM)):
        e = mi.EqualizerBank(1, 2, 6); e.set_sample_rate(SR); e.set_mode(mode)
        for i, p in enumerate(_EQ):
            e.set_params(i, *p)
        ys = []
        for a, b in ((0, 257), (257, 700)):
            o = mi.DeviceBuffer((1, b - a))
            e.process(o, mi.DeviceBuffer.from_host(x[:, a:b]), b - a)
            ys.append(o.download()[0])
        out[name] = np.concatenate(ys)
        e.close()
    return out


# ---- spectral processor / analyzer / splitter --------------------------------------------------------------------
def _mask(rank):
    return np.random.default_rng(50 + rank).uniform(0.0, 2.0, (1 << (rank - 1)) + 1).astype(np.float32)


def spectral_oracle():
    rank = 6
    m = _mask(rank)
    full = np.concatenate([m, m[-2:0:-1]]).astype(np.float32)
    p = osp.SpectralProcessor(rank)

    def cb(spec, r):
        o = spec.copy(); o[0::2] *= full; o[1::2] *= full
        return o
    p.bind(cb)
    x = _noise(5, (1, 500))
    an = osp.Analyzer(2, 6, SR, 1.0, 0)
    an.configure(sample_rate=SR, rate=100.0, rank=6, window_name="hann", reactivity=0.2, shift=1.0)
    xa = _noise(6, (2, 1500))
    an.process(xa)
    return {"stft": p.process(x[0]), "analyzer": an.get_spectrum(np.arange(33))}


def spectral_gpu(mi):
    rank = 6
    b = mi.SpectralBank(1, rank)
    b.bind_mask(_mask(rank))
    x = _noise(5, (1, 500))
    o = mi.DeviceBuffer((1, 500))
    b.process(o, mi.DeviceBuffer.from_host(x), 500)
    stft = o.download()[0]
    b.close()
    an = mi.AnalyzerBank(2, 6, SR, 1.0, 0)
    for what, v in ((an.SAMPLE_RATE, SR), (an.RATE, 100.0), (an.RANK, 6), (an.WINDOW, 0), (an.REACTIVITY, 0.2), (an.SHIFT, 1.0)):
        an.configure(what, v)
    an.process(mi.DeviceBuffer.from_host(_noise(6, (2, 1500))), 1500)
    spec = an.get_spectrum(np.arange(33, dtype=np.uint32))
    an.close()
    return {"stft": stft, "analyzer": spec}


def splitter_oracle():
    rank = 6
    sp = ospl.SpectralSplitter(rank, 2)
    sp.set_chunk_rank(5)
    masks = [ospl.lopass_fft_set(3000.0, -24.0, float(SR), rank), ospl.hipass_fft_set(3000.0, -24.0, float(SR), rank)]
    got = [[], []]
    for i, m in enumerate(masks):
        def func(spec, r, m=m):
            spec[0::2] *= m; spec[1::2] *= m
            return spec
        sp.bind(i, func, lambda s, first, count, i=i: got[i].append(s.copy()))
    x = _noise(7, (1, 400))
    sp.process(x[0], 400)
    return {"low": np.concatenate(got[0]), "high": np.concatenate(got[1])}


def splitter_gpu(mi):
    rank = 6
    b = mi.SplitterBank(1, rank, 2)
    b.set_chunk_rank(5)
    b.bind_mask(0, ospl.lopass_fft_set(3000.0, -24.0, float(SR), rank))
    b.bind_mask(1, ospl.hipass_fft_set(3000.0, -24.0, float(SR), rank))
    outs = [mi.DeviceBuffer((1, 400)), mi.DeviceBuffer((1, 400))]
    b.process(outs, mi.DeviceBuffer.from_host(_noise(7, (1, 400))), 400)
    res = {"low": outs[0].download()[0], "high": outs[1].download()[0]}
    b.close()
    return res


# ---- delay (bit-exact) ------------------------------------------------------------------------------------------
def delay_oracle():
    d = od.Delay(700)
    d.set_delay(100)
    x = _noise(8, (1, 3000), 1.0)[0]
    a = d.process(x[:900])
    b = d.process_ramping(x[900:1900], 650, gain=0.5)
    c = d.process_ramping(x[1900:1902], 20)                  # a fast change: the index wraps modulo 2^64
    e = d.process(x[1902:], gain=2.0)
    return {"y": np.concatenate([a, b, c, e])}


def delay_gpu(mi):
    d = mi.DelayBank(1, 700)
    d.set_delay(100)
    x = _noise(8, (1, 3000), 1.0)
    ys = []
    for a, b, kind in ((0, 900, "p"), (900, 1900, "r650"), (1900, 1902, "r20"), (1902, 3000, "g")):
        o = mi.DeviceBuffer((1, b - a)); i = mi.DeviceBuffer.from_host(x[:, a:b])
        if kind == "p":
            d.process(o, i, b - a)
        elif kind == "r650":
            d.process_ramping(o, i, [650], b - a, gain=0.5)
        elif kind == "r20":
            d.process_ramping(o, i, [20], b - a)
        else:
            d.process(o, i, b - a, gain=2.0)
        ys.append(o.download()[0])
    d.close()
    return {"y": np.concatenate(ys)}


# ---- crossover, meters ------------------------------------------------------------------------------------------------
def crossover_oracle():
    c = oc.Crossover(3)
    c.set_sample_rate(SR)
    for i, f in enumerate((600.0, 5000.0)):
        c.set_slope(i, 2); c.set_frequency(i, f)
    c.set_gain(1, 1.5)
    out = c.process(_noise(9, (1, 600))[0])
    return {"band%d" % b: out[b] for b in range(3)}


def crossover_gpu(mi):
    c = mi.CrossoverBank(1, 3)
    c.set_sample_rate(SR)
    for i, f in enumerate((600.0, 5000.0)):
        c.set_slope(i, 2); c.set_frequency(i, f)
    c.set_gain(1, 1.5)
    outs = [mi.DeviceBuffer((1, 600)) for _ in range(3)]
    c.process(outs, mi.DeviceBuffer.from_host(_noise(9, (1, 600))), 600)
    res = {"band%d" % b: outs[b].download()[0] for b in range(3)}
    c.close()
    return res


def meters_oracle():
    x = _noise(10, (2, 6000))
    m = ol.LoudnessMeter(2, 100.0)
    m.set_sample_rate(SR); m.set_period(20.0)
    out, _ = m.process(x)
    im = oi.ILUFSMeter(2, 1.0, 40.0)
    im.set_sample_rate(SR)
    iout = im.process(x, gain=1.0)
    return {"momentary": out[::10].copy(), "integrated": iout[::10].copy()}


def meters_gpu(mi):
    x = _noise(10, (2, 6000))
    m = mi.LoudnessBank(1, 2, 100.0)
    m.set_sample_rate(SR); m.set_period(20.0)
    o = mi.DeviceBuffer((1, 6000))
    m.process(o, None, mi.DeviceBuffer.from_host(x), 6000)
    mom = o.download()[0]
    m.close()
    im = mi.ILUFSBank(1, 2, 1.0, 40.0)
    im.set_sample_rate(SR)
    o2 = mi.DeviceBuffer((1, 6000))
    im.process(o2, mi.DeviceBuffer.from_host(x), 6000, gain=1.0)
    integ = o2.download()[0]
    im.close()
    return {"momentary": mom[::10].copy(), "integrated": integ[::10].copy()}


CASES = {
    "biquad": (biquad_oracle, biquad_gpu, 1e-5),
    "convolver": (convolver_oracle, convolver_gpu, 1e-5),
    "equalizer": (equalizer_oracle, equalizer_gpu, 1e-5),
    "spectral": (spectral_oracle, spectral_gpu, 1e-5),
    "splitter": (splitter_oracle, splitter_gpu, 1e-5),
    "delay": (delay_oracle, delay_gpu, 0.0),
    "crossover": (crossover_oracle, crossover_gpu, 1e-5),
    "meters": (meters_oracle, meters_gpu, 1e-5),
}
