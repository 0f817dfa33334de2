import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import equalizer as oe, filter_design as fd
seed = int(sys.argv[1]); TOL = 1e-5
rng = np.random.default_rng(seed)
C, nfilt, rank, sr = 2, 3, 7, 48000
N = 1 << rank
types = [fd.FLT_BT_RLC_BELL, fd.FLT_BT_RLC_HISHELF, fd.FLT_BT_RLC_LOSHELF, fd.FLT_MT_RLC_BELL, fd.FLT_BT_BWC_HIPASS,
         fd.FLT_BT_LRX_LOPASS, fd.FLT_DR_APO_PEAKING, fd.FLT_NONE]
modes = [oe.IIR, oe.FIR, oe.FFT, oe.SPM, oe.BYPASS]
eq = gpu.EqualizerBank(C, nfilt, rank); eq.set_sample_rate(sr)
refs = [oe.Equalizer(nfilt, rank) for _ in range(C)]
for o in refs: o.set_sample_rate(sr)
for step in range(60):
    op = rng.choice(["process", "process", "process", "retune", "mode", "reset", "smooth", "latency"])
    if op == "process":
        k = int(rng.choice([1, 7, N // 2 - 1, N // 2, N, N + 3, 3 * N, int(rng.integers(1, 4 * N))]))
        x = (rng.standard_normal((C, k)) * 0.25).astype(np.float32)
        dout = gpu.DeviceBuffer((C, k)); eq.process(dout, gpu.DeviceBuffer.from_host(x), k); y = dout.download()
        errs = []
        for c in range(C):
            ref = refs[c].process(x[c]); scale = max(float(np.abs(ref).max()), 0.25)
            errs.append(float(np.abs(y[c] - ref).max()) / scale)
        print(step, "process", k, "mode", refs[0].mode, "bufsize", refs[0].bufsize, "errs", ["%.2e" % e for e in errs], "xfade", refs[0].xfade)
    elif op == "retune":
        c = int(rng.integers(0, C)); i = int(rng.integers(0, nfilt))
        p = (int(rng.choice(types)), int(rng.integers(1, 3)), float(rng.uniform(800.0, 12000.0)),
             float(rng.uniform(800.0, 12000.0)), float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.0, 2.0)))
        eq.set_params(i, *p, channel=c); refs[c].set_params(i, fd.Params(*p)); print(step, "retune", c, i, p)
    elif op == "mode":
        m = int(rng.choice(modes)); eq.set_mode(m); [o.set_mode(m) for o in refs]; print(step, "mode", m)
    elif op == "reset":
        eq.reset(); [o.reset() for o in refs]; print(step, "reset")
    elif op == "smooth":
        s = bool(rng.integers(0, 2)); eq.set_smooth(s); [o.set_smooth(s) for o in refs]; print(step, "smooth", s)
    else:
        print(step, "latency", eq.get_latency(), [o.get_latency() for o in refs])
