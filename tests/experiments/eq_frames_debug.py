import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib, numpy as np
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import equalizer as oe, filter_design as fd
rank, mode = 12, oe.FIR
rng = np.random.default_rng(40 + rank)
C, nfilt, N = 5, 6, 1 << rank
blocks = 4
x = (rng.standard_normal((C, N * (blocks + 1))) * 0.25).astype(np.float32)
curves = [[(fd.FLT_BT_RLC_BELL, 1, float(f), float(f), float(g), 2.0) for f, g in zip(np.exp(rng.uniform(np.log(100), np.log(15000), nfilt)), 10 ** (rng.uniform(-9, 9, nfilt) / 20))] for _ in range(C)]
def make():
    eq = gpu.EqualizerBank(C, nfilt, rank); eq.set_mode(mode); eq.set_sample_rate(48000)
    for c in range(C):
        for i, p in enumerate(curves[c]): eq.set_params(i, *p, channel=c)
    return eq
def run(kind):
    eq = make(); ys = []
    d, o = gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, :N])), gpu.DeviceBuffer((C, N)); eq.process(o, d, N); ys.append(o.download())
    seg = x[:, N:]
    if kind == "each":
        for j in range(blocks):
            d, o = gpu.DeviceBuffer.from_host(np.ascontiguousarray(seg[:, j*N:(j+1)*N])), gpu.DeviceBuffer((C, N)); eq.process(o, d, N); ys.append(o.download())
    else:
        d, o = gpu.DeviceBuffer.from_host(np.ascontiguousarray(seg)), gpu.DeviceBuffer((C, N*blocks)); eq.process(o, d, N*blocks); ys.append(o.download())
    eq.close(); return np.concatenate(ys, axis=1)
a, b = run("each"), run("call")
bad = np.argwhere(a != b)
print("mismatches", len(bad), "of", a.size)
if len(bad):
    for c, i in bad[:5]: print(c, i, i // N, i % N, a[c, i], b[c, i])
    print("per block:", [int((a[:, k*N:(k+1)*N] != b[:, k*N:(k+1)*N]).sum()) for k in range(blocks + 1)])
    print("max abs diff", np.abs(a - b).max())
