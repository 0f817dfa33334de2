import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import loudness as ol
seed = int(sys.argv[1])
rng = np.random.default_rng(15000 + seed)
M, K, sr = 2, 3, 48000
bank = gpu.LoudnessBank(M, K, 200.0); refs = [ol.LoudnessMeter(K, 200.0) for _ in range(M)]
for obj in [bank] + refs: obj.set_sample_rate(sr)
weight = ol.WEIGHT_K
for step in range(40):
    op = rng.choice(["process", "process", "process", "period", "weighting", "designation", "link", "active", "clear"])
    if op == "process":
        n = int(rng.choice([1, 100, 1023, 1024, 1025, 4096, 4097, int(rng.integers(1, 9000))]))
        x = (rng.standard_normal((M * K, n)) * 0.2).astype(np.float32); g = float(rng.choice([1.0, 0.5]))
        out = gpu.DeviceBuffer((M, n)); ch = gpu.DeviceBuffer.from_host(np.full((M * K, n), -1.0, np.float32))
        bank.process(out, ch, gpu.DeviceBuffer.from_host(x), n, gain=g); y, yc = out.download(), ch.download()
        errs = []
        for m in range(M):
            o, c = refs[m].process(x[m * K:(m + 1) * K], gain=g)
            peak = max(float(np.abs(o).max()), float(np.abs(c).max()), 1e-3)
            e = [float(np.abs(y[m] - o).max()) / peak] + [float(np.abs(yc[m * K + k] - c[k]).max()) / peak if refs[m].ch[k]["enabled"] else 0.0 for k in range(K)]
            errs.append(["%.1e" % v for v in e])
        print(step, "process", n, "period", refs[0].period, "head", refs[0].head, "refresh", refs[0].refresh, errs)
    elif op == "period":
        p = float(rng.choice([50.0, 120.0, 200.0, 400.0])); [obj.set_period(p) for obj in [bank] + refs]; print(step, "period", p)
    elif op == "weighting":
        weight = int(rng.choice([ol.WEIGHT_NONE, ol.WEIGHT_K, ol.WEIGHT_K, ol.WEIGHT_A])); [obj.set_weighting(weight) for obj in [bank] + refs]; print(step, "weighting", weight)
    elif op == "designation":
        k, d = int(rng.integers(0, K)), int(rng.choice([ol.CHANNEL_LEFT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1, ol.CHANNEL_NONE])); [obj.set_designation(k, d) for obj in [bank] + refs]; print(step, "designation", k, d)
    elif op == "link":
        k, l = int(rng.integers(0, K)), float(rng.choice([0.0, 0.3, 1.0, 1.5, -0.5])); [obj.set_link(k, l) for obj in [bank] + refs]; print(step, "link", k, l)
    elif op == "active":
        k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2)); [obj.set_active(k, a) for obj in [bank] + refs]; print(step, "active", k, a)
    else:
        [obj.clear() for obj in [bank] + refs]; print(step, "clear")
