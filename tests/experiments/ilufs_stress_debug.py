import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import ilufs as oi, loudness as ol
seed = int(sys.argv[1])
rng = np.random.default_rng(17000 + seed)
M, K, sr = 2, 2, 48000
max_int = float(rng.choice([0.0, 2.0]))
bank = gpu.ILUFSBank(M, K, max_int, 100.0); refs = [oi.ILUFSMeter(K, max_int, 100.0) for _ in range(M)]
for obj in [bank] + refs: obj.set_sample_rate(sr)
blk = refs[0].block_size
print("max_int", max_int, "blk", blk, "ms_size", refs[0].ms_size)
for step in range(40):
    op = rng.choice(["process", "process", "process", "period", "weighting", "designation", "active", "clear"])
    if op == "process":
        n = int(rng.choice([1, blk - 1, blk, blk + 1, 4 * blk, 4 * blk + 1, 9 * blk + 7, int(rng.integers(1, 12 * blk))]))
        x0 = rng.standard_normal((M * K, n)); amp = float(rng.choice([0.2, 1e-6]))
        x = (x0 * amp).astype(np.float32)
        g = float(rng.choice([1.0, 0.9235]))
        out = gpu.DeviceBuffer((M, n)); bank.process(out, gpu.DeviceBuffer.from_host(x), n, gain=g); got = out.download()
        want = np.stack([refs[m].process(x[m * K:(m + 1) * K], gain=g) for m in range(M)])
        print(step, "process", n, "err", float(np.abs(got - want).max()), "want max", float(want.max()), "loud", bank.loudness(), [float(r.loud) for r in refs],
              "ms_int", refs[0].ms_int, "count", refs[0].ms_count, "head", refs[0].ms_head, "int_time", float(refs[0].int_time), "amp", amp, "hist", refs[0].hist[:4], "blocks", [c["block"].tolist() for c in refs[0].ch], "part", refs[0].block_part, "off", refs[0].block_offset)
    elif op == "period":
        p = float(rng.choice([0.05, 0.4, 1.0, 2.0, 5.0])); [obj.set_integration_period(p) for obj in [bank] + refs]; print(step, "period", p, "->", float(refs[0].int_time))
    elif op == "weighting":
        w = int(rng.choice([ol.WEIGHT_NONE, ol.WEIGHT_K, ol.WEIGHT_K, ol.WEIGHT_A])); [obj.set_weighting(w) for obj in [bank] + refs]; print(step, "weighting", w)
    elif op == "designation":
        k, d = int(rng.integers(0, K)), int(rng.choice([ol.CHANNEL_LEFT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1])); [obj.set_designation(k, d) for obj in [bank] + refs]; print(step, "designation", k, d)
    elif op == "active":
        k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2)); [obj.set_active(k, a) for obj in [bank] + refs]; print(step, "active", k, a)
    else:
        [obj.clear() for obj in [bank] + refs]; print(step, "clear")
