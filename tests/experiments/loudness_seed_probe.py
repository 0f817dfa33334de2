"""Experiment: replays one seed of tests/test_loudness_gpu.py::test_random_operation_sequences and, at the last sample of every
process() call, compares the GPU bank and the float32 oracle with the window sums evaluated in float64 from the oracle's own
lines of squares (who is closer to exact arithmetic?).   python tests/experiments/loudness_seed_probe.py <seed>"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import loudness as ol

seed = int(sys.argv[1])
rng = np.random.default_rng(15000 + seed)
M, K, sr = 2, 3, 48000
bank = gpu.LoudnessBank(M, K, 200.0)
refs = [ol.LoudnessMeter(K, 200.0) for _ in range(M)]
for obj in [bank] + refs:
    obj.set_sample_rate(sr)
weight = ol.WEIGHT_K
for step in range(40):
    op = rng.choice(["process", "process", "process", "period", "weighting", "designation", "link", "active", "bound", "clear"])
    if op == "process":
        n = int(rng.choice([1, 100, 1023, 1024, 1025, 4096, 4097, int(rng.integers(1, 9000))]))
        x = (rng.standard_normal((M * K, n)) * 0.2).astype(np.float32)
        g = [None, 1.0, 0.5][int(rng.integers(0, 3))]
        out = gpu.DeviceBuffer((M, n)); ch = gpu.DeviceBuffer.from_host(np.full((M * K, n), -1.0, np.float32))
        bank.process(out, ch, gpu.DeviceBuffer.from_host(x), n, gain=g)
        y = out.download()
        for m in range(M):
            r = refs[m]
            if len(sys.argv) > 2 and step == int(sys.argv[2]):
                print("   oracle filter memory before the call:", [None if cc["state"] is None else cc["state"].tolist() for cc in r.ch])
            o, c = r.process(x[m * K:(m + 1) * K], gain=g)
            exact = 0.0
            tail = (r.head + r.size - r.period) & (r.size - 1)
            for cch in r.ch:
                if not cch["enabled"] or not cch["bound"]:
                    continue
                d = cch["data"].astype(np.float64)
                s = d[tail:r.head].sum() if tail < r.head else d[:r.head].sum() + d[tail:].sum()
                exact += float(cch["weight"]) * s / r.period
            gg = g or 1.0
            if len(sys.argv) > 2 and step == int(sys.argv[2]):
                yc = ch.download()
                print("   flags", [(cc["enabled"], cc["bound"], float(cc["weight"]), float(cc["link"])) for cc in r.ch], "gain", g)
                for k in range(K):
                    d = np.abs(yc[m * K + k].astype(np.float64) - c[k])
                    print("   channel %d: max |gpu - oracle| %.3e at %d, oracle there %.6e; first samples gpu %s oracle %s"
                          % (k, d.max(), int(d.argmax()), float(c[k][int(d.argmax())]), yc[m * K + k][:4].tolist(), c[k][:4].tolist()))
                dd = np.abs(y[m].astype(np.float64) - o)
                print("   meter: max |gpu - oracle| %.3e at %d; gpu %s oracle %s" % (dd.max(), int(dd.argmax()), y[m][:6].tolist(), o[:6].tolist()))
                print("   rel diff along the call:", ((y[m].astype(np.float64) - o) / np.maximum(o, 1e-30))[::10].tolist())
            yo, oo = (float(y[m][-1]) / gg) ** 2, (float(o[-1]) / gg) ** 2
            print("step %2d m %d n %5d weight %d period %5d  exact %.6e  oracle-exact %+.2e  gpu-exact %+.2e  (rel %+.1e %+.1e)  refresh-left %d"
                  % (step, m, n, weight, r.period, exact, oo - exact, yo - exact, (oo - exact) / max(exact, 1e-30), (yo - exact) / max(exact, 1e-30), r.refresh))
    elif op == "period":
        p = float(rng.choice([50.0, 120.0, 200.0, 400.0]))
        for obj in [bank] + refs:
            obj.set_period(p)
    elif op == "weighting":
        weight = int(rng.choice([ol.WEIGHT_NONE, ol.WEIGHT_K, ol.WEIGHT_K, ol.WEIGHT_A]))
        for obj in [bank] + refs:
            obj.set_weighting(weight)
    elif op == "designation":
        k, d = int(rng.integers(0, K)), int(rng.choice([ol.CHANNEL_LEFT, ol.CHANNEL_CENTER, 7, ol.CHANNEL_LFE1, ol.CHANNEL_NONE]))
        for obj in [bank] + refs:
            obj.set_designation(k, d)
    elif op == "link":
        k, l = int(rng.integers(0, K)), float(rng.choice([0.0, 0.3, 1.0, 1.5, -0.5]))
        for obj in [bank] + refs:
            obj.set_link(k, l)
    elif op == "active":
        k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2))
        for obj in [bank] + refs:
            obj.set_active(k, a)
    elif op == "bound":
        k, a = int(rng.integers(0, K)), bool(rng.integers(0, 2))
        for obj in [bank] + refs:
            obj.set_bound(k, a)
    else:
        for obj in [bank] + refs:
            obj.clear()
    if op != "process":
        print("step %2d %s" % (step, op), [(cc["enabled"], cc["bound"]) for cc in refs[0].ch])
