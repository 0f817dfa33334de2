"""Replays one seed of test_biquad_gpu.test_random_operation_sequences with a log of the operations (experiment)."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import filter_design as fd
import workloads as wl
gpu = importlib.import_module("lsp-dsp-units_amd")

seed = int(sys.argv[1])
rng = np.random.default_rng(11000 + seed)
C = 4
bank = gpu.BiquadBank(C, 8)
types = [fd.FLT_BT_RLC_BELL, fd.FLT_BT_RLC_HISHELF, fd.FLT_BT_LRX_LOPASS, fd.FLT_MT_RLC_BELL, fd.FLT_BT_BWC_HIPASS]
coef = [None] * C
hist = [np.zeros(0, np.float32) for _ in range(C)]
state0 = [None] * C
enabled = [True] * C


def redesign(c, clear):
    t = int(rng.choice(types))
    slope = int(rng.integers(1, 3))
    q = wl.design(t, slope, float(rng.uniform(700.0, 15000.0)), 0, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 2.0)))[:8]
    old = coef[c]
    keep = (old is not None) and (len(old) == len(q)) and not clear
    if keep:
        if hist[c].size:
            _, st = oracle.biquad_cascade(hist[c], old, state0[c])
            state0[c] = st
    else:
        state0[c] = None
    hist[c] = np.zeros(0, np.float32)
    coef[c] = q
    bank.set_chains(c, q, clear=clear)
    print("   redesign ch %d type %d slope %d sections %d clear %s keep %s" % (c, t, slope, len(q), clear, keep))


for c in range(C):
    redesign(c, True)
for step in range(40):
    op = rng.choice(["process", "process", "process", "redesign", "reset", "toggle", "ir"])
    if op == "process":
        n = int(rng.choice([1, 7, 15, 16, 17, 1000, 1024, 1031, 2048, 2049, 4096, int(rng.integers(1, 6000))]))
        x = rng.standard_normal((C, n)).astype(np.float32)
        din = gpu.DeviceBuffer.from_host(x)
        in_place = bool(rng.integers(0, 2))
        dout = din if in_place else gpu.DeviceBuffer.from_host(np.full((C, n), 9.0, np.float32))
        bank.process(dout, din, n)
        y = dout.download()
        msg = []
        for c in range(C):
            if not enabled[c]:
                msg.append("off")
                continue
            hist[c] = np.concatenate([hist[c], x[c]])
            ref, _ = oracle.biquad_cascade(hist[c], coef[c], state0[c])
            err = np.abs(y[c] - ref[-n:])
            msg.append("%.1e@%d" % (err.max() / max(np.abs(ref).max(), 1e-30), int(err.argmax())))
        print("step %2d process n=%d in_place=%s  rel err per channel: %s" % (step, n, in_place, " ".join(msg)))
    elif op == "redesign":
        print("step %2d" % step, end="")
        redesign(int(rng.integers(0, C)), bool(rng.integers(0, 2)))
    elif op == "reset":
        c = int(rng.integers(-1, C))
        bank.reset(None if c < 0 else c)
        for k in (range(C) if c < 0 else [c]):
            hist[k] = np.zeros(0, np.float32); state0[k] = None
        print("step %2d reset %d" % (step, c))
    elif op == "toggle":
        c = int(rng.integers(0, C))
        enabled[c] = not enabled[c]
        bank.set_row_enabled(c, enabled[c])
        print("step %2d toggle ch %d -> %s" % (step, c, enabled[c]))
    else:
        out = gpu.DeviceBuffer((C, 300))
        bank.impulse_response(out, 300)
        print("step %2d ir" % step)
bank.close()
