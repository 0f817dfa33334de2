import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import delay as od
seed = int(sys.argv[1]); stop = int(sys.argv[2])
rng = np.random.default_rng(7000 + seed)
C, maxd = 3, int(rng.choice([100, 511, 513, 1000]))
bank = gpu.DelayBank(C, maxd); refs = [od.Delay(maxd) for _ in range(C)]
size = refs[0].size
for step in range(60):
    op = rng.choice(["plain", "scalar", "vector", "add", "add_vector", "ramp", "ramp_gain", "set", "clear", "append"])
    n = int(rng.choice([1, 2, 63, size - 1, size, size + 1, 2 * size + 5, int(rng.integers(1, 3 * size))]))
    if op == "set":
        d = int(rng.integers(0, maxd + 1)); c = int(rng.integers(-1, C))
        if c < 0:
            bank.set_delay(d); [r.set_delay(d) for r in refs]
        else:
            bank.set_delay(d, c); refs[c].set_delay(d)
    elif op == "clear":
        bank.clear(); [r.buf.fill(0) for r in refs]
    elif op == "append":
        x = rng.standard_normal((C, n)).astype(np.float32); bank.append(gpu.DeviceBuffer.from_host(x), n); [r.append(x[c]) for c, r in enumerate(refs)]
    else:
        x = rng.standard_normal((C, n)).astype(np.float32); g = rng.uniform(0.5, 2.0, (C, n)).astype(np.float32); base = rng.standard_normal((C, n)).astype(np.float32)
        in_place = bool(rng.integers(0, 2)) and not op.startswith("add")
        din = gpu.DeviceBuffer.from_host(x); dout = din if in_place else gpu.DeviceBuffer.from_host(base); dg = gpu.DeviceBuffer.from_host(g)
        if op == "plain": bank.process(dout, din, n); ref = [r.process(x[c]) for c, r in enumerate(refs)]
        elif op == "scalar": bank.process(dout, din, n, gain=0.37); ref = [r.process(x[c], gain=0.37) for c, r in enumerate(refs)]
        elif op == "vector": bank.process(dout, din, n, gain_vec=dg); ref = [r.process(x[c], gain=g[c]) for c, r in enumerate(refs)]
        elif op == "add": bank.process(dout, din, n, add=True); ref = [r.process(x[c], add_to=base[c]) for c, r in enumerate(refs)]
        elif op == "add_vector": bank.process(dout, din, n, add=True, gain_vec=dg); ref = [r.process(x[c], gain=g[c], add_to=base[c]) for c, r in enumerate(refs)]
        else:
            targets = [int(rng.integers(0, maxd + 1)) for _ in range(C)]
            if rng.integers(0, 4) == 0: targets[0] = refs[0].delay
            before = [(r.delay, r.head, r.tail) for r in refs]
            if op == "ramp": bank.process_ramping(dout, din, targets, n); ref = [r.process_ramping(x[c], targets[c]) for c, r in enumerate(refs)]
            else: bank.process_ramping(dout, din, targets, n, gain_vec=dg); ref = [r.process_ramping(x[c], targets[c], gain=g[c]) for c, r in enumerate(refs)]
            if step == stop: print("ramp", n, "targets", targets, "before", before, "in_place", in_place, "size", size, "x", x, "g", g)
        y = dout.download()
        if step == stop or not np.array_equal(y, np.stack(ref)):
            print(step, op, n, "equal", np.array_equal(y, np.stack(ref))); print(y); print(np.stack(ref)); 
            if step == stop: break
    print(step, op, n, [bank.get(c) for c in range(1)], refs[0].head, refs[0].tail)
