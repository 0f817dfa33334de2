"""How fast can the host enqueue biquad steps, and how long does the GPU take for them?  (experiment, not a test)"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
mi = importlib.import_module("lsp-dsp-units_amd")
from oracle import filter_design as fd  # noqa: E402

C, N, K = 1024, 4096, 2000
bank = mi.BiquadBank(C, 8)
rng = np.random.default_rng(3)
for c in range(C):
    f = float(np.exp(rng.uniform(np.log(200.0), np.log(18000.0))))
    bank.set_chains(c, mi.design_filter(fd.FLT_BT_LRX_LOPASS, 4, f, f, 1.0, 0.75, 48000)[2])
bank.commit()
x = torch.randn(16, C, N, device="cuda") * 0.25
y = torch.empty_like(x)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        bank.process(y[i & 15], x[i & 15], N)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.2f us/step, total %.2f us/step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
