# the delay bank's process() forms on resident buffers, 1024 channels x 4096 samples per call.  The loop below is bound by
# the eager launches (about 10 us per call from Python); kernel durations come from running it under
#   rocprofv3 --kernel-trace --stats -- python3 tests/experiments/delay_rate.py
# Round 3: delay_exchange_kernel 15.4 us (16 B per sample: 4.4 TB/s), delay_direct_kernel 13.2, ring_append_kernel 9.2 with one
# sample per lane; with four (calls on the quad grid): 11.3 (5.9 TB/s) / 7.2 / 6.5 us.  Replacing the per-element `% size` by an incrementally wrapped position made them SLOWER (17.3 / 15.1 / 10.6 us): at
# one element per thread the two 32-bit modulos cost less than setting the walk up.
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
C, n = 1024, 4096
x = torch.randn((C, n)).cuda(); y = torch.empty_like(x)
for maxd, d, what in ((48000, 24000, "delay 24000 (> block: one exchange pass)"), (48000, 1000, "delay 1000 (< block: direct + append)"), (4096, 0, "delay 0")):
    bank = gpu.DelayBank(C, maxd)
    bank.set_delay(d)
    for _ in range(5):
        bank.process(y, x, n)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        bank.process(y, x, n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-45s %7.1f us per call, %6.0f GB/s of 12 B per sample (in, ring write + read, out)" % (what, dt * 1e6, 12.0 * C * n / dt / 1e9))
    bank.close()
