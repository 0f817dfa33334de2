"""SpectralProcessor bank (rank 12, 1024 channels, fused gain mask): us per 4096 samples in calls of 4096 .. 65536 samples."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, rank = 1024, 12
sp = mi.SpectralBank(C, rank)
sp.set_rank(rank)
sp.bind_mask(np.linspace(1.0, 0.25, (1 << (rank - 1)) + 1).astype(np.float32))
dev = torch.device("cuda:0")
for n in (4096, 8192, 16384, 65536):
    x = torch.randn(C, n, device=dev) * 0.25
    y = torch.empty_like(x)
    reps = max(4, 262144 // n)
    for _ in range(3):
        sp.process(y, x, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sp.process(y, x, n)
    torch.cuda.synchronize()
    print("calls of %6d samples: %.2f us per 4096 samples" % (n, (time.perf_counter() - t0) * 1e6 / reps * 4096 / n), flush=True)
