"""Time per 4096-sample step of the splitter bank against the number of bands and channels (experiment)."""
import importlib, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
mi = importlib.import_module("lsp-dsp-units_amd")
rank, n = 12, 4096
for C, bands in ((256, 1), (256, 2), (256, 4), (256, 8), (1024, 4), (64, 4)):
    sp = mi.SplitterBank(C, rank, bands)
    for b in range(bands):
        sp.bind_mask(b, np.ones(1 << rank, np.float32))
    x = torch.randn(C, n, device="cuda") * 0.25
    outs = [torch.empty(C, n, device="cuda") for _ in range(bands)]
    for _ in range(20):
        sp.process(outs, x, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        sp.process(outs, x, n)
    torch.cuda.synchronize()
    print("channels %4d bands %d: %.1f us per step (two hops)" % (C, bands, (time.perf_counter() - t0) / K * 1e6))
    sp.close()
