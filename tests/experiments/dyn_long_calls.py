"""DynamicFilters: us per 4096 samples of 1024 channels in calls of 4096 .. 65536 samples (does a long call run faster per sample?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C = 1024
dev = torch.device("cuda:0")
df = mi.DynFilterBank(C, 1)
df.set_sample_rate(48000)
df.set_params(0, 11, 2, 1000.0, 1000.0, 1.0, 2.0)
df.set_filter_active(0, True)
for n in (4096, 8192, 16384, 65536):
    x = torch.randn(C, n, device=dev) * 0.25
    t = torch.arange(n, dtype=torch.float32, device=dev) / 4096
    curve = (1.0 + 0.8 * torch.sin(2.0 * 3.14159265 * (3.0 * t[None, :] + torch.rand((C, 1), device=dev)))).contiguous()
    out = torch.empty_like(x)
    reps = max(4, 262144 // n)
    for _ in range(3):
        df.process(0, out, x, curve, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        df.process(0, out, x, curve, n)
    torch.cuda.synchronize()
    print("calls of %6d samples: %.2f us per 4096 samples" % (n, (time.perf_counter() - t0) * 1e6 / reps * 4096 / n), flush=True)
