"""SpectralSplitter bank (rank 12, 4 bands, 256 channels): us per 4096 samples in calls of 4096 .. 65536 samples."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, bands, rank = 256, 4, 12
sp = mi.SplitterBank(C, rank, bands)
edges = [(None, (300.0, -32.0)), ((300.0, -32.0), (2000.0, -32.0)), ((2000.0, -32.0), (8000.0, -32.0)), ((8000.0, -32.0), None)]
for b, (hp, lp) in enumerate(edges):
    sp.bind_mask(b, mi.crossover_fft_mask(hp, lp, 1.0, 1.0, 48000, rank))
dev = torch.device("cuda:0")
for n in (4096, 8192, 16384, 65536):
    x = torch.randn(C, n, device=dev) * 0.25
    outs = [torch.empty_like(x) for _ in range(bands)]
    reps = max(4, 262144 // n)
    for _ in range(3):
        sp.process(outs, x, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sp.process(outs, x, n)
    torch.cuda.synchronize()
    print("calls of %6d samples: %.2f us per 4096 samples" % (n, (time.perf_counter() - t0) * 1e6 / reps * 4096 / n), flush=True)
