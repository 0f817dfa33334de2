"""LoudnessMeter / ILUFSMeter banks (512 stereo meters): us per 4096 samples in calls of 4096 .. 32768 samples."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
M, K = 512, 2
dev = torch.device("cuda:0")
lm = mi.LoudnessBank(M, K, 400.0); lm.set_sample_rate(48000)
im = mi.ILUFSBank(M, K, 10.0, 400.0); im.set_sample_rate(48000)
for n in (4096, 8192, 16384, 32768):
    x = torch.randn(M * K, n, device=dev) * 0.25
    o1 = torch.empty(M, n, device=dev); o2 = torch.empty(M, n, device=dev)
    for name, fn in (("loudness", lambda: lm.process(o1, None, x, n)), ("ilufs", lambda: im.process(o2, x, n))):
        reps = max(4, 131072 // n)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        print("%-9s calls of %6d samples: %.2f us per 4096 samples" % (name, n, (time.perf_counter() - t0) * 1e6 / reps * 4096 / n), flush=True)
