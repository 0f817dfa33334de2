"""Experiment: where the fused crossover launch spends its time -- bands (sections) and written outputs varied separately.
1024 channels x 4096 samples, LR4 split points; time per process() call over 200 calls."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, n, K = 1024, 4096, 200
freqs = (200.0, 1500.0, 7000.0, 12000.0, 15000.0)
x = (torch.randn((4, C, n)) * 0.25).cuda()

def run(bands, slope, written):
    xo = mi.CrossoverBank(C, bands)
    xo.set_sample_rate(48000)
    for i in range(bands - 1):
        xo.set_slope(i, slope); xo.set_frequency(i, freqs[i])
    outs = [[torch.empty((C, n), device="cuda") if b in written else None for b in range(bands)] for _ in range(4)]
    for i in range(10):
        xo.process(outs[i % 4], x[i % 4], n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        xo.process(outs[i % 4], x[i % 4], n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e6
    xo.close()
    return dt

for bands, slope, written in ((2, 2, (0, 1)), (3, 2, (0, 1, 2)), (4, 2, (0, 1, 2, 3)), (4, 2, (3,)), (4, 2, (0,)), (4, 1, (0, 1, 2, 3)), (4, 3, (0, 1, 2, 3)),
                              (6, 2, (0, 1, 2, 3, 4, 5))):
    print("bands %d slope %d outputs written %s: %.2f us per call" % (bands, slope, written, run(bands, slope, written)))
