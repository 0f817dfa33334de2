"""Experiment: launches of one MultiSpectralProcessor-style hop (eager timing, a bound function that leaves the spectra as they
are), 1024 channels, rank 12, one hop per call -- run under rocprofv3 --kernel-trace --stats to see the kernels per hop."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, rank, n, K = 1024, 12, 2048, 100
bank = mi.SpectralBank(C, rank)
bank.set_rank(rank); bank.set_timing(True)
bank.bind(lambda spec, r, ch, st: None)
x = (torch.randn((4, C, n)) * 0.25).cuda(); y = torch.empty((4, C, n), device="cuda")
for i in range(10):
    bank.process(y[i % 4], x[i % 4], n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    bank.process(y[i % 4], x[i % 4], n)
torch.cuda.synchronize()
print("%.2f us per hop call (python callback included)" % ((time.perf_counter() - t0) / K * 1e6))
# identity: the function changed nothing, so the output is the input delayed by the latency (2^rank samples = 2 calls)
assert float((y[(K - 1) % 4] - x[(K - 3) % 4]).abs().max()) < 1e-4, float((y[(K - 1) % 4] - x[(K - 3) % 4]).abs().max())
