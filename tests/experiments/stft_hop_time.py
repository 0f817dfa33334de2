"""Experiment: time of one SpectralProcessor hop with a fused gain mask (stft_hop_kernel<11, 1>), 1024 channels, rank 12,
one hop (2048 samples) per call, 200 calls in one hipGraph-free loop.   python tests/experiments/stft_hop_time.py [channels]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rank, n, K = 12, 2048, 200
bank = mi.SpectralBank(C, rank)
bank.set_rank(rank)
bank.bind_mask(np.linspace(1.0, 0.2, (1 << (rank - 1)) + 1).astype(np.float32))
ring = 8
x = (torch.randn((ring, C, n)) * 0.25).cuda(); y = torch.empty((ring, C, n), device="cuda")
for i in range(20):
    bank.process(y[i % ring], x[i % ring], n)
torch.cuda.synchronize()
ts = []
for r in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        bank.process(y[i % ring], x[i % ring], n)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
ts.sort()
print("%d channels: %.2f us per hop call (median of 7 runs of %d), best %.2f" % (C, ts[3], K, ts[0]))
