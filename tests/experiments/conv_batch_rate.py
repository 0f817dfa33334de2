"""C3 geometry (256 channels, 65 536 taps, rank 13): us per 4096-sample frame, a launch per frame (conv_step_kernel) against
mi_convolver_bank_process_blocks in batches of K frames.  usage: conv_batch_rate.py [channels = 256]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np
import torch

mi = importlib.import_module("lsp-dsp-units_amd")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
taps, frame = int(os.environ.get("MI_RATE_TAPS", "65536")), 4096   # (MI_RATE_TAPS: another partition count)
rng = np.random.default_rng(4)
irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
dev = torch.device("cuda:0")
ring = 16
x = torch.randn(ring, C, frame, device=dev)
y = torch.empty_like(x)
bank = mi.ConvolverBank(irs, 13)

def per_frame(n):
    for i in range(n):
        bank.process(y[i % ring], x[i % ring], frame)

def batched(n, K):
    for i in range(0, n, K):
        bank.process_blocks([y[(i + j) % ring] for j in range(K)], [x[(i + j) % ring] for j in range(K)], frame)

def timed(fn, *a):
    fn(*a); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(*a); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / a[0]

n = 320
if len(sys.argv) <= 2:
    print("%d channels: a launch per frame %.2f us per frame" % (C, timed(per_frame, n)))
for K in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (2, 4, 8, 16)):
    print("  batches of %2d frames: %.2f us per frame" % (K, timed(batched, n, K)))
print("faults", bank.faults())
