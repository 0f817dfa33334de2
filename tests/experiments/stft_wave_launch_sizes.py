"""How long a launch of stft_wave_blocks_kernel takes for runs of 1 .. 64 blocks (experiment): python tests/experiments/stft_wave_launch_sizes.py"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
C, rank = 1024, 12
N = 1 << rank
sp = gpu.SpectralBank(C, rank); sp.set_rank(rank)
sp.bind_mask(np.linspace(1.0, 0.25, N // 2 + 1).astype(np.float32))
x = torch.randn((64, C, N), device="cuda") * 0.25
y = torch.empty_like(x)
ins = [x[k] for k in range(64)]; outs = [y[k] for k in range(64)]
sp.process(outs[0], ins[0], N)
for K in (1, 2, 4, 8, 16, 32, 64):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(20):
            if K == 1:
                sp.process(outs[0], ins[0], N)
            else:
                sp.process_blocks(outs[:K], ins[:K], N)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("K = %2d: %.1f us per launch, %.2f us per block" % (K, dt * 1e6, dt * 1e6 / K))
