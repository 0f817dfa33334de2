"""C2 (1024 channels x 4096 samples x 8 sections, runs of 100 blocks in one launch): does the row stride of the caller's buffers
matter?  Rows 16 KiB apart (stride 4096) against rows padded by 32 / 64 / 256 / 1056 samples.  usage: biquad_row_stride.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
import numpy as np
import torch
import workloads as wl
mi = importlib.import_module("lsp-dsp-units_amd")
C, n, K, ring = 1024, 4096, 100, 8
dev = torch.device("cuda:0")
coef, _ = wl.c2_coefficients(C)
for rep in range(2):
    for pad in (0, 32, 64, 256, 1056):
        stride = n + pad
        bank = mi.BiquadBank(C, 8)
        bank.set_all_chains(coef)
        x = torch.randn(ring, C, stride, device=dev) * 0.25
        y = torch.empty_like(x)
        outs = [y[k % ring] for k in range(K)]
        ins = [x[k % ring] for k in range(K)]
        for _ in range(3):
            bank.process_blocks(outs, ins, n, stride, stride)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            bank.process_blocks(outs, ins, n, stride, stride)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6 / K)
        ts.sort()
        print("row stride %5d samples: %.2f us per block (median of 9 calls of %d blocks; min %.2f)" % (stride, ts[4], K, ts[0]), flush=True)
        bank.close()
