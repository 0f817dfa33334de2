"""The Convolver fed in calls shorter than its frame, rank by rank: us per 4096 samples of every channel against whole-frame
calls.  Ranks 11 .. 13 (frames of 1024 .. 4096) take aligned 256-sample calls through conv_small_kernel; at ranks 9 and 10
(frames of 256 and 512) a sub-frame call goes through the time-domain kernel over the head partition (at most 512 taps).
usage: conv_call_stream.py  (prints one line per rank and call size)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np
import torch

mi = importlib.import_module("lsp-dsp-units_amd")
C = 256
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
only = [int(v) for v in sys.argv[1:3]]                    # [rank, call]: one case (for a kernel trace)
for rank, taps in ((9, 8192), (10, 8192), (11, 16384), (13, 65536)):
    frame = 1 << (rank - 1)
    if only and rank != only[0]:
        continue
    irs = (rng.standard_normal((C, taps)) * 1e-2).astype(np.float32)
    for call in (64, 128, 256, frame):
        if call > frame or (len(only) > 1 and call != only[1]):
            continue
        bank = mi.ConvolverBank(irs, rank)
        n = max(4096, frame)
        x = torch.randn(C, n, device=dev)
        y = torch.empty_like(x)
        def run(reps):
            for _ in range(reps):
                for d in range(0, n, call):
                    bank.process(y[:, d:], x[:, d:], call, n, n)
        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        run(reps)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / reps * (4096.0 / n)
        print("rank %2d (frame %4d, %5d taps)  calls of %4d: %8.1f us per 4096 samples%s" %
              (rank, frame, taps, call, us, "   <- whole frames" if call == frame else ""), flush=True)
        bank.close()
