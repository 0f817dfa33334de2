"""C3 at full size (256 and 512 channels, 65 536 taps, rank 13): mi_convolver_bank_process_blocks in batches of 16 frames against
process() frame by frame on a twin bank, bit for bit, over many frames, every other round with a second stream copying 256 MiB buffers meanwhile (the tail kernel's request queue lands by LDS-DMA: a
read in front of its data would show here as a rare difference).  usage: conv_batch_fullsize_bits.py [rounds = 6]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np
import torch

mi = importlib.import_module("lsp-dsp-units_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
busy = torch.cuda.Stream()
junk_a = torch.randn(64 << 20, device=dev)
junk_b = torch.empty_like(junk_a)
bad = 0
for C in (256, 512):
    taps, frame, K = 65536, 4096, 32
    rng = np.random.default_rng(C)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    a, b = mi.ConvolverBank(irs, 13), mi.ConvolverBank(irs, 13)
    for r in range(rounds):
        x = torch.randn(K, C, frame, device=dev)
        ya, yb = torch.empty_like(x), torch.empty_like(x)
        if busy is not None:                                # a second stream keeps the memory system busy meanwhile (odd rounds)
            with torch.cuda.stream(busy):
                for _ in range(40 if r % 2 else 0):
                    junk_b.copy_(junk_a)
        a.process_blocks([ya[k] for k in range(K)], [x[k] for k in range(K)], frame)
        for k in range(K):
            b.process(yb[k], x[k], frame)
        torch.cuda.synchronize()
        same = bool(torch.equal(ya, yb))
        if not same:
            bad += 1
            d = (ya != yb).nonzero()
            print("C %d round %d: %d samples differ, first at %s" % (C, r, d.shape[0], d[0].tolist()), flush=True)
    assert a.faults() == 0 and b.faults() == 0
    a.close(); b.close()
    print("%d channels: %d rounds of %d frames, %d rounds with differences so far" % (C, rounds, K, bad), flush=True)
sys.exit(1 if bad else 0)
