"""C2 shape, one block per call: process() (biquad_bank_kernel<16,2>) against process_blocks with K = 1 (biquad_stream_kernel<2>: two
sub-blocks) and K = 2, 4.  Wall clock of 400 calls on one stream.  python tests/experiments/biquad_single_block_paths.py"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
mi = importlib.import_module("lsp-dsp-units_amd")
dev = torch.device("cuda:0")
C, n, ring = 1024, 4096, 16
fc = np.exp(np.random.default_rng(3).uniform(np.log(200.0), np.log(18000.0), size=C))
coef = np.zeros((C, 8, 5), np.float32)
for c in range(C):
    coef[c] = mi.design_filter(47, 4, float(fc[c]), float(fc[c]), 1.0, 0.75, 48000)[2]      # FLT_BT_LRX_LOPASS
bank = mi.BiquadBank(C, 8)
bank.set_all_chains(coef)
x = (torch.randn((ring, C, n)) * 0.25).to(dev)
y = torch.empty_like(x)
stream = torch.cuda.Stream(device=dev)
bank.commit(stream)
st = ctypes.c_void_p(stream.cuda_stream)

def blocks(K):
    def run(i):
        po = (ctypes.c_void_p * K)(*[y[(i * K + k) % ring].data_ptr() for k in range(K)])
        pi = (ctypes.c_void_p * K)(*[x[(i * K + k) % ring].data_ptr() for k in range(K)])
        mi.check(mi.lib.mi_biquad_bank_process_blocks(bank.handle, po, pi, K, n, n, n, st))
    return run

for name, fn, K in (("process()", lambda i: bank.process(y[i % ring], x[i % ring], n, stream=stream), 1),
                    ("process_blocks K=1", blocks(1), 1), ("process_blocks K=2", blocks(2), 2), ("process_blocks K=4", blocks(4), 4)):
    for i in range(20):
        fn(i)
    stream.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(400 // K):
            fn(i)
        stream.synchronize()
        t = (time.perf_counter() - t0) / 400
        best = t if best is None else min(best, t)
    print("%-22s %.2f us per block   last launch %s" % (name, best * 1e6, mi.last_launch()), flush=True)
