import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import filter_design as fd
import workloads as wl
mi = importlib.import_module("lsp-dsp-units_amd")

def run(n, coef, nblocks=1, C=1, seed=1):
    x = (np.random.default_rng(seed).standard_normal((nblocks, C, n)) * 0.25).astype(np.float32)
    bank = mi.BiquadBank(C, max(1, len(coef)))
    for c in range(C): bank.set_chains(c, coef)
    din = mi.DeviceBuffer((C, n)); dout = mi.DeviceBuffer((C, n))
    st = None
    for b in range(nblocks):
        din.upload(x[b]); bank.process(dout, din, n); y = dout.download()
        for c in range(C):
            ref, st_c = oracle.biquad_cascade(x[b, c], coef, st if C == 1 else None)
            if C == 1: st = st_c
            err = np.abs(y[c] - ref)
            i = int(err.argmax())
            print("n=%d ns=%d block=%d ch=%d maxerr=%.3e at %d (chunk32 %d, pos %d) peak=%.3f" % (n, len(coef), b, c, err.max(), i, i // 32, i % 32, np.abs(ref).max()))
            bad = np.nonzero(err > 1e-4)[0]
            if len(bad): print("   first bad", bad[:8], "count", len(bad))
    print("   gpu state", bank.get_state()[0].ravel()[:8], "ref", None if st is None else st.ravel()[:8])
    bank.close()

bq2 = wl.design(fd.FLT_BT_BWC_HISHELF, 2, 1000.0, 1000.0, 1.9952623, 0.0)
bq8 = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 3000.0, 0, 1.0, 0.75)
run(4096, bq2); run(4096, bq2, nblocks=3); run(4096, bq8, nblocks=2); run(512, bq2, nblocks=2); run(2944, bq2, nblocks=2); run(100, bq2, nblocks=2); run(8192, bq2)
