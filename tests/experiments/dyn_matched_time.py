# times DynamicFilters calls of 1024 channels x 4096 samples for a few filter types, per-type kernels against the
# any-type kernel (MI_DYNFILTER_GENERIC=1 in the environment):  python3 tests/experiments/dyn_matched_time.py
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
gpu = importlib.import_module("lsp-dsp-units_amd")
C, n = 1024, 4096
x = (torch.randn((C, n)) * 0.25).cuda()
t = torch.arange(n, dtype=torch.float32) / n
curve = (1.0 + 0.8 * torch.sin(2.0 * 3.14159265 * (3.0 * t[None, :] + torch.rand((C, 1))))).contiguous().cuda()
out = torch.empty_like(x)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dynamic_filters as odf
from oracle import filter_design as ofd
types = [t for t in range(1, len(ofd.FILTER_TYPES)) if odf.cascade_count(t, 1) > 0]
if len(sys.argv) > 1:
    types = [t for t in types if (t & 1) == (1 if sys.argv[1] == "bilinear" else 0)]
for typ in types:
    name = ofd.FILTER_TYPES[typ]
    df = gpu.DynFilterBank(C, 1)
    df.set_sample_rate(48000)
    df.set_params(0, typ, 2 if odf.cascade_count(typ, 2) <= 16 else 1, 1000.0, 3000.0, 1.0, 2.0)
    df.set_filter_active(0, True)
    for _ in range(5):
        df.process(0, out, x, curve, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        df.process(0, out, x, curve, n)
    torch.cuda.synchronize()
    print("type %2d %-20s %7.1f us per call%s" % (typ, name, (time.perf_counter() - t0) / K * 1e6, "  (any-type kernel)" if os.environ.get("MI_DYNFILTER_GENERIC") else ""))
    assert bool(torch.isfinite(out).all())
    df.close()
