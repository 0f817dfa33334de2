"""C2, 20-call regions: one hipGraph launch per region against 20 direct calls of the C-ABI with pre-built arguments."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
C, n, K, R = 1024, 4096, 20, 25
coef = mi.design_filter(47, 4, 3000.0, 3000.0, 1.0, 0.75)[2]
bank = mi.BiquadBank(C, 8)
for c in range(C):
    bank.set_chains(c, coef)
bank.commit()
ring = 16
x = (torch.randn((ring, C, n)) * 0.25).cuda(); y = torch.empty((ring, C, n), device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
f = mi.lib.mi_biquad_bank_process
args = [(bank.handle, ctypes.c_void_p(y[i % ring].data_ptr()), ctypes.c_void_p(x[i % ring].data_ptr()), ctypes.c_size_t(n),
         ctypes.c_size_t(n), ctypes.c_size_t(n), ctypes.c_void_p(st.cuda_stream)) for i in range(K)]
for a in args:
    f(*a)
torch.cuda.synchronize()
def region_direct():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for a in args:
        f(*a)
    torch.cuda.synchronize(); return time.perf_counter() - t0
ts = sorted(region_direct() for _ in range(R))
print("direct C-ABI calls: median region %.1f us = %.2f us per call (min %.1f)" % (ts[R // 2] * 1e6, ts[R // 2] * 1e6 / K, ts[0] * 1e6))
mi.check(mi.lib.mi_dspu_graph_begin_capture(ctypes.c_void_p(st.cuda_stream)))
for a in args:
    f(*a)
exe = ctypes.c_void_p(); mi.check(mi.lib.mi_dspu_graph_end_capture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(exe)))
def region_graph():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mi.lib.mi_dspu_graph_launch(exe, ctypes.c_void_p(st.cuda_stream))
    torch.cuda.synchronize(); return time.perf_counter() - t0
ts = sorted(region_graph() for _ in range(R))
print("one graph launch:   median region %.1f us = %.2f us per call (min %.1f)" % (ts[R // 2] * 1e6, ts[R // 2] * 1e6 / K, ts[0] * 1e6))

# ---- what the fence around a 20-call region costs: plain synchronize, a polling wait, and an uploaded graph -------------------
hip = ctypes.CDLL("libamdhip64.so")
S = ctypes.c_void_p(st.cuda_stream)
def region_poll():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mi.lib.mi_dspu_graph_launch(exe, S)
    while hip.hipStreamQuery(S) != 0:
        pass
    return time.perf_counter() - t0
ts = sorted(region_poll() for _ in range(R))
print("graph + polling wait: median region %.1f us = %.2f us per call (min %.1f)" % (ts[R // 2] * 1e6, ts[R // 2] * 1e6 / K, ts[0] * 1e6))
try:
    rc = hip.hipGraphUpload(exe, S)
    torch.cuda.synchronize()
    ts = sorted(region_graph() for _ in range(R))
    print("uploaded graph (rc %d): median region %.1f us = %.2f us per call (min %.1f)" % (rc, ts[R // 2] * 1e6, ts[R // 2] * 1e6 / K, ts[0] * 1e6))
    ts = sorted(region_poll() for _ in range(R))
    print("uploaded + polling:   median region %.1f us = %.2f us per call (min %.1f)" % (ts[R // 2] * 1e6, ts[R // 2] * 1e6 / K, ts[0] * 1e6))
except Exception as e:
    print("hipGraphUpload:", e)
