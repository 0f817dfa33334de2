"""Experiment: do two independent kernels captured as parallel branches of one hipGraph overlap, and what does the fork/join
cost?  A: the C2 biquad bank (arithmetic bound, 12.7 us); B: a second bank without sections (the same kernel as a copy,
bandwidth bound, 7 us).  20 iterations per graph: serial (A then B on one stream) against fork/join (B on a second stream)."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
hip = ctypes.CDLL("libamdhip64.so")
C, n, K, R = 1024, 4096, 20, 25
coef = mi.design_filter(47, 4, 3000.0, 3000.0, 1.0, 0.75)[2]
A = mi.BiquadBank(C, 8)
for c in range(C):
    A.set_chains(c, coef)
A.commit()
B = mi.BiquadBank(C, 8)
B.commit()
ring = 8
x = (torch.randn((ring, C, n)) * 0.25).cuda(); ya = torch.empty((ring, C, n), device="cuda"); yb = torch.empty((ring, C, n), device="cuda")
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream(); torch.cuda.set_stream(s1)
S1, S2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
f = mi.lib.mi_biquad_bank_process
def call(bank, y, i, st):
    mi.check(f(bank.handle, ctypes.c_void_p(y[i % ring].data_ptr()), ctypes.c_void_p(x[i % ring].data_ptr()), ctypes.c_size_t(n),
               ctypes.c_size_t(n), ctypes.c_size_t(n), st))
for i in range(4):
    call(A, ya, i, S1); call(B, yb, i, S1)
torch.cuda.synchronize()
ev = [ctypes.c_void_p() for _ in range(2 * K)]
for e in ev:
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0          # hipEventDisableTiming

def capture(mode):
    mi.check(mi.lib.mi_dspu_graph_begin_capture(S1))
    for i in range(K):
        if mode == "serial":
            call(A, ya, i, S1); call(B, yb, i, S1)
        elif mode == "only_a":
            call(A, ya, i, S1)
        elif mode == "only_b":
            call(B, yb, i, S1)
        else:
            assert hip.hipEventRecord(ev[2 * i], S1) == 0
            assert hip.hipStreamWaitEvent(S2, ev[2 * i], 0) == 0
            call(B, yb, i, S2)
            call(A, ya, i, S1)
            assert hip.hipEventRecord(ev[2 * i + 1], S2) == 0
            assert hip.hipStreamWaitEvent(S1, ev[2 * i + 1], 0) == 0
    exe = ctypes.c_void_p(); mi.check(mi.lib.mi_dspu_graph_end_capture(S1, ctypes.byref(exe)))
    return exe

def timed(exe):
    def region():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mi.lib.mi_dspu_graph_launch(exe, S1)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    for _ in range(3):
        region()
    ts = sorted(region() for _ in range(R))
    return ts[R // 2] * 1e6 / K, ts[0] * 1e6 / K

for mode in ("only_a", "only_b", "serial", "fork_join"):
    try:
        med, best = timed(capture(mode))
        print("%-10s %.2f us per iteration (best %.2f)" % (mode, med, best))
    except Exception as e:
        print(mode, "failed:", str(e)[:300])
