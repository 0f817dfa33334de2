"""biquad_wide_kernel against the oracle and against the four-wave stream kernel (MI_BIQUAD_NO_WIDE=1 in a child process is the
other side): C2 shape, 20 blocks in one mi_biquad_bank_process_blocks call; prints parity figures and the launch's duration.
usage: biquad_wide_check.py [blocks = 20]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib, ctypes
import numpy as np
import torch
mi = importlib.import_module("lsp-dsp-units_amd")
import oracle, workloads as wl

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C, n = 1024, 4096
coef, fc = wl.c2_coefficients(C)
x = wl.c2_input(C, n, blocks=nb)
dev = torch.device("cuda:0")
xin = torch.from_numpy(x).to(dev)
yout = torch.empty_like(xin)
bank = mi.BiquadBank(C, 8); bank.set_all_chains(coef)
st = torch.cuda.current_stream()
po = (ctypes.c_void_p * nb)(*[yout[k].data_ptr() for k in range(nb)])
pi = (ctypes.c_void_p * nb)(*[xin[k].data_ptr() for k in range(nb)])
sp = ctypes.c_void_p(st.cuda_stream)
mi.check(mi.lib.mi_biquad_bank_process_blocks(bank.handle, po, pi, nb, n, n, n, sp))
torch.cuda.synchronize()
y = yout.cpu().numpy()
state_gpu = bank.get_state()
# oracle
state = np.zeros((C, 8, 2), np.float32); nsec = np.full(C, 8, np.uint32)
y32 = np.empty_like(x)
for b in range(nb):
    y32[b] = oracle.biquad_bank(x[b], coef, nsec, state)
worst_r, worst_x = 0.0, 0.0
ratios = []
for c in range(0, C, 4):
    y64 = oracle.biquad_cascade_f64(x[:, c, :].reshape(-1), coef[c]).reshape(nb, n)
    peak = np.abs(y64).max()
    noise = np.abs(y32[:, c] - y64).max() / peak
    ex = np.abs(y[:, c] - y64).max() / peak
    rf = np.abs(y[:, c] - y32[:, c]).max() / peak
    if noise > 2e-6:
        ratios.append((ex / noise, rf / noise, fc[c]))
    else:
        worst_r = max(worst_r, rf)
r = np.array(ratios)
print("strict channels worst |gpu-oracle| %.2e; noisy: |gpu-exact|/noise max %.2f median %.2f; |gpu-oracle|/noise max %.2f" %
      (worst_r, r[:, 0].max(), np.median(r[:, 0]), r[:, 1].max()))
print("state vs oracle state: max abs diff %.3e (state peak %.3e)" % (np.abs(state_gpu - state).max(), np.abs(state).max()))
# timing
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(20):
        mi.check(mi.lib.mi_biquad_bank_process_blocks(bank.handle, po, pi, nb, n, n, n, sp))
    ev[0].record()
    for _ in range(50):
        mi.check(mi.lib.mi_biquad_bank_process_blocks(bank.handle, po, pi, nb, n, n, n, sp))
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    print("launch of %d blocks: %.1f us = %.2f us per block, frac %.3f" % (nb, us, us / nb, 8.0 * C * n * nb / (us * 1e-6) / 8e12))
bank.close()
