"""PCIe-inclusive rate of the C2 biquad path: pinned host buffers in, pinned host buffers out.
Serial (copy in, filter, copy out on one stream) and pipelined over three streams.  Run on the GPU box."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mi = importlib.import_module("lsp-dsp-units_amd")
import workloads as wl

dev = torch.device("cuda", 0)
C, n, ring, steps = 1024, 4096, 4, 200
coef, _ = wl.c2_coefficients(C)
bank = mi.BiquadBank(C, 8)
bank.set_all_chains(np.ascontiguousarray(coef[:, :8]))
hin = [(torch.randn((C, n)) * 0.25).pin_memory() for _ in range(ring)]
hout = [torch.empty((C, n)).pin_memory() for _ in range(ring)]
din = [torch.empty((C, n), device=dev) for _ in range(ring)]
dout = [torch.empty((C, n), device=dev) for _ in range(ring)]
s = torch.cuda.Stream()
bank.commit(s)


def serial(count):
    with torch.cuda.stream(s):
        for i in range(count):
            k = i % ring
            din[k].copy_(hin[k], non_blocking=True)
            bank.process(dout[k], din[k], n, stream=s)
            hout[k].copy_(dout[k], non_blocking=True)
    s.synchronize()


def pipelined(count):
    up, down = torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in range(ring)]
    ev_done = [torch.cuda.Event() for _ in range(ring)]
    ev_out = [None] * ring
    for i in range(count):
        k = i % ring
        with torch.cuda.stream(up):
            if ev_out[k] is not None:
                up.wait_event(ev_done[k])               # the filter of the previous lap has read din[k]
            din[k].copy_(hin[k], non_blocking=True)
            ev_in[k].record(up)
        s.wait_event(ev_in[k])
        if ev_out[k] is not None:
            s.wait_event(ev_out[k])                     # the previous lap's copy out has read dout[k]
        bank.process(dout[k], din[k], n, stream=s)
        ev_done[k].record(s)
        with torch.cuda.stream(down):
            down.wait_event(ev_done[k])
            hout[k].copy_(dout[k], non_blocking=True)
            ev_out[k] = torch.cuda.Event()
            ev_out[k].record(down)
    torch.cuda.synchronize()


for name, fn in (("serial, one stream", serial), ("pipelined, three streams", pipelined)):
    fn(20)
    t0 = time.perf_counter()
    fn(steps)
    dt = time.perf_counter() - t0
    print("%-26s %8.1f Msamples/s  (%.1f us per 1024 x 4096 block, %.1f GB/s each way)"
          % (name, C * n * steps / dt / 1e6, dt / steps * 1e6, C * n * 4 * steps / dt / 1e9), flush=True)
assert bool(torch.isfinite(hout[0]).all())
bank.close()
