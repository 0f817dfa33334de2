"""C2 biquad: eager launches against a captured hipGraph of 16 calls (experiment)."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gpu = importlib.import_module("lsp-dsp-units_amd")
import workloads as wl
hip = ctypes.CDLL("libamdhip64.so")
C, N, R = 1024, 4096, 16
coef, _ = wl.c2_coefficients(C)
b = gpu.BiquadBank(C, 8)
b.set_all_chains(np.ascontiguousarray(coef[:, :8]))
s = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(s), 1); st = s.value
b.commit(st)
rng = np.random.default_rng(2)
ins = [gpu.DeviceBuffer.from_host((rng.standard_normal((C, N)) * 0.25).astype(np.float32), stream=st) for _ in range(R)]
outs = [gpu.DeviceBuffer((C, N)) for _ in range(R)]

def eager(reps):
    for _ in range(reps):
        for k in range(R):
            b.process(outs[k], ins[k], N, stream=st)
    hip.hipStreamSynchronize(s)

eager(5)
t0 = time.perf_counter(); eager(64); dt = time.perf_counter() - t0
print("eager : %.2f us per call" % (dt / (64 * R) * 1e6))
g, e = ctypes.c_void_p(), ctypes.c_void_p()
assert hip.hipStreamBeginCapture(s, 0) == 0
for k in range(R):
    b.process(outs[k], ins[k], N, stream=st)
assert hip.hipStreamEndCapture(s, ctypes.byref(g)) == 0
assert hip.hipGraphInstantiate(ctypes.byref(e), g, None, None, ctypes.c_size_t(0)) == 0
for _ in range(5):
    hip.hipGraphLaunch(e, s)
hip.hipStreamSynchronize(s)
t0 = time.perf_counter()
for _ in range(64):
    hip.hipGraphLaunch(e, s)
hip.hipStreamSynchronize(s)
dt = time.perf_counter() - t0
print("graph : %.2f us per call" % (dt / (64 * R) * 1e6))
b.close()
