"""hipGraph capture of a biquad bank call: which capture mode works (experiment)."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gpu = importlib.import_module("lsp-dsp-units_amd")
from oracle import filter_design as fd
import workloads as wl
hip = ctypes.CDLL("libamdhip64.so")
C, N = 64, 4096
q = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 3000.0, 0, 1.0, 0.75)
for mode in (0, 1, 2):
    s = ctypes.c_void_p()
    print("stream", hip.hipStreamCreateWithFlags(ctypes.byref(s), 1))
    b = gpu.BiquadBank(C, 8)
    for c in range(C):
        b.set_chains(c, q)
    b.commit(s.value)
    d = gpu.DeviceBuffer.from_host(np.ones((C, N), np.float32), stream=s.value); y = gpu.DeviceBuffer((C, N))
    b.process(y, d, N, stream=s.value)
    hip.hipStreamSynchronize(s)
    print("mode", mode, "begin", hip.hipStreamBeginCapture(s, mode))
    try:
        b.process(y, d, N, stream=s.value)
        print("  process ok")
    except Exception as e:
        print("  process failed:", str(e)[:200])
    g = ctypes.c_void_p()
    print("  end", hip.hipStreamEndCapture(s, ctypes.byref(g)), g.value)
    print("  last error", hip.hipGetLastError())
    b.close()
