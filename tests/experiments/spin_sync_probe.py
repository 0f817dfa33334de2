# Does the way the host waits in torch.cuda.synchronize() matter for a 20-step region?  hipDeviceScheduleSpin against the default.
#   python3 tests/experiments/spin_sync_probe.py [spin]
import ctypes, os, sys, time
spin = len(sys.argv) > 1 and sys.argv[1] == "spin"
import torch
if spin:
    hip = None
    for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), "libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name); break
        except OSError:
            pass
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))
x = torch.zeros(1 << 20, device="cuda")
def region(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        x.add_(1.0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for k in (1, 20, 200):
    ts = sorted(region(k) for _ in range(50))
    print("spin" if spin else "default", "k=%d" % k, "median region %.1f us, min %.1f us" % (ts[25] * 1e6, ts[0] * 1e6))
