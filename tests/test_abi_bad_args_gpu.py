"""Every entry point of every bank called on a live bank with zeros / NULL for all other arguments: MI_OK or a negative
code, never a crash; the bank can still be destroyed.  Child process, so that a crash is reported by name."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, importlib, sys
import numpy as np
sys.path.insert(0, %r)
gpu = importlib.import_module("lsp-dsp-units_amd")
capi = importlib.import_module("lsp-dsp-units_amd.capi")
C = 4
banks = {
    "mi_biquad_bank_": lambda: gpu.BiquadBank(C, 4),
    "mi_convolver_bank_": lambda: gpu.ConvolverBank(np.ones((C, 600), np.float32), 8),
    "mi_spectral_bank_": lambda: gpu.SpectralBank(C, 10),
    "mi_analyzer_bank_": lambda: gpu.AnalyzerBank(C, 10, 48000, 1.0, 0),
    "mi_delay_bank_": lambda: gpu.DelayBank(C, 1000),
    "mi_ring_bank_": lambda: gpu.RingBank(C, 1024),
    "mi_loudness_bank_": lambda: gpu.LoudnessBank(2, 2, 400.0),
    "mi_ilufs_bank_": lambda: gpu.ILUFSBank(2, 2, 10.0, 400.0),
    "mi_splitter_bank_": lambda: gpu.SplitterBank(C, 10, 2),
    "mi_crossover_bank_": lambda: gpu.CrossoverBank(C, 3),
    "mi_equalizer_bank_": lambda: gpu.EqualizerBank(C, 4, 9),
}
bad, n = [], 0
for prefix, make in banks.items():
    for name, (res, args) in sorted(capi.PROTOTYPES.items()):
        if not name.startswith(prefix) or name.endswith(("_create", "_destroy")) or res is not ctypes.c_int:
            continue
        b = make()                                   # a fresh bank for every call
        print("CALL", name, flush=True)
        zeros = [ctypes.c_void_p(b.handle.value if hasattr(b.handle, "value") else b.handle)]
        for a in args[1:]:
            if a in (ctypes.c_float, ctypes.c_double):
                zeros.append(0.0)
            elif a in (ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int64):
                zeros.append(0)
            else:
                zeros.append(None)
        code = getattr(capi.lib, name)(*zeros)
        n += 1
        if code > 0:
            bad.append((name, code))
        b.close()
print("DONE", n, bad, flush=True)
sys.exit(1 if bad else 0)
'''


def test_zero_arguments_on_live_banks_never_crash(gpu):
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=900)
    calls = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("CALL")]
    assert len(calls) > 80, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, "last call: %s\n%s\n%s" % (calls[-1] if calls else None, r.stdout[-1500:], r.stderr[-1500:])
