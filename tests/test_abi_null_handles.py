"""Every mi_*_bank_* entry point called with a NULL bank (and zeros for everything else) answers with a negative MI_E*
code and a message -- no crash, no device needed.  Runs in a child process so that a crash is reported by name."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, importlib, sys
sys.path.insert(0, %r)
capi = importlib.import_module("lsp-dsp-units_amd.capi")
bad = []
for name, (res, args) in sorted(capi.PROTOTYPES.items()):
    if "_bank_" not in name or res is not ctypes.c_int:
        continue
    print("CALL", name, flush=True)
    zeros = []
    for a in args:
        if a in (ctypes.c_float, ctypes.c_double):
            zeros.append(0.0)
        elif a in (ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int64):
            zeros.append(0)
        else:
            zeros.append(None)
    code = getattr(capi.lib, name)(*zeros)
    msg = capi.lib.mi_dspu_last_error() or b""
    if name.endswith("_destroy"):
        ok = code <= 0                      # destroying nothing is allowed to succeed
    else:
        ok = code < 0 and len(msg) > 0
    if not ok:
        bad.append((name, code, msg))
print("DONE", bad, flush=True)
sys.exit(1 if bad else 0)
'''


def test_null_bank_is_refused_everywhere():
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600)
    calls = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("CALL")]
    assert len(calls) > 80, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, "last call: %s\n%s\n%s" % (calls[-1] if calls else None, r.stdout[-1500:], r.stderr[-1500:])
