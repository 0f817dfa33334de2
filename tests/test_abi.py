"""The C-ABI library loads and exports exactly what include/mi_dspu.h declares (no GPU needed)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mi_dspu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(mi):
    names = _declared()
    assert len(names) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", mi.LIB_PATH]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in mi_dspu.h but not exported: %s" % missing


def test_binding_covers_header(mi):
    from importlib import import_module
    capi = import_module("lsp-dsp-units_amd.capi")
    names = set(_declared())
    assert names == set(capi.PROTOTYPES), (names ^ set(capi.PROTOTYPES))


def test_abi_version_and_error_string(mi):
    assert mi.lib.mi_dspu_abi_version() == 1
    assert isinstance(mi.lib.mi_dspu_last_error(), bytes)


def test_no_cpu_fallback(mi):
    """Without a device the product must refuse to compute, loudly."""
    if mi.device_count() > 0:
        return
    try:
        mi.BiquadBank(2, 8)
    except mi.MiError as e:
        assert e.code == -3
    else:
        raise AssertionError("BiquadBank was created without a HIP device")


def test_product_never_touches_oracle():
    """Nothing under lsp-dsp-units_amd/ or include/ may reference oracle/ (it is test infrastructure)."""
    bad = []
    for base in ("lsp-dsp-units_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in fn:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    s = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"\boracle\b|liborc", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_profile_summarizer_names_existing_kernels():
    """tests/prof_summarize.py attributes PMC traffic to kernels BY NAME: a renamed kernel would silently yield no traffic figure
    (round 2 renamed the convolver's dominant kernel once).  Every name it looks for must exist in the sources."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("prof_summarize", os.path.join(ROOT, "tests", "prof_summarize.py"))
    text = open(spec.origin).read()
    names = re.findall(r'"(\w+)":\s*"([\w<>, ]+)"', text[text.index("KERNELS"):text.index("KERNELS") + 400])
    assert names, "KERNELS table not found"
    src = "".join(open(os.path.join(ROOT, "lsp-dsp-units_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "lsp-dsp-units_amd", "csrc")) if f.endswith(".hip"))
    for workload, kernel in names:
        base = kernel.split("<")[0]
        assert re.search(r"\bvoid\s+" + base + r"\s*\(", src), (workload, kernel)
