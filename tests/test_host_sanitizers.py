"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-only part of the product (the filter designer, windows,
envelopes and FFT-crossover curves of csrc/host): a CPU build of those sources with ROCm's clang, driven by
tests/sanitize/driver.py in a child process.  (GPU sanitizers are not available on the pool; the kernels are covered by
the differential tests instead.)"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HOST = ["filter_design.cpp", "windows.cpp", "fft_crossover.cpp", "filter_capi.cpp"]


def test_host_code_under_asan_and_ubsan(tmp_path):
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not (os.path.exists(CLANG) and rt):
        pytest.skip("ROCm clang or its sanitizer runtime is not installed")
    lib = str(tmp_path / "libmi_host_san.so")
    pkg = os.path.join(ROOT, "lsp-dsp-units_amd")
    cmd = [CLANG, "-x", "hip", "--offload-host-only", "-nogpulib", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-shared-libasan",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"), "-I" + os.path.join(pkg, "include")]
    cmd += [os.path.join(pkg, "csrc", "host", f) for f in HOST] + [os.path.join(ROOT, "tests", "sanitize", "fail_stub.cpp"), "-o", lib]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    # the harness itself first: a deliberate 56-float overrun of an 8-float buffer has to come back as a report
    self = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "driver.py"), lib, "--selftest"],
                          capture_output=True, text=True, env=env, timeout=300)
    assert self.returncode != 0 and "heap-buffer-overflow" in self.stderr, (self.returncode, self.stdout[-300:], self.stderr[-1500:])
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "driver.py"), lib], capture_output=True, text=True,
                         env=env, timeout=900)
    report = run.stderr[-4000:]
    assert run.returncode == 0 and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, report
    assert "no report" in run.stdout, run.stdout[-500:]
