"""The lsp::dspu::* C++ compatibility classes: the reference's own unit tests replayed in C++ (tests/cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "reference_utests")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


def test_cpp_replay_builds_and_lists():
    """No GPU needed: the public headers compile with plain g++ -std=c++11 and link against the library."""
    _build()
    out = subprocess.check_output([BIN, "--list"]).decode()
    assert "convolver.test_small" in out and "ringbuffer" in out


@pytest.mark.gpu
def test_cpp_replay_of_reference_utests(gpu):
    if not os.path.exists(BIN):
        _build()
    p = subprocess.run([BIN], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print(p.stdout.decode())
    assert p.returncode == 0, p.stdout.decode()
