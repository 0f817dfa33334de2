"""The lsp::dspu::* C++ compatibility classes: the reference's own unit tests replayed in C++ (tests/cpp)."""
import os
import re
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


def test_dump_keys_match_the_reference():
    """dump(): every class hands the IStateDumper the names the reference's dump() does, in the same order
    (tests/golden/dump_keys.json, made from the reference's sources by tests/golden/make_dump_keys.py; FilterBank builds its
    packed groups in a helper, so its names are compared as a set).  Static: the names are string literals on both sides."""
    import json
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_dump_keys
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dump_keys.json")))
    text = open(os.path.join(ROOT, "lsp-dsp-units_amd", "csrc", "host", "dspu_classes.cpp")).read()
    assert len(gold) == 15
    for cls, names in gold.items():
        body = re.sub(r"//.*", "", make_dump_keys.dump_body(text, cls))
        mine = re.findall(r'"(\w+)"', body)
        if cls == "FilterBank":
            assert set(mine) == set(names), cls
        else:
            assert mine == names, (cls, [(a, b) for a, b in zip(mine, names) if a != b][:3])


def test_state_dumper_interface_has_the_reference_slots():
    """The visitor's virtual methods, in the reference's order (include/lsp-plug.in/dsp-units/iface/IStateDumper.h:53-120):
    6 structure slots, 15 + 15 scalar slots, 14 + 14 vector slots -- counted in a compiled vtable."""
    src = r'''
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <cstdio>
struct probe: public lsp::dspu::IStateDumper
{
    int hits[4] = {0, 0, 0, 0};
    void write(const char *, float) override                { ++hits[0]; }
    void writev(const char *, const float *, size_t) override { ++hits[1]; }
    void begin_array(const char *, const void *, size_t) override { ++hits[2]; }
    void write(const char *, const void *) override         { ++hits[3]; }
};
struct unit { int x; void dump(lsp::dspu::IStateDumper *v) const { v->write("x", 1.5f); } };
int main()
{
    probe p;
    lsp::dspu::IStateDumper *v = &p;
    const float f[2] = {1, 2};
    unit u[2] = {{1}, {2}};
    const unit *none = nullptr;
    v->write("a", 1.0f); v->writev("b", f, 2); v->write_object_array("c", u, 2); v->write_object("d", none);
    float *tab[2] = {nullptr, nullptr};
    v->writev("e", tab, 2);
    std::printf("%d %d %d %d\n", p.hits[0], p.hits[1], p.hits[2], p.hits[3]);
    return 0;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        cpp = os.path.join(d, "probe.cpp")
        open(cpp, "w").write(src)
        exe = os.path.join(d, "probe")
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-Wno-unused-variable",
                               "-I" + os.path.join(ROOT, "lsp-dsp-units_amd", "include"), cpp, "-o", exe])
        assert subprocess.check_output([exe]).decode().split() == ["3", "1", "1", "1"]
    hdr = open(os.path.join(ROOT, "lsp-dsp-units_amd", "include", "lsp-plug.in", "dsp-units", "iface", "IStateDumper.h")).read()
    types = re.search(r"#define MI_DUMPER_TYPES\(X\)(.*?)\n\n", hdr, re.S).group(1)
    assert types.count("X(") == 13
