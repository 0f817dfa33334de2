"""Build-time checks of the generated gfx950 ISA (no GPU needed: hipcc cross-compiles).

The hand-over between the two roles of conv_step_kernel is only correct if every wave that stores the frame's image drains
its stores (s_waitcnt vmcnt(0)) before it arrives at the workgroup barrier behind which the `done` counter moves
(MI355X_MICROARCH.md, "Valid forms", condition 3).  The wait is written out as inline asm in frame_role(); this test reads
the compiler's output and fails if a scheduling change ever separates the three."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lsp-dsp-units_amd", "csrc")


def _isa(source, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(source) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-w",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(ROOT, "lsp-dsp-units_amd", "include"),
           "-S", "--offload-device-only", source, "-o", out]
    subprocess.check_call(cmd)
    return open(out).read().split("\n")


def _kernel_bodies(lines, name_part):
    bodies = {}
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.endswith(":") and name_part in l and l.startswith("_Z") and "@" not in l.split(":")[0]:
            name = l.split(":")[0]
            j = i + 1
            while j < len(lines) and ".amdhsa_kernel" not in lines[j] and not lines[j].startswith(".Lfunc_end"):
                j += 1
            bodies[name] = lines[i:j]
            i = j
        else:
            i += 1
    return bodies


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_conv_step_frame_role_drains_image_stores_before_the_counter(tmp_path):
    lines = _isa(os.path.join(CSRC, "convolver.hip"), tmp_path)
    bodies = _kernel_bodies(lines, "conv_step_kernel")
    assert len(bodies) >= 12, sorted(bodies)               # 6 transform sizes x {plain, non-temporal}
    for name, body in bodies.items():
        text = [l.strip() for l in body]
        # the counter: the only returning-free atomic add behind a barrier in the frame role is the LAST atomic add of the body
        adds = [i for i, l in enumerate(text) if l.startswith("global_atomic_add")]
        assert adds, name
        done_add = adds[-1]
        barrier = max(i for i in range(done_add) if text[i].startswith("s_barrier"))
        # last write-through store of the image before that barrier
        stores = [i for i in range(barrier) if text[i].startswith("buffer_store") and " sc1" in text[i]]
        assert stores, name
        last_store = stores[-1]
        window = text[last_store:barrier]
        # the hand-written wait (inline asm is bracketed by ;;#ASMSTART / ;;#ASMEND) sits between the two
        k = [i for i, l in enumerate(window) if l == ";;#ASMSTART"]
        assert k and any(window[i + 1].replace(" ", "") == "s_waitcntvmcnt(0)" for i in k), (name, window[-12:])
        # and nothing stores to memory between the wait and the barrier
        w = max(i for i in k if window[i + 1].replace(" ", "") == "s_waitcntvmcnt(0)")
        assert not [l for l in window[w:] if re.match(r"(buffer|global|flat)_store", l)], name
