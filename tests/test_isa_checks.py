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
        m = re.match(r"^(_Z\w+):", l)                       # "<mangled name>:  ; @<mangled name>"
        if m and name_part in m.group(1):
            name = m.group(1)
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
        text = [l.strip() for l in body if l.strip() and not l.strip().startswith(";") or l.strip().startswith(";;#ASM")]
        # the hand-written wait (inline asm is bracketed by ;;#ASMSTART / ;;#ASMEND)
        waits = [i for i, l in enumerate(text) if l == ";;#ASMSTART" and text[i + 1].replace(" ", "") == "s_waitcntvmcnt(0)"]
        # two hand-overs: the image (always), and the accumulator when the tail role folds into it (a frame received in blocks)
        assert len(waits) == 2, (name, len(waits))
        for w in waits:
            # behind it: the workgroup barrier, reached without another store to memory, and then the counter's atomic add
            labels = {l[:-1]: i for i, l in enumerate(text) if l.endswith(":")}
            path, i = [], w
            # (a one-wave workgroup, plan<7>, has no s_barrier at all: the wave's own wait orders its stores before its add)
            while not text[i].startswith("s_barrier") and not text[i].startswith("global_atomic_add"):
                path.append(text[i])                            # straight-line walk; an unconditional branch is followed
                i = labels[text[i].split()[1]] if text[i].startswith("s_branch") else i + 1
                assert i < len(text) and len(path) < 200, (name, path[-10:])
            nxt = i
            assert not [l for l in path if re.match(r"(buffer|global|flat)_store", l)], (name, path)
            after = text[nxt:nxt + 40]
            assert any(l.startswith("global_atomic_add") for l in after), (name, after)
            assert not any(l.startswith("s_barrier") for l in after[1:next(i for i, l in enumerate(after) if l.startswith("global_atomic_add"))]), name
            # in front of it: the write-through stores of the image, with no barrier between the last of them and the wait
            prev = max([i for i in range(w) if text[i].startswith("s_barrier")] or [0])
            stores = [l for l in text[prev:w] if l.startswith("buffer_store") and " sc1" in l]
            assert stores, (name, "no sc1 image store between the previous barrier and the wait")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_fft16_index_arithmetic_on_the_host(tmp_path):
    """fft16.h's phases (butterflies, exchange image addressing, twiddle table) run for all threads of a workgroup in lock step
    on the CPU, 1024 .. 8192 points, forward and inverse, against a double-precision DFT (tests/cpp/fft16_host.cpp)."""
    exe = os.path.join(str(tmp_path), "fft16_host")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-w", "-I" + CSRC,
                           os.path.join(ROOT, "tests", "cpp", "fft16_host.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stdout.decode()[-800:]
    assert out.stdout.count(b"max error") == 8


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_dynfilter_kernels_keep_the_cascades_in_registers(tmp_path):
    """The analog cascades t[3] / b[3] of a sample must live in registers: a selected destination pointer (written in the
    source, or made by the optimiser out of two mirrored branches) parks them in scratch memory, a round trip to memory per
    sample.  Neither the any-type kernels nor the per-type ones may have a private segment, nor call a function."""
    out = os.path.join(str(tmp_path), "dynfilter.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize", "-w",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(ROOT, "lsp-dsp-units_amd", "include"),
           "-S", "--offload-device-only", os.path.join(CSRC, "dynfilter.hip"), "-o", out]
    subprocess.check_call(cmd)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]
    seen = 0
    for block in meta:
        name = re.search(r"\.name: *(\S+)", block).group(1)
        if "dynfilter_kernel" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size: *(\d+)", block).group(1))
        m = re.search(r"ILi(\d+)ELj(\d+)E", name)          # dynfilter_kernel<NW, BASE>
        base = int(m.group(2))
        if base < 256:
            assert scratch == 0, name                        # any-type kernels (BASE 0) and the bilinear per-type kernels
        else:
            # matched-Z per-type kernels: the double-precision normalisation may spill a few registers at 256 VGPRs (and
            # says so in its spill count) -- the cascades themselves are 24 bytes of arrays and would show as such
            spills = int(re.search(r"\.vgpr_spill_count: *(\d+)", block).group(1))
            assert scratch <= 4 * spills + 4 and scratch <= 128, (name, scratch, spills)
    assert seen >= 3 + 27 + 27, seen                         # NW = 1, 2, 4 of the any-type kernel + one per base type and transform
    assert "s_swappc_b64" not in text
    # the Makefile builds this file without the SLP vectorizer (see the note there); the flag above must stay in step
    mk = open(os.path.join(ROOT, "lsp-dsp-units_amd", "Makefile")).read()
    assert "dynfilter.hip.o: HIPFLAGS += -fno-slp-vectorize" in mk


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_splitter_hop_kernels_with_one_handler_per_workgroup_fit_four_waves_per_simd(tmp_path):
    """1024 workgroups of 256 threads (256 channels x 4 bands, rank 12) are ONE round on the device only at four waves per
    SIMD, i.e. with at most 128 VGPRs; twice in round 3 a harmless-looking change of the source took the one-hop kernel from
    102 to 135 / 152 registers (30 -> 36 us per block).  No scratch either: the budget must be met without spilling."""
    text = "\n".join(_isa(os.path.join(CSRC, "splitter.hip"), tmp_path))
    seen = 0
    for block in text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name: *(\S+)", block).group(1)
        m = re.search(r"splitter_hop_kernelILi(\d+)ELb0ELb1ELi([01])E", name)
        if m is None or int(m.group(1)) > 12:
            continue
        seen += 1
        vgprs = int(re.search(r"\.vgpr_count: *(\d+)", block).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size: *(\d+)", block).group(1))
        assert vgprs <= 128 and scratch == 0, (name, vgprs, scratch)
    assert seen >= 14                                           # one-hop: 4 .. 12, several hops: the register path's sizes


def test_no_kernel_parks_data_in_scratch(tmp_path):
    """A private segment means round trips to memory in the middle of a kernel.  Twice it was not a register spill but a
    POINTER picked at run time into a small local (the cascades of dynfilter.hip; `&myblk.x / .y / .z / .w` in the integrated
    meter's bookkeeping): values are picked instead.  Every kernel of the library is held to no private segment at all, but for
    the known register spills: SpectralSplitter hops of 8192-point transforms (128-VGPR budget of their 1024 threads) and the
    matched-Z per-type kernels of DynamicFilters (checked in detail above)."""
    # (analyzer_kernel<13>: the one-strobe launch of 16384-point frames, 1024 threads at a 128-register budget, parks two registers
    # since the smoothing became three rounded operations -- mix2 -- in round 6; no measured row runs it)
    allowed = ("splitter_hop_kernelILi13E", "dynfilter_kernel", "analyzer_kernelILi13E")
    offenders = []
    for src in sorted(os.listdir(CSRC)):
        if not src.endswith(".hip"):
            continue
        out = os.path.join(str(tmp_path), src + ".s")
        extra = ["-fno-slp-vectorize"] if src == "dynfilter.hip" else []
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-w"] + extra +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(ROOT, "lsp-dsp-units_amd", "include"),
                               "-S", "--offload-device-only", os.path.join(CSRC, src), "-o", out])
        text = open(out).read()
        if "amdhsa.kernels:" not in text:
            continue
        for block in text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name: *(\S+)", block).group(1)
            scratch = int(re.search(r"\.private_segment_fixed_size: *(\d+)", block).group(1))
            if scratch > 0 and not any(a in name for a in allowed):
                offenders.append((src, name, scratch))
        # the streaming hop kernels read windows, gains and the callers' rows as GLOBAL memory: pointers that were laundered
        # against hoisting or came out of memory are generic to the compiler, and flat loads count against lgkmcnt as well
        # (every wait for LDS data then waits for them too)
        for part in ("stft_stream_kernel", "splitter_hop_kernel"):
            for name, body in _kernel_bodies(text.split("\n"), part).items():
                flat = sum(1 for l in body if l.strip().startswith("flat_"))
                if flat > 4:
                    offenders.append((src, name, "flat instructions", flat))
    assert not offenders, offenders


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_biquad_kernels_in_the_compilers_output(tmp_path):
    """Three properties of biquad.hip that only the generated code shows:
    * biquad_stream_kernel<4, true> (the headline launch) fits four waves per SIMD (at most 128 VGPRs), touches its hand-over cells
      with DS instructions (not flat ones: the cells are addressed through an LDS-qualified pointer) and forms its register
      pairs without the 64 moves per sub-block the padded tile cost (the interleaved tile: a 16-byte LDS read IS two pairs);
    * biquad_reference_ir_kernel (the impulse response the Equalizer's FIR is synthesised from) contains no fused
      multiply-add: every product and every sum of the reference's recurrence is rounded on its own;
    * biquad_sumsq_ilufs_kernel (the meter's bookkeeping riding on the filter launch) drains its additions -- the written-out
      s_waitcnt vmcnt(0) -- before the barrier behind which the row is counted in."""
    lines = _isa(os.path.join(CSRC, "biquad.hip"), tmp_path)
    text = "\n".join(lines)
    meta = {re.search(r"\.name: *(\S+)", b).group(1): b for b in text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]}
    # <4, true>: four waves per channel, the scan's per-lane operand in LDS (the form the C2 launch takes)
    stream4 = [n for n in meta if "biquad_stream_kernelILi4ELb1E" in n]
    assert len(stream4) == 1, sorted(meta)
    assert int(re.search(r"\.vgpr_count: *(\d+)", meta[stream4[0]]).group(1)) <= 128
    body = _kernel_bodies(lines, "biquad_stream_kernelILi4ELb1E")[stream4[0]]
    ops = [l.split()[0] for l in body if l.strip() and not l.strip().startswith((";", "."))]
    assert not any(o.startswith(("flat_", "scratch_")) for o in ops)
    assert sum(1 for o in ops if o == "ds_read_b32") >= 3 and sum(1 for o in ops if o == "ds_read2_b32") >= 8
    assert sum(1 for o in ops if o == "v_mov_b32_e32") <= 48, sum(1 for o in ops if o == "v_mov_b32_e32")
    # no vector-memory wait inside the sections (behind the hand-over's s_sleep), and the wait for the rows at the top of a
    # sub-block leaves the eight stores behind them in flight: vmcnt(15) .. vmcnt(8)
    stripped = [l.strip() for l in body]
    sleep_at = stripped.index("s_sleep 1")
    # (the kernel's last instructions stamp the shader's cycle counter -- mi_dspu_last_stream_clock -- behind a wait of their own:
    # the body ends at the last s_memtime)
    end_at = max(i for i, l in enumerate(stripped) if l.startswith("s_memtime"))
    waits = [(i, int(re.search(r"vmcnt\((\d+)\)", l).group(1))) for i, l in enumerate(stripped)
             if l.startswith("s_waitcnt") and "vmcnt" in l and i < end_at]
    assert not [w for w in waits if w[0] > sleep_at], waits
    assert [w[1] for w in waits][-8:] == list(range(15, 7, -1)), waits
    # the chain on a run of blocks: four waves per SIMD as well (the branch's copy and the prefetched rows never alive together)
    chain4 = [n for n in meta if "biquad_stream_chain_kernelILi4E" in n]
    assert len(chain4) == 2, sorted(meta)
    for n in chain4:
        assert int(re.search(r"\.vgpr_count: *(\d+)", meta[n]).group(1)) <= 128, n
        assert int(re.search(r"\.private_segment_fixed_size: *(\d+)", meta[n]).group(1)) == 0, n
    for name, b in _kernel_bodies(lines, "biquad_reference_ir_kernel").items():
        o2 = [l.split()[0] for l in b if l.strip() and not l.strip().startswith((";", "."))]
        assert not any(o.startswith(("v_fma", "v_fmac", "v_mac_f", "v_pk_fma", "v_mad_f", "v_mad_legacy", "v_mad_mix")) for o in o2), name
        assert sum(1 for o in o2 if o.startswith("v_mul_f32")) + 2 * sum(1 for o in o2 if o.startswith("v_pk_mul_f32")) >= 5
    riding = _kernel_bodies(lines, "biquad_sumsq_ilufs_kernel")
    assert riding
    for name, b in riding.items():
        t = [l.strip() for l in b if l.strip() and (not l.strip().startswith(";") or l.strip().startswith(";;#ASM"))]
        waits = [i for i, l in enumerate(t) if l == ";;#ASMSTART" and t[i + 1].replace(" ", "") == "s_waitcntvmcnt(0)"]
        assert waits, name
        w = waits[-1]
        adds = [i for i, l in enumerate(t) if l.startswith("global_atomic_add_u32") or l.startswith("global_atomic_add ")]
        assert adds and min(a for a in adds if a > w) > w, name
        between = t[w:min(a for a in adds if a > w)]
        assert any(l.startswith("s_barrier") for l in between), name
        assert not any(l.startswith(("global_atomic_add_f32", "buffer_store", "global_store")) for l in between), name


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_conv_batch_tail_queue_waits_for_its_own_requests_and_nothing_else(tmp_path):
    """conv_batch_tail_kernel asks for a partition's image and the frame the window takes in two steps ahead by LDS-DMA
    (global_load_lds_dwordx4, written as asm) and reads each lane's 16 bytes back from LDS.  Nothing orders that read behind
    the DMA except the issuing wave's counted vmcnt (MI355X_MICROARCH.md): every step must open with the hand-written
    `s_waitcnt vmcnt(2)` (the pair of two steps ago has landed, the pair of the step before may still be on its way), and the
    compiler must not have put a vmcnt wait of its own between the requests (the window's loads are waited for in front of
    the loop: a `vmcnt(0)` inside it is a memory round trip per step)."""
    lines = _isa(os.path.join(CSRC, "convolver.hip"), tmp_path)
    bodies = _kernel_bodies(lines, "conv_batch_tail_kernel")
    assert len(bodies) == 4, sorted(bodies)                 # K = 2, 4, 8, 16 (the staged frames kept in LDS: the re-reading form went in round 6)
    for name, body in bodies.items():
        text = [l.strip() for l in body if l.strip() and (not l.strip().startswith(";") or l.strip().startswith(";;#ASM"))]
        dma = [i for i, l in enumerate(text) if l.startswith("global_load_lds_dwordx4")]
        assert len(dma) >= 4, (name, len(dma))
        region = text[dma[0]:dma[-1] + 1]
        waits = [l for l in region if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert waits and all(w == "s_waitcnt vmcnt(2)" for w in waits), (name, sorted(set(waits)))
        # each step: the counted wait, then the two reads of the queue, then lgkmcnt(0) in front of the refill of the slot
        for i, l in enumerate(region):
            if l == "s_waitcnt vmcnt(2)":
                after = [x for x in region[i + 1:i + 24] if not x.startswith(";;#ASM")]
                refill = next(k for k, x in enumerate(after) if x.startswith("global_load_lds_dwordx4"))
                reads = [k for k, x in enumerate(after[:refill]) if x.startswith("ds_read_b128")]
                assert len(reads) >= 2 and "s_waitcnt lgkmcnt(0)" in after[reads[1]:refill], (name, after)
        # the asm's own vmcnt(0) behind the loop (the repeats past the last partition) sits outside the region by construction
