"""GPU parity of mi_convolver_bank_* (lsp::dspu::Convolver) against the CPU oracle, through the C-ABI.

The oracle restates the reference's non-uniform partitioned algorithm; the GPU uses a uniform partition
(DESIGN.md).  Both compute the same linear convolution, so they are compared sample by sample, relative to
the block peak, next to a float64 FFT convolution that tells how far float32 itself is off."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from conftest import TOL, note, parity_report, record_parity

pytestmark = pytest.mark.gpu


def exact_conv(x, ir):
    n = x.size + ir.size
    nfft = 1 << int(np.ceil(np.log2(n)))
    y = np.fft.irfft(np.fft.rfft(x.astype(np.float64), nfft) * np.fft.rfft(ir.astype(np.float64), nfft), nfft)
    return y[:x.size]


def run_gpu(gpu, irs, rank, x, chunks, counts=None, in_place=False):
    """x: [C][n]; chunks: list of call sizes covering n."""
    C, n = x.shape
    bank = gpu.ConvolverBank(irs, rank, counts=counts)
    y = np.empty_like(x)
    pos = 0
    for c in chunks:
        din = gpu.DeviceBuffer.from_host(x[:, pos:pos + c])
        dout = din if in_place else gpu.DeviceBuffer((C, c))
        bank.process(dout, din, c)
        y[:, pos:pos + c] = dout.download()
        pos += c
    assert pos == n
    info = bank.info()
    assert bank.faults() == 0                                # the one-launch frame step never gave up a hand-over
    bank.close()
    return y, info


def chunks_of(n, step):
    out = [step] * (n // step)
    if n % step:
        out.append(n % step)
    return out


def check(gpu_y, ref32, ref64, what, tol=TOL):
    r = parity_report(gpu_y, ref32, ref64)
    assert np.all(np.isfinite(gpu_y)), what
    record_parity("convolver: |gpu - oracle| <= max(tol, 3 noise)", r["gpu_vs_ref32"], max(tol, 3.0 * r["noise"]), noise=r["noise"], tol=tol)
    record_parity("convolver: |gpu - exact| <= max(tol, 3 noise)", r["gpu_vs_exact"], max(tol, 3.0 * r["noise"]), noise=r["noise"], tol=tol)
    assert r["gpu_vs_ref32"] <= max(tol, 3.0 * r["noise"]), "%s: %s" % (what, r)
    assert r["gpu_vs_exact"] <= max(tol, 3.0 * r["noise"]), "%s: %s" % (what, r)
    return r


def test_reference_utest_small(gpu):
    """src/test/utest/util/convolver.cpp:88-136 replayed through the GPU path."""
    conv = np.arange(1, 0x20, dtype=np.float32)
    src = np.zeros(0x2000 + conv.size, np.float32)
    for j, i in enumerate(range(0, 0x2000, 5)):
        src[i] = (1.0, 0.1, 0.01)[j % 3]
    ref = oracle.convolve(src, conv, 0x2000)[:src.size]
    y, info = run_gpu(gpu, conv, 9, src.reshape(1, -1), chunks_of(src.size, 31))
    assert info["rank"] == 9 and info["data_size"] == 31
    mask = np.abs(ref) > 1e-3
    assert np.all(np.abs(y[0][mask] - ref[mask]) <= 1e-4 * np.abs(ref[mask]))      # equals_relative 1e-4
    assert np.abs(y[0] - ref).max() <= 1e-4


def test_reference_utest_large(gpu):
    """convolver.cpp:184-223: rank 10, 0x2000 random taps, 0x20 random samples + zeros, 31-sample chunks."""
    rng = np.random.default_rng(1234)
    conv = rng.uniform(0.0, 1.0, 0x2000).astype(np.float32)
    src = np.zeros(0x20 + conv.size, np.float32)
    src[:0x20] = rng.uniform(0.0, 1.0, 0x20).astype(np.float32)
    ref = oracle.convolve(src, conv, 0x20)[:src.size]
    y, _ = run_gpu(gpu, conv, 10, src.reshape(1, -1), chunks_of(src.size, 31))
    assert np.abs(y[0] - ref).max() <= 1e-4                                          # equals_absolute 1e-4


def test_collisions_subset(gpu):
    """convolver.cpp:138-182 (disabled upstream): two unit impulses, 127-sample chunks, abs 1e-5."""
    rng = np.random.default_rng(7)
    conv = rng.uniform(-1.0, 1.0, 4096).astype(np.float32)
    for gap in (1, 127, 128, 129, 1000, 4095):
        src = np.zeros(4096 + conv.size, np.float32)
        src[0] = 1.0; src[gap] = 1.0
        ref = exact_conv(src, conv)
        y, _ = run_gpu(gpu, conv, 10, src.reshape(1, -1), chunks_of(src.size, 127))
        assert np.abs(y[0] - ref).max() <= 1e-5, gap


@pytest.mark.parametrize("rank", [8, 9, 10, 11, 12, 13, 14, 15, 16])        # every rank of util/Convolver.h:28-29
def test_whole_frames_every_rank(gpu, rank):
    rng = np.random.default_rng(rank)
    frame = 1 << (rank - 1)
    taps = 3 * frame + 17
    ir = (rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 3.0))).astype(np.float32)
    x = rng.standard_normal((1, 4 * frame)).astype(np.float32)
    y, info = run_gpu(gpu, ir, rank, x, chunks_of(x.shape[1], frame))
    assert info["rank"] == rank and info["frame"] == min(frame, 4096)
    o = oracle.Convolver(ir, rank)
    ref32 = np.concatenate([o.process(x[0, i:i + frame]) for i in range(0, x.shape[1], frame)])
    check(y[0], ref32, exact_conv(x[0], ir), "rank %d" % rank)


def test_mixed_call_sizes_multi_channel(gpu):
    """Arbitrary chunking across frame boundaries, distinct IRs and IR lengths per channel, in place."""
    rng = np.random.default_rng(21)
    C, rank, frame = 5, 10, 512
    counts = np.array([1, 100, 512, 513, 2500], np.uint32)
    irs = rng.standard_normal((C, 2500)).astype(np.float32)
    n = 6000
    x = rng.standard_normal((C, n)).astype(np.float32)
    chunks, left = [], n
    while left:
        c = int(min(left, rng.choice([1, 7, 31, 128, 500, 512, 1024, 1300])))
        chunks.append(c); left -= c
    y, info = run_gpu(gpu, irs, rank, x, chunks, counts=counts, in_place=True)
    assert info["partitions"] == 5
    for c in range(C):
        ir = irs[c, :counts[c]]
        ref32 = oracle.Convolver(ir, rank).process_chunked(x[c], 512)
        check(y[c], ref32, exact_conv(x[c], ir), "ch %d" % c)


def test_uninitialised_bank_outputs_zero(gpu):
    """Convolver::init(count = 0) leaves the object empty; process() writes zeros (Convolver.cpp:80-84,219-223)."""
    bank = gpu.ConvolverBank(np.zeros((2, 0), np.float32), 9)
    din = gpu.DeviceBuffer.from_host(np.ones((2, 100), np.float32))
    dout = gpu.DeviceBuffer.from_host(np.full((2, 100), 7.0, np.float32))
    bank.process(dout, din, 100)
    np.testing.assert_array_equal(dout.download(), np.zeros((2, 100), np.float32))
    assert bank.info()["rank"] == 0
    assert bank.faults() == 0                                # the one-launch frame step never gave up a hand-over
    bank.close()


def test_reset_and_linearity(gpu):
    rng = np.random.default_rng(9)
    ir = rng.standard_normal(3000).astype(np.float32)
    x = rng.standard_normal((1, 2048)).astype(np.float32)
    bank = gpu.ConvolverBank(ir, 10)
    din = gpu.DeviceBuffer.from_host(x)
    dout = gpu.DeviceBuffer((1, 2048))
    bank.process(dout, din, 2048)
    y1 = dout.download()
    bank.reset()
    bank.process(dout, din, 2048)
    np.testing.assert_array_equal(dout.download(), y1)            # same history -> same bits
    bank.reset()
    din.upload(x * np.float32(4.0))
    bank.process(dout, din, 2048)
    np.testing.assert_array_equal(dout.download(), y1 * np.float32(4.0))   # power-of-two scaling is exact
    assert bank.faults() == 0                                # the one-launch frame step never gave up a hand-over
    bank.close()


def test_c3_full_size(gpu):
    """BASELINE config 2 at full size: 256 channels, distinct 65536-tap IRs (N(0,1)*exp(-t/16384), seed 4), rank 13,
    SEVENTEEN 4096-sample frames (seed 5): the 15-slot ring of frame images is lapped once and wraps, so every partition
    of the conv_step_kernel<12> instantiation multiplies real history.  Every channel against the oracle (the reference's
    non-uniform algorithm, float32) and exact (float64) linear convolution over all frames, and once more against float64
    over the LAST frame alone (which contains the contributions of all sixteen partitions) with the strict 1e-5."""
    C, taps, frame, nf = 256, 65536, 4096, 17
    rng = np.random.default_rng(4)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    x = np.random.default_rng(5).standard_normal((C, nf * frame)).astype(np.float32)
    y, info = run_gpu(gpu, irs, 13, x, [frame] * nf)
    assert info == {"rank": 13, "frame": 4096, "partitions": 16, "data_size": 65536}
    picked = list(range(C))

    def ref(c):
        o = oracle.Convolver(irs[c], 13)
        return np.concatenate([o.process(x[c, i:i + frame]) for i in range(0, nf * frame, frame)])

    def exact_last(c):                                           # float64 linear convolution, its last frame
        return exact_conv(x[c], irs[c])[(nf - 1) * frame:]
    with ThreadPoolExecutor(max_workers=8) as ex:
        refs = dict(zip(picked, ex.map(ref, picked)))
        last = list(ex.map(exact_last, range(C)))
    worst, worst_last = 0.0, 0.0
    for c in picked:
        r = check(y[c], refs[c], exact_conv(x[c], irs[c]), "C3 ch %d" % c)
        worst = max(worst, r["gpu_vs_ref32"])
    for c in range(C):
        peak = float(np.abs(last[c]).max())
        err = float(np.abs(y[c, (nf - 1) * frame:] - last[c]).max()) / peak
        record_parity("convolver C3, frame 17 of every channel: |gpu - exact| <= 1e-5 peak", err, TOL)
        assert err <= TOL, (c, err)
        worst_last = max(worst_last, err)
    print("C3 full size, 17 frames, 256 channels: worst |gpu - oracle| / peak = %.2e, worst |gpu - exact| / peak on the last "
          "frame = %.2e" % (worst, worst_last))



def test_c3_full_size_process_blocks(gpu):
    """BASELINE config 2 at full size through the call bench.py's C3 `value` is measured on: 256 channels, distinct 65536-tap
    IRs, rank 13, THIRTY-THREE 4096-sample frames as ONE mi_convolver_bank_process_blocks call = two batches of sixteen
    frames and a single one.  With 256 channels a batch takes the `direct` branch of launch_batch (one run per channel: the
    frames kernel leaves the accumulator in place and zeroes its upper half itself, no finish launch) that no smaller test
    reaches.  Every channel against the oracle (the reference's non-uniform algorithm, float32) and exact (float64) linear
    convolution over all frames; frame 33 of every channel -- all sixteen partitions' contributions, three laps of batches
    behind it -- against float64 with the strict 1e-5; and, bit for bit, against a twin bank stepped frame by frame
    (conv_step_kernel: what test_c3_full_size holds against the oracle)."""
    C, taps, frame, nf = 256, 65536, 4096, 33
    rng = np.random.default_rng(4)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    x = np.random.default_rng(55).standard_normal((C, nf * frame)).astype(np.float32)
    bank, twin = gpu.ConvolverBank(irs, 13), gpu.ConvolverBank(irs, 13)
    ins = [gpu.DeviceBuffer.from_host(np.ascontiguousarray(x[:, f * frame:(f + 1) * frame])) for f in range(nf)]
    outs = [gpu.DeviceBuffer((C, frame)) for _ in range(nf)]
    touts = [gpu.DeviceBuffer((C, frame)) for _ in range(nf)]
    bank.process_blocks(outs, ins, frame)
    for f in range(nf):
        twin.process(touts[f], ins[f], frame)
    assert bank.faults() == 0 and twin.faults() == 0
    y = np.concatenate([o.download() for o in outs], axis=1)
    yt = np.concatenate([o.download() for o in touts], axis=1)
    bank.close(); twin.close()
    np.testing.assert_array_equal(y, yt)                       # the batches' sums are the frame step's, at BASELINE size

    def ref(c):
        o = oracle.Convolver(irs[c], 13)
        return np.concatenate([o.process(x[c, i:i + frame]) for i in range(0, nf * frame, frame)])

    def exact(c):
        return exact_conv(x[c], irs[c])
    with ThreadPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(ref, range(C)))
        exacts = list(ex.map(exact, range(C)))
    worst, worst_last = 0.0, 0.0
    for c in range(C):
        r = check(y[c], refs[c], exacts[c], "C3 batches ch %d" % c)
        worst = max(worst, r["gpu_vs_ref32"])
        last = exacts[c][(nf - 1) * frame:]
        err = float(np.abs(y[c, (nf - 1) * frame:] - last).max()) / float(np.abs(last).max())
        record_parity("convolver C3 batches, frame 33 of every channel: |gpu - exact| <= 1e-5 peak", err, TOL)
        assert err <= TOL, (c, err)
        worst_last = max(worst_last, err)
    note("C3 full size as ONE process_blocks call (16 + 16 + 1 frames, 256 channels, direct branch): worst |gpu - oracle| / peak = "
         "%.2e, worst |gpu - exact| / peak on frame 33 = %.2e; bit-identical to 33 process() calls" % (worst, worst_last))


_DIRECT_VS_FINISH = r"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
mi = importlib.import_module("lsp-dsp-units_amd")
C, rank, taps, nf = 130, 10, 2000, 21
frame = 1 << (rank - 1)
rng = np.random.default_rng(130)
irs = (rng.standard_normal((C, taps)) * 0.05).astype(np.float32)
x = (rng.standard_normal((nf + 2, C, frame)) * 0.25).astype(np.float32)
bank = mi.ConvolverBank(irs, rank)
ins = [mi.DeviceBuffer.from_host(x[k]) for k in range(nf + 2)]
outs = [mi.DeviceBuffer((C, frame)) for _ in range(nf + 2)]
bank.process(outs[0], ins[0], 100, frame, frame)            # a frame in pieces in front: the accumulator's upper half is in use
bank.process(outs[0].ptr + 400, ins[0].ptr + 400, frame - 100, frame, frame)
bank.process_blocks(outs[1:nf + 1], ins[1:nf + 1], frame)   # 16 + 4 + 1
bank.process(outs[nf + 1], ins[nf + 1], frame)              # a frame on what the batches left
assert bank.faults() == 0
np.save(sys.argv[2], np.stack([o.download() for o in outs]))
bank.close()
"""


def test_process_blocks_direct_branch_equals_the_finish_launch(gpu, tmp_path):
    """launch_batch with one run per channel (more than 128 channels) writes the accumulator a batch leaves straight into the
    bank's (`direct`), with fewer channels a finish launch does (conv_batch_finish_kernel; MI_DSPU_TEST_PATH=conv_batch_finish forces it).  130
    channels at rank 10, a frame in pieces in front (the accumulator's upper half is live when the first batch starts), 21 frames
    as batches of 16 + 4 + 1, a frame behind: both branches bit for bit, and against frame-by-frame calls and the oracle.  (The
    knob is read once per process: each variant runs in a process of its own.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, env in (("direct", {}), ("finish", {"MI_DSPU_TEST_PATH": "conv_batch_finish"}), ("calls", {"MI_DSPU_TEST_PATH": "conv_frame_per_launch"})):
        out = str(tmp_path / (name + ".npy"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", _DIRECT_VS_FINISH, root, out], check=True, env=e, timeout=600)
        got[name] = np.load(out)
    np.testing.assert_array_equal(got["direct"], got["finish"])
    np.testing.assert_array_equal(got["direct"], got["calls"])
    C, rank, taps, nf = 130, 10, 2000, 21
    frame = 1 << (rank - 1)
    rng = np.random.default_rng(130)
    irs = (rng.standard_normal((C, taps)) * 0.05).astype(np.float32)
    x = (rng.standard_normal((nf + 2, C, frame)) * 0.25).astype(np.float32)
    for ch in (0, 64, 129):
        xs = x[:, ch, :].reshape(-1)
        ref32 = oracle.Convolver(irs[ch], rank).process_chunked(xs, frame)
        check(got["direct"][:, ch, :].reshape(-1), ref32, exact_conv(xs, irs[ch]), "direct branch ch %d" % ch)


def test_one_launch_step_while_another_stream_keeps_every_cu_busy(gpu):
    """The whole-frame step is ONE launch of two roles that wait for each other inside the launch (conv_step_kernel: the tail
    role of a channel waits for its frame role's image).  That is safe as long as the frame workgroups are placed -- they
    never wait -- which index-ordered dispatch gives on an idle device.  Here a second stream keeps every CU occupied with
    long calls of a 1024-channel biquad bank (four waves per SIMD each, launched ahead and in between) while 256-channel,
    16-partition frames are stepped: the hand-over must neither give up (faults() == 0) nor change a bit of the result
    against the same frames stepped on an idle device, and the result is the oracle's."""
    import ctypes
    import workloads as wl
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        streams.append(s)
    sa, sb = streams[0].value, streams[1].value
    C, taps, rank, frame, nf = 256, 16 * 1024, 11, 1024, 20
    rng = np.random.default_rng(77)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 4096.0)).astype(np.float32)
    x = rng.standard_normal((C, nf * frame)).astype(np.float32)
    quiet, info = run_gpu(gpu, irs, rank, x, [frame] * nf)
    assert info["partitions"] == 16
    # the load: 1024 channels x 131072 samples per call, about a quarter of a millisecond of every SIMD each
    BC, bn = 1024, 131072
    coef, _ = wl.c2_coefficients(BC)
    noise = gpu.BiquadBank(BC, 8)
    noise.set_all_chains(coef)
    nin = gpu.DeviceBuffer.from_host((rng.standard_normal((BC, bn)) * 0.1).astype(np.float32), stream=sb)
    nout = gpu.DeviceBuffer((BC, bn))
    noise.commit(sb)
    bank = gpu.ConvolverBank(irs, rank)
    dins = [gpu.DeviceBuffer.from_host(x[:, f * frame:(f + 1) * frame], stream=sa) for f in range(nf)]
    douts = [gpu.DeviceBuffer((C, frame)) for _ in range(nf)]
    assert hip.hipStreamSynchronize(streams[0]) == 0 and hip.hipStreamSynchronize(streams[1]) == 0
    for f in range(nf):
        for _ in range(2):                                   # keep the other stream's queue full
            noise.process(nout, nin, bn, stream=sb)
        bank.process(douts[f], dins[f], frame, stream=sa)
    assert hip.hipStreamSynchronize(streams[0]) == 0 and hip.hipStreamSynchronize(streams[1]) == 0
    assert bank.faults(stream=sa) == 0
    busy = np.concatenate([d.download(stream=sa) for d in douts], axis=1)
    np.testing.assert_array_equal(busy, quiet)
    for c in (0, 97, 255):
        o = oracle.Convolver(irs[c], rank)
        ref = np.concatenate([o.process(x[c, i:i + frame]) for i in range(0, nf * frame, frame)])
        check(busy[c], ref, exact_conv(x[c], irs[c]), "contended step ch %d" % c)
    bank.close()
    noise.close()
    for s in streams:
        hip.hipStreamDestroy(s)


@pytest.mark.parametrize("seed", range(10))
def test_random_geometry_and_call_sizes(gpu, seed):
    """Differential stress against the exact (float64) linear convolution: random ranks, tap counts around the partition
    size, per-channel counts, in-place or not, resets, and call sizes from one sample to several frames in random order."""
    rng = np.random.default_rng(9000 + seed)
    C = 3
    rank = int(rng.choice([8, 9, 10]))
    frame = 1 << (rank - 1)
    taps = int(rng.choice([1, 31, frame - 1, frame, frame + 1, 2 * frame, 3 * frame + 17]))
    counts = np.array([taps] + [int(rng.integers(1, taps + 1)) for _ in range(C - 1)], np.uint32)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / (0.3 * taps + 1))).astype(np.float32)
    bank = gpu.ConvolverBank(irs, rank, counts=counts)
    hist = [np.zeros(0, np.float32) for _ in range(C)]
    for step in range(40):
        if rng.integers(0, 12) == 0:
            bank.reset()
            hist = [np.zeros(0, np.float32) for _ in range(C)]
            continue
        k = int(rng.choice([1, 2, 127, 128, 129, frame - 1, frame, frame + 1, 2 * frame, int(rng.integers(1, 4 * frame))]))
        x = rng.standard_normal((C, k)).astype(np.float32)
        din = gpu.DeviceBuffer.from_host(x)
        dout = din if rng.integers(0, 2) else gpu.DeviceBuffer((C, k))
        bank.process(dout, din, k)
        y = dout.download()
        for c in range(C):
            hist[c] = np.concatenate([hist[c], x[c]])[-(4 * frame + taps + k):]       # enough history for the tail
            ref = exact_conv(hist[c], irs[c, :counts[c]])[-k:]
            peak = max(float(np.abs(ref).max()), 1.0)
            err = float(np.abs(y[c] - ref).max())
            assert err <= 2 * TOL * peak, (seed, step, c, k, rank, taps, int(counts[c]), err / peak)
    assert bank.faults() == 0                                # the one-launch frame step never gave up a hand-over
    bank.close()


@pytest.mark.parametrize("seed", range(8))
def test_subframe_calls_on_large_frames(gpu, seed):
    """Sub-frame calls of partitioned banks with frames of 1024 samples and more go through the small-block delay line
    (conv_small_kernel: 256-sample blocks inside the frame, the frame's spill settled at its completion), whole frames through
    the frame step; both mixed at random -- aligned blocks, ragged pieces, pieces across block and frame boundaries, whole
    frames in between, resets -- against the exact (float64) linear convolution of the channel's history."""
    rng = np.random.default_rng(9100 + seed)
    C = 3
    rank = int(rng.choice([11, 12, 13]))
    frame = 1 << (rank - 1)
    taps = int(rng.choice([frame + 1, 2 * frame, 3 * frame + 17, 5 * frame]))
    counts = np.array([taps] + [int(rng.integers(frame + 1, taps + 1)) for _ in range(C - 1)], np.uint32)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / (0.3 * taps + 1))).astype(np.float32)
    bank = gpu.ConvolverBank(irs, rank, counts=counts)
    hist = [np.zeros(0, np.float32) for _ in range(C)]
    sizes = [1, 2, 100, 255, 256, 257, 511, 512, 768, 1000, frame - 256, frame - 1, frame, frame + 1, frame + 256, 2 * frame]
    for step in range(36):
        if rng.integers(0, 15) == 0:
            bank.reset()
            hist = [np.zeros(0, np.float32) for _ in range(C)]
            continue
        k = int(rng.choice(sizes + [int(rng.integers(1, 3 * frame))]))
        x = rng.standard_normal((C, k)).astype(np.float32)
        din = gpu.DeviceBuffer.from_host(x)
        dout = din if rng.integers(0, 2) else gpu.DeviceBuffer((C, k))
        bank.process(dout, din, k)
        y = dout.download()
        for c in range(C):
            hist[c] = np.concatenate([hist[c], x[c]])[-(taps + 4 * frame + k):]
            full = exact_conv(hist[c], irs[c, :counts[c]])
            ref = full[-k:]
            # the peak of the channel's recent output, not of this call alone: a call of one or two samples can land on a
            # zero crossing of a signal whose float32 partition sums carry the rounding of its LEVEL (seeds 33045, 33204 of
            # tests/experiments/stress_sweep.py: 2.0e-5 and 3.1e-5 of a single sample's own magnitude, with every path)
            peak = max(float(np.abs(full[-max(k, 1024):]).max()), 1.0)
            err = float(np.abs(y[c] - ref).max())
            record_parity("convolver sub-frame calls: |gpu - exact| <= 2e-5 peak", err, 2 * TOL * peak)
            assert err <= 2 * TOL * peak, (seed, step, c, k, rank, taps, int(counts[c]), err / peak)
    assert bank.faults() == 0
    bank.close()


@pytest.mark.parametrize("seed", range(10))
def test_subframe_calls_on_small_frames(gpu, seed):
    """Ranks 9 and 10 (frames of 256 / 512 samples; the reference runs its doubling levels there, Convolver.cpp:251-262):
    sub-frame calls of partitioned banks go through the small-block delay line with blocks of HALF a frame (128 / 256
    samples) -- aligned blocks, ragged pieces, pieces across block and frame boundaries, whole frames in between, resets --
    against the exact (float64) linear convolution of the channel's history."""
    rng = np.random.default_rng(9300 + seed)
    C = 3
    rank = 9 + seed % 2
    frame = 1 << (rank - 1)
    half = frame // 2
    taps = int(rng.choice([frame + 1, 2 * frame, 3 * frame + 17, 9 * frame - 5]))
    counts = np.array([taps] + [int(rng.integers(frame + 1, taps + 1)) for _ in range(C - 1)], np.uint32)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / (0.3 * taps + 1))).astype(np.float32)
    bank = gpu.ConvolverBank(irs, rank, counts=counts)
    hist = [np.zeros(0, np.float32) for _ in range(C)]
    sizes = [1, 2, 31, half - 1, half, half, half, half + 1, frame - 1, frame, frame + 1, frame + half, 2 * frame]
    for step in range(48):
        if rng.integers(0, 16) == 0:
            bank.reset()
            hist = [np.zeros(0, np.float32) for _ in range(C)]
            continue
        k = int(rng.choice(sizes + [int(rng.integers(1, 3 * frame))]))
        x = rng.standard_normal((C, k)).astype(np.float32)
        din = gpu.DeviceBuffer.from_host(x)
        dout = din if rng.integers(0, 2) else gpu.DeviceBuffer((C, k))
        bank.process(dout, din, k)
        y = dout.download()
        for c in range(C):
            hist[c] = np.concatenate([hist[c], x[c]])[-(taps + 4 * frame + k):]
            full = exact_conv(hist[c], irs[c, :counts[c]])
            ref = full[-k:]
            peak = max(float(np.abs(full[-max(k, 1024):]).max()), 1.0)
            err = float(np.abs(y[c] - ref).max())
            record_parity("convolver sub-frame calls: |gpu - exact| <= 2e-5 peak", err, 2 * TOL * peak)
            assert err <= 2 * TOL * peak, (seed, step, c, k, rank, taps, int(counts[c]), err / peak)
    assert bank.faults() == 0
    bank.close()


@pytest.mark.parametrize("rank", [9, 10])
def test_stream_of_half_frame_calls_small_ranks(gpu, rank):
    """A host that calls with half frames at ranks 9 / 10: every call is one aligned small block in the
    frequency domain (conv_small_kernel, one launch), every second one completes the frame (the one-launch frame step takes
    its image and the tail); against the oracle's non-uniform partitioning and the exact convolution, and against the same
    input given as whole frames."""
    rng = np.random.default_rng(780 + rank)
    frame = 1 << (rank - 1)
    C, taps, nf = 8, 6 * frame + 11, 12
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / (2.0 * frame))).astype(np.float32)
    x = rng.standard_normal((C, nf * frame)).astype(np.float32)
    bank = gpu.ConvolverBank(irs, rank)
    y = np.empty_like(x)
    seen = set()
    for i in range(2 * nf):
        din = gpu.DeviceBuffer.from_host(x[:, i * frame // 2:(i + 1) * frame // 2])
        dout = gpu.DeviceBuffer((C, frame // 2))
        bank.process(dout, din, frame // 2)
        seen.add(gpu.last_launch().split("<")[0].strip("("))          # (the last launch of the call)
        y[:, i * frame // 2:(i + 1) * frame // 2] = dout.download()
    assert bank.faults() == 0
    bank.close()
    assert seen <= {"conv_small_kernel", "conv_step_kernel"}, seen          # (launches that note themselves: no time-domain head)
    whole, _ = run_gpu(gpu, irs, rank, x, [frame] * nf)
    for c in range(C):
        ex = exact_conv(x[c], irs[c])
        peak = float(np.abs(ex).max())
        err = float(np.abs(y[c] - ex).max()) / peak
        record_parity("convolver, half-frame calls at ranks 9-10: |gpu - exact| <= 1e-5 peak", err, TOL)
        assert err <= TOL, (c, err)
        assert float(np.abs(y[c] - whole[c]).max()) / peak <= TOL
    ref = oracle.Convolver(irs[0], rank).process_chunked(x[0], frame // 2)
    check(y[0], ref, exact_conv(x[0], irs[0]), "channel 0, half-frame calls at rank %d" % rank)


def test_stream_of_256_sample_calls_c3_shape(gpu):
    """What a plugin host does: 256-sample calls, for ever.  16 channels of the C3 shape (65536 taps, rank 13), five frames'
    worth of calls: every call is one aligned small block (one launch), every sixteenth completes a frame."""
    rng = np.random.default_rng(77)
    C, taps, frame, nf = 16, 65536, 4096, 5
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 16384.0)).astype(np.float32)
    x = rng.standard_normal((C, nf * frame)).astype(np.float32)
    y, info = run_gpu(gpu, irs, 13, x, [256] * (nf * frame // 256))
    assert info["frame"] == 4096 and info["partitions"] == 16
    for c in range(C):
        ex = exact_conv(x[c], irs[c])
        err = float(np.abs(y[c] - ex).max()) / float(np.abs(ex).max())
        record_parity("convolver, stream of 256-sample calls: |gpu - exact| <= 1e-5 peak", err, TOL)
        assert err <= TOL, (c, err)
    ref = oracle.Convolver(irs[0], 13).process_chunked(x[0], 256)
    check(y[0], ref, exact_conv(x[0], irs[0]), "channel 0, 256-sample calls")


def test_more_channels_than_compute_units(gpu):
    """The one-launch frame step pairs a frame and a tail workgroup per CU; a bank with more channels than the device has
    CUs goes in several such launches per frame.  300 channels at rank 13 (distinct three-partition responses), whole frames
    and a ragged tail; channels on both sides of the launch boundary against exact float64 convolution and, a few, the oracle."""
    rng = np.random.default_rng(31)
    C, taps, n = 300, 9000, 3 * 4096 + 777
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 3000.0)).astype(np.float32)
    x = (rng.standard_normal((C, n)) * 0.3).astype(np.float32)
    y, info = run_gpu(gpu, irs, 13, x, [4096, 4096, 4096, 777])
    assert info["frame"] == 4096 and info["partitions"] == 3
    for c in (0, 1, 128, 254, 255, 256, 257, 299):
        ex = exact_conv(x[c], irs[c])
        err = float(np.abs(y[c] - ex).max()) / float(np.abs(ex).max())
        assert err <= TOL, (c, err)
    for c in (255, 256):
        ref = oracle.Convolver(irs[c], 13).process_chunked(x[c], 4096)
        check(y[c], ref, exact_conv(x[c], irs[c]), "channel %d" % c)


def test_block_stream_with_more_channels_than_compute_units(gpu):
    """The completion of a frame received in blocks is the one-launch step with the tail role folding into the accumulator:
    300 channels at rank 13 go in two such launches per frame (one CU-count of channels each).  256-sample calls over
    two and a half frames, then a ragged piece; channels on both sides of the launch boundary against float64."""
    rng = np.random.default_rng(32)
    C, taps = 300, 9000
    calls = [256] * 40 + [100]
    n = sum(calls)
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 3000.0)).astype(np.float32)
    x = (rng.standard_normal((C, n)) * 0.3).astype(np.float32)
    y, info = run_gpu(gpu, irs, 13, x, calls)
    assert info["frame"] == 4096 and info["partitions"] == 3
    for c in (0, 1, 127, 254, 255, 256, 257, 299):
        ex = exact_conv(x[c], irs[c])
        err = float(np.abs(y[c] - ex).max()) / float(np.abs(ex).max())
        record_parity("convolver, stream of 256-sample calls: |gpu - exact| <= 1e-5 peak", err, TOL)
        assert err <= TOL, (c, err)


def test_more_workgroups_than_the_device_holds(gpu):
    """6000 channels at rank 9: the one-launch frame step is a grid of 12 000 single-wave workgroups, more than can be resident
    at once, so tail workgroups are dispatched while frame workgroups are still queued behind others -- the hand-over must not
    depend on everybody being resident (frame workgroups come first in every XCD's share of the grid).  Whole frames and a
    ragged tail; channels spread over the grid against exact float64 convolution; no hand-over may have timed out."""
    rng = np.random.default_rng(33)
    C, rank, frame = 6000, 9, 256
    taps = 3 * frame + 5
    irs = (rng.standard_normal((C, taps)) * np.exp(-np.arange(taps) / 300.0)).astype(np.float32)
    x = (rng.standard_normal((C, 5 * frame + 100)) * 0.3).astype(np.float32)
    y, info = run_gpu(gpu, irs, rank, x, [frame, 2 * frame, frame, frame, 100])
    assert info["frame"] == frame and info["partitions"] == 4
    for c in list(range(0, C, 509)) + [C - 1]:
        ex = exact_conv(x[c], irs[c])
        err = float(np.abs(y[c] - ex).max()) / float(np.abs(ex).max())
        assert err <= TOL, (c, err)


@pytest.mark.parametrize("rank,taps", [(10, 2500), (12, 9000)])
def test_sub_frame_calls_whose_rows_overlap_or_lie_apart(gpu, rank, taps):
    """A sub-frame call reads the caller's rows itself when its output lies apart from them (one launch) and through a copy
    in the frame otherwise: the output rows shifted DOWN against the input rows by one row and by a few samples (what a
    block-by-block walk like Convolver.cpp:217-313 survives: nothing is written that is still to be read), in place, and
    apart -- the same samples every time."""
    rng = np.random.default_rng(77 + rank)
    C, n, stride = 6, 3000, 3400
    irs = rng.standard_normal((C, taps)).astype(np.float32) * 0.05
    x = rng.standard_normal((C, n)).astype(np.float32)
    chunks, left = [], n
    while left:
        c = int(min(left, rng.choice([1, 40, 64, 128, 200, 256, 300])))
        chunks.append(c); left -= c
    want = None
    for mode in ("apart", "in place", "one row down", "five samples down"):
        bank = gpu.ConvolverBank(irs, rank)
        arena = gpu.DeviceBuffer((2 * C + 2, stride))
        y = np.empty_like(x)
        pos = 0
        for c in chunks:
            host = np.zeros((2 * C + 2, stride), np.float32)
            host[1:C + 1, :c] = x[:, pos:pos + c]
            arena.upload(host)
            src = arena.ptr + 4 * stride
            dst = {"apart": arena.ptr + 4 * stride * (C + 1), "in place": src, "one row down": arena.ptr,
                   "five samples down": src - 20}[mode]
            bank.process(dst, src, c, stride, stride)
            got = arena.download().reshape(-1)
            o0 = (dst - arena.ptr) // 4
            y[:, pos:pos + c] = np.stack([got[o0 + r * stride:o0 + r * stride + c] for r in range(C)])
            pos += c
        assert bank.faults() == 0
        bank.close()
        arena.free()
        if want is None:
            want = y
            for ch in (0, C - 1):
                ref32 = oracle.Convolver(irs[ch], rank).process_chunked(x[ch], 1 << (rank - 1))
                check(y[ch], ref32, exact_conv(x[ch], irs[ch]), "rank %d ch %d" % (rank, ch))
        else:
            assert np.array_equal(y, want), mode


@pytest.mark.parametrize("rank,taps,K", [(10, 2500, 16), (10, 2500, 7), (11, 5000, 37), (12, 9000, 4), (13, 20000, 6), (10, 1024, 5), (10, 3 * 512 + 1, 21), (12, 9000, 48)])
def test_process_blocks_batches_of_frames_equal_frame_by_frame(gpu, rank, taps, K):
    """mi_convolver_bank_process_blocks: whole frames of a partitioned bank in batches of 16 / 8 / 4 / 2 (three launches per
    batch: the frames' images, ALL their tails in one pass over the partitions, the frames' outputs) against K process() calls on
    a twin bank -- bit for bit the one-launch frame step's samples, and the ring, the pending tail and the accumulator left behind
    (further frames through both, one of them in pieces).  Oracle parity of the first and last channel."""
    rng = np.random.default_rng(1000 * rank + K)
    C, frame = 5, 1 << (rank - 1)
    counts = np.array([taps, taps - 7, max(1, taps // 2), taps, frame + 1], np.uint32)
    irs = (rng.standard_normal((C, taps)) * 0.05).astype(np.float32)
    x = (rng.standard_normal((K + 4, C, frame)) * 0.25).astype(np.float32)
    a = gpu.ConvolverBank(irs, rank, counts=counts)
    b = gpu.ConvolverBank(irs, rank, counts=counts)
    ins = [gpu.DeviceBuffer.from_host(x[k]) for k in range(K + 4)]
    oa = [gpu.DeviceBuffer((C, frame)) for _ in range(K + 4)]
    ob = [gpu.DeviceBuffer((C, frame)) for _ in range(K + 4)]
    a.process(oa[0], ins[0], frame)                          # a frame in front: the ring, the tail and the accumulator are not empty
    a.process_blocks(oa[1:K + 1], ins[1:K + 1], frame)
    a.process(oa[K + 1], ins[K + 1], frame)
    a.process(oa[K + 2], ins[K + 2], 100, frame, frame)      # a frame in pieces on what the batch left
    half = gpu.DeviceBuffer((C, frame))
    a.process(half.ptr, ins[K + 2].ptr + 400, frame - 100, frame, frame)
    a.process(oa[K + 3], ins[K + 3], frame)
    for k in range(K + 2):
        b.process(ob[k], ins[k], frame)
    b.process(ob[K + 2], ins[K + 2], 100, frame, frame)
    half_b = gpu.DeviceBuffer((C, frame))
    b.process(half_b.ptr, ins[K + 2].ptr + 400, frame - 100, frame, frame)
    b.process(ob[K + 3], ins[K + 3], frame)
    assert a.faults() == 0 and b.faults() == 0
    for k in list(range(K + 2)) + [K + 3]:
        np.testing.assert_array_equal(oa[k].download(), ob[k].download(), err_msg="block %d" % k)
    np.testing.assert_array_equal(oa[K + 2].download()[:, :100], ob[K + 2].download()[:, :100])
    np.testing.assert_array_equal(half.download()[:, :frame - 100], half_b.download()[:, :frame - 100])
    y = np.concatenate([oa[k].download() for k in range(K + 2)], axis=1)
    xs = np.concatenate([x[k] for k in range(K + 2)], axis=1)
    for ch in (0, C - 1):
        ir = irs[ch, :counts[ch]]
        ref32 = oracle.Convolver(ir, rank).process_chunked(xs[ch], frame)
        check(y[ch], ref32, exact_conv(xs[ch], ir), "rank %d ch %d" % (rank, ch))
    a.close(); b.close()


def test_process_blocks_splits_where_blocks_depend_on_each_other(gpu):
    """A block that reads or writes what an earlier block of the call writes starts a new batch (here: a ring of two output
    buffers, a block fed from it); a block in place stays in its batch; other block sizes and single-partition banks are plain
    loops of calls."""
    rng = np.random.default_rng(99)
    C, rank, taps, K = 3, 10, 2000, 9
    frame = 1 << (rank - 1)
    irs = (rng.standard_normal((C, taps)) * 0.05).astype(np.float32)
    x = (rng.standard_normal((K, C, frame)) * 0.25).astype(np.float32)
    res = []
    for blocks_call in (True, False):
        bank = gpu.ConvolverBank(irs, rank)
        ins = [gpu.DeviceBuffer.from_host(x[k]) for k in range(K)]
        ring = [gpu.DeviceBuffer((C, frame)) for _ in range(2)]
        outs = [ring[k % 2] for k in range(K)]
        outs[2] = ins[2]                                     # in place
        ins[5] = ring[0]                                     # reads block 4's output
        outs[7] = gpu.DeviceBuffer((C, frame))
        got = []
        if blocks_call:
            for lo, hi in ((0, 3), (3, 9)):
                bank.process_blocks(outs[lo:hi], ins[lo:hi], frame)
                got += [outs[hi - 1].download(), outs[hi - 2].download()]
            got.append(outs[7].download())
        else:
            for k in range(K):
                bank.process(outs[k], ins[k], frame)
                if k in (2, 8):
                    got += [outs[k].download(), outs[k - 1].download()]
            got.append(outs[7].download())
        res.append(got)
        bank.close()
    for ya, yb in zip(*res):
        np.testing.assert_array_equal(ya, yb)
    for rank2, taps2, n in ((10, 300, 512), (10, 2000, 300)):   # one partition; blocks that are not frames
        a, b = gpu.ConvolverBank(irs[:, :taps2], rank2), gpu.ConvolverBank(irs[:, :taps2], rank2)
        ins = [gpu.DeviceBuffer.from_host(x[k][:, :n]) for k in range(4)]
        oa, ob = [gpu.DeviceBuffer((C, n)) for _ in range(4)], [gpu.DeviceBuffer((C, n)) for _ in range(4)]
        a.process_blocks(oa, ins, n)
        for k in range(4):
            b.process(ob[k], ins[k], n)
        for k in range(4):
            np.testing.assert_array_equal(oa[k].download(), ob[k].download())
        a.close(); b.close()


@pytest.mark.parametrize("seed", range(10))
def test_process_blocks_random_scripts(gpu, seed):
    """Differential stress of the batched frames against frame-by-frame calls, bit for bit: random ranks (10 .. 13), tap counts
    per channel, channel counts, and a random script of process_blocks calls (1 .. 40 frames, some in place, some with outputs
    that come round again or feed a later block), whole-frame calls and calls of odd sizes in between."""
    rng = np.random.default_rng(5100 + seed)
    rank = int(rng.choice([10, 10, 11, 12, 13]))
    frame = 1 << (rank - 1)
    C = int(rng.integers(1, 7))
    taps = int(rng.integers(frame + 1, 9 * frame))
    counts = np.array([int(rng.integers(1, taps + 1)) for _ in range(C)], np.uint32)
    counts[int(rng.integers(0, C))] = taps
    irs = (rng.standard_normal((C, taps)) * 0.05).astype(np.float32)
    script = []
    for _ in range(int(rng.integers(3, 7))):
        kind = int(rng.integers(0, 4))
        if kind <= 1:
            script.append(("blocks", int(rng.choice([2, 3, 5, 8, 16, 17, 40]))))
        elif kind == 2:
            script.append(("frame", 1))
        else:
            n = int(rng.integers(1, frame))
            script.append(("odd", n))
            script.append(("odd", frame - n))
    results = []
    for blocks_call in (False, True):
        r2 = np.random.default_rng(6100 + seed)
        bank = gpu.ConvolverBank(irs, rank, counts=counts)
        got = []
        for kind, n in script:
            if kind == "odd":
                d = gpu.DeviceBuffer.from_host((r2.standard_normal((C, n)) * 0.25).astype(np.float32))
                o = gpu.DeviceBuffer((C, n))
                bank.process(o, d, n)
                got.append(o.download())
                continue
            pool = [gpu.DeviceBuffer.from_host((r2.standard_normal((C, frame)) * 0.25).astype(np.float32)) for _ in range(n)]
            ins, outs = [], []
            for k in range(n):
                mode = int(r2.integers(0, 8))
                i = pool[k]
                if mode == 0:
                    o = i                                    # in place
                elif mode == 1 and outs:
                    o = outs[int(r2.integers(0, len(outs)))] # an output buffer again
                else:
                    o = gpu.DeviceBuffer((C, frame))
                if mode == 2 and outs:
                    i = outs[int(r2.integers(0, len(outs)))] # reads what an earlier block wrote
                ins.append(i); outs.append(o)
            if blocks_call and kind == "blocks":
                bank.process_blocks(outs, ins, frame)
            else:
                for o, i in zip(outs, ins):
                    bank.process(o, i, frame)
            got += [b.download() for b in pool] + [o.download() for o in outs]
        assert bank.faults() == 0
        results.append(got)
        bank.close()
    for u, v in zip(*results):
        np.testing.assert_array_equal(u, v, err_msg=str((seed, rank, C, taps, script)))
