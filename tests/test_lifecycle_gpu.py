"""Create / use / destroy cycles of every bank: device memory comes back (no leak per object), through the C-ABI."""
import numpy as np
import pytest

from oracle import filter_design as fd
import workloads as wl

pytestmark = pytest.mark.gpu


def _free_bytes():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")             # the runtime the library itself runs on (already loaded)
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _cycle(gpu, kind):
    C, n = 8, 1024
    x = gpu.DeviceBuffer.from_host((np.random.default_rng(1).standard_normal((C, n)) * 0.25).astype(np.float32))
    y = gpu.DeviceBuffer((C, n))
    if kind == "biquad":
        b = gpu.BiquadBank(C, 8)
        q = wl.design(fd.FLT_BT_LRX_LOPASS, 4, 3000.0, 0, 1.0, 0.75)
        for c in range(C):
            b.set_chains(c, q)
        b.process(y, x, n)
    elif kind == "convolver":
        b = gpu.ConvolverBank(np.random.default_rng(2).standard_normal((C, 3000)).astype(np.float32), 9)
        b.process(y, x, n)
    elif kind == "spectral":
        b = gpu.SpectralBank(C, 12); b.set_rank(10); b.process(y, x, n)
    elif kind == "analyzer":
        b = gpu.AnalyzerBank(C, 12, 48000, 1.0, 0)
        for what, v in ((b.SAMPLE_RATE, 48000), (b.RATE, 50.0), (b.RANK, 10), (b.WINDOW, 0), (b.REACTIVITY, 0.2), (b.SHIFT, 1.0)):
            b.configure(what, v)
        b.process(x, n)
    elif kind == "delay":
        b = gpu.DelayBank(C, 5000); b.process(y, x, n)
    elif kind == "ring":
        b = gpu.RingBank(C, 4096); b.append(x, n)
    elif kind == "loudness":
        b = gpu.LoudnessBank(C // 2, 2, 400.0); b.set_sample_rate(48000)
        b.process(gpu.DeviceBuffer((C // 2, n)), None, x, n)
    elif kind == "ilufs":
        b = gpu.ILUFSBank(C // 2, 2, 10.0, 400.0); b.set_sample_rate(48000)
        b.process(gpu.DeviceBuffer((C // 2, n)), x, n)
    elif kind == "splitter":
        b = gpu.SplitterBank(C, 12, 2); b.set_rank(10); b.set_chunk_rank(8)
        b.bind_copy(0); b.bind_mask(1, np.ones(1 << 10, np.float32))
        b.process([y, gpu.DeviceBuffer((C, n))], x, n)
    elif kind == "crossover":
        b = gpu.CrossoverBank(C, 3); b.set_sample_rate(48000)
        for i, f in enumerate((500.0, 4000.0)):
            b.set_slope(i, 2); b.set_frequency(i, f)
        b.process([y, gpu.DeviceBuffer((C, n)), gpu.DeviceBuffer((C, n))], x, n)
    else:
        b = gpu.EqualizerBank(C, 4, 10); b.set_mode(2); b.set_sample_rate(48000)
        for c in range(C):
            b.set_params(0, fd.FLT_BT_RLC_BELL, 1, 1000.0, 1000.0, 2.0, 1.0, channel=c)
        b.process(y, x, n)
    b.close()


@pytest.mark.parametrize("kind", ["biquad", "convolver", "spectral", "analyzer", "delay", "ring", "loudness", "ilufs",
                                  "splitter", "crossover", "equalizer"])
def test_create_use_destroy_returns_device_memory(gpu, kind):
    for _ in range(3):                      # first objects: code objects, allocator pools, the shared twiddle table
        _cycle(gpu, kind)
    before = _free_bytes()
    for _ in range(40):
        _cycle(gpu, kind)
    after = _free_bytes()
    assert before - after < 8 << 20, "%s: %.1f MiB of device memory did not come back after 40 objects" % (kind, (before - after) / 2 ** 20)
