"""Small object wrappers over the C-ABI handles (no arithmetic here)."""
import ctypes
from ctypes import byref, c_float, c_int, c_uint32, c_void_p

import numpy as np

from .capi import BiquadX1, FilterCascade, FilterParams, check, lib


def device_count():
    return int(lib.mi_dspu_device_count())


def last_launch():
    """The hot-path kernel this thread launched last (mi_dspu_last_launch): which launch a call took."""
    return (lib.mi_dspu_last_launch() or b"").decode()


def source_sha(name):
    """SHA-256 (16 hex digits) of csrc/<name> as the loaded library was built from it (mi_dspu_source_sha), or None."""
    v = lib.mi_dspu_source_sha(name.encode())
    return v.decode() if v else None


def last_stream_clock():
    """(GHz, microseconds) of the last mi_biquad_bank_process_blocks launch (mi_dspu_last_stream_clock)."""
    g, us = ctypes.c_double(), ctypes.c_double()
    check(lib.mi_dspu_last_stream_clock(byref(g), byref(us)))
    return g.value, us.value


def _ptr(x):
    """Device address of a DeviceBuffer, a torch tensor or a raw int."""
    if isinstance(x, DeviceBuffer):
        return c_void_p(x.ptr)
    if hasattr(x, "data_ptr"):
        return c_void_p(x.data_ptr())
    return c_void_p(int(x))


def _stream(s):
    if s is None:
        return c_void_p(0)
    if hasattr(s, "cuda_stream"):
        return c_void_p(s.cuda_stream)
    return c_void_p(int(s))


class DeviceBuffer:
    """float32 device array owned by the library allocator (mi_dspu_malloc)."""

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in (shape if hasattr(shape, "__len__") else (shape,)))
        self.size = int(np.prod(self.shape)) if self.shape else 1
        p = c_void_p()
        check(lib.mi_dspu_malloc(byref(p), self.size * 4))
        self.ptr = p.value or 0

    @classmethod
    def from_host(cls, array, stream=None):
        a = np.ascontiguousarray(array, dtype=np.float32)
        buf = cls(a.shape)
        buf.upload(a, stream)
        return buf

    def upload(self, array, stream=None):
        a = np.ascontiguousarray(array, dtype=np.float32)
        assert a.size == self.size
        check(lib.mi_dspu_copy_h2d(c_void_p(self.ptr), a.ctypes.data_as(c_void_p), a.nbytes, _stream(stream)))
        check(lib.mi_dspu_stream_synchronize(_stream(stream)))

    def download(self, stream=None):
        out = np.empty(self.shape, dtype=np.float32)
        check(lib.mi_dspu_copy_d2h(out.ctypes.data_as(c_void_p), c_void_p(self.ptr), out.nbytes, _stream(stream)))
        check(lib.mi_dspu_stream_synchronize(_stream(stream)))
        return out

    def zero(self, stream=None):
        check(lib.mi_dspu_memset(c_void_p(self.ptr), 0, self.size * 4, _stream(stream)))

    def free(self):
        if self.ptr:
            lib.mi_dspu_free(c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _chains(coefs):
    """(n,5) float32 {b0,b1,b2,a1,a2} rows -> ctypes array of mi_biquad_x1_t."""
    c = np.ascontiguousarray(coefs, dtype=np.float32).reshape(-1, 5)
    full = np.zeros((c.shape[0], 8), dtype=np.float32)
    full[:, :5] = c
    arr = (BiquadX1 * max(1, c.shape[0])).from_buffer_copy(full.tobytes() if c.shape[0] else bytes(32))
    return arr, c.shape[0]


class BiquadBank:
    """`channels` x lsp::dspu::FilterBank on the device (mi_biquad_bank_*)."""

    def __init__(self, channels, max_sections):
        h = c_void_p()
        check(lib.mi_biquad_bank_create(byref(h), channels, max_sections))
        self.handle = h
        self.channels = channels
        self.max_sections = max(1, max_sections)

    def set_chains(self, channel, coefs, clear=False):
        arr, n = _chains(coefs)
        check(lib.mi_biquad_bank_set_chains(self.handle, channel, arr, n, int(clear)))

    def set_row_enabled(self, channel, enabled=True):
        """A channel that is switched off is skipped by process(): state kept, output row not written."""
        check(lib.mi_biquad_bank_set_row_enabled(self.handle, channel, 1 if enabled else 0))

    def set_exact(self, on=True):
        """The reference's serial recurrence, bit for bit (mi_biquad_bank_set_exact), instead of the time-parallel kernels."""
        check(lib.mi_biquad_bank_set_exact(self.handle, 1 if on else 0))

    def set_all_chains(self, coefs, clear=False):
        c = np.ascontiguousarray(coefs, dtype=np.float32)
        assert c.ndim == 3 and c.shape[0] == self.channels and c.shape[2] == 5
        arr, _ = _chains(c.reshape(-1, 5))
        check(lib.mi_biquad_bank_set_all_chains(self.handle, arr, c.shape[1], int(clear)))

    def size(self, channel):
        n = c_uint32()
        check(lib.mi_biquad_bank_size(self.handle, channel, byref(n)))
        return n.value

    def commit(self, stream=None):
        check(lib.mi_biquad_bank_commit(self.handle, _stream(stream)))

    def reset(self, channel=None, stream=None):
        check(lib.mi_biquad_bank_reset(self.handle, 0xFFFFFFFF if channel is None else channel, _stream(stream)))

    def process(self, out, inp, samples, out_stride=None, in_stride=None, stream=None):
        check(lib.mi_biquad_bank_process(self.handle, _ptr(out), _ptr(inp), samples,
                                         samples if out_stride is None else out_stride,
                                         samples if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, outs, inps, samples, out_stride=None, in_stride=None, stream=None):
        """len(outs) consecutive process() calls issued by one C call (mi_biquad_bank_process_blocks)."""
        n = len(outs)
        assert n == len(inps)
        po = (c_void_p * n)(*[_ptr(b) for b in outs])
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_biquad_bank_process_blocks(self.handle, po, pi, n, samples,
                                                samples if out_stride is None else out_stride,
                                                samples if in_stride is None else in_stride, _stream(stream)))

    def impulse_response(self, out, samples, out_stride=None, stream=None):
        check(lib.mi_biquad_bank_impulse_response(self.handle, _ptr(out), samples,
                                                  samples if out_stride is None else out_stride, _stream(stream)))

    def get_state(self, stream=None):
        st = np.empty((self.channels, self.max_sections, 2), dtype=np.float32)
        check(lib.mi_biquad_bank_get_state(self.handle, st.ctypes.data_as(c_void_p), _stream(stream)))
        return st

    def set_state(self, state, stream=None):
        st = np.ascontiguousarray(state, dtype=np.float32)
        assert st.shape == (self.channels, self.max_sections, 2)
        check(lib.mi_biquad_bank_set_state(self.handle, st.ctypes.data_as(c_void_p), _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_biquad_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConvolverBank:
    """`channels` x lsp::dspu::Convolver on the device (mi_convolver_bank_*)."""

    def __init__(self, irs, rank, counts=None, phase=0.0, stream=None):
        irs = np.ascontiguousarray(irs, dtype=np.float32)
        if irs.ndim == 1:
            irs = irs.reshape(1, -1)
        self.channels = irs.shape[0]
        cnt = None
        if counts is not None:
            cnt = np.ascontiguousarray(counts, dtype=np.uint32)
            assert cnt.shape == (self.channels,)
        h = c_void_p()
        check(lib.mi_convolver_bank_create(byref(h), self.channels,
                                           irs.ctypes.data_as(c_void_p) if irs.size else None, irs.shape[1],
                                           cnt.ctypes.data_as(c_void_p) if cnt is not None else None,
                                           irs.shape[1], rank, phase, _stream(stream)))
        self.handle = h

    def info(self):
        v = [c_uint32() for _ in range(4)]
        check(lib.mi_convolver_bank_info(self.handle, *[byref(x) for x in v]))
        return dict(zip(("rank", "frame", "partitions", "data_size"), [x.value for x in v]))

    def faults(self, stream=None):
        """How often the two roles of the one-launch frame step gave up waiting for each other (0 on a healthy device)."""
        n = c_uint32(0)
        check(lib.mi_convolver_bank_faults(self.handle, byref(n), _stream(stream)))
        return int(n.value)

    def reset(self, stream=None):
        check(lib.mi_convolver_bank_reset(self.handle, _stream(stream)))

    def process(self, out, inp, samples, out_stride=None, in_stride=None, stream=None):
        check(lib.mi_convolver_bank_process(self.handle, _ptr(out), _ptr(inp), samples,
                                            samples if out_stride is None else out_stride,
                                            samples if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, outs, inps, samples, out_stride=None, in_stride=None, stream=None):
        """len(outs) consecutive process() calls issued by one C call (mi_convolver_bank_process_blocks)."""
        n = len(outs)
        assert n == len(inps)
        po = (c_void_p * n)(*[_ptr(b) for b in outs])
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_convolver_bank_process_blocks(self.handle, po, pi, n, samples,
                                                   samples if out_stride is None else out_stride,
                                                   samples if in_stride is None else in_stride, _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_convolver_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def design_filter(ftype, slope=1, freq=1000.0, freq2=1000.0, gain=1.0, quality=0.0, sample_rate=48000):
    """Filter::update + rebuild on the host: returns (mode, cascades[n][2][3], sections[n][5])."""
    fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), float(gain), float(quality))
    chains = (BiquadX1 * 256)()
    casc = (FilterCascade * 256)()
    nch, nca, mode = c_uint32(), c_uint32(), ctypes.c_int()
    check(lib.mi_filter_design(byref(fp), sample_rate, chains, 256, byref(nch), casc, 256, byref(nca), byref(mode)))
    sec = np.array([[c.b0, c.b1, c.b2, c.a1, c.a2] for c in chains[:nch.value]], dtype=np.float32).reshape(-1, 5)
    cas = np.array([[list(c.t)[:3], list(c.b)[:3]] for c in casc[:nca.value]], dtype=np.float32).reshape(-1, 2, 3)
    return mode.value, cas, sec


def filter_freq_chart(freqs, ftype, slope=1, freq=1000.0, freq2=1000.0, gain=1.0, quality=0.0, sample_rate=48000):
    fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), float(gain), float(quality))
    f = np.ascontiguousarray(freqs, dtype=np.float32)
    c = np.empty(2 * f.size, np.float32)
    check(lib.mi_filter_freq_chart(byref(fp), sample_rate, c.ctypes.data_as(c_void_p), f.ctypes.data_as(c_void_p), f.size))
    return c[0::2] + 1j * c[1::2]


def make_window(n, wtype):
    out = np.empty(n, np.float32)
    check(lib.mi_window(out.ctypes.data_as(c_void_p), n, int(wtype)))
    return out


def make_window_general(n, wtype, params):
    """windows::*_general (misc/windows.h:71-155): the window family `wtype` with its parameter list."""
    out = np.empty(n, np.float32)
    q = np.ascontiguousarray(params, dtype=np.float32)
    check(lib.mi_window_general(out.ctypes.data_as(c_void_p), n, int(wtype), q.ctypes.data_as(c_void_p), q.size))
    return out


class SpectralBank:
    """`channels` x lsp::dspu::SpectralProcessor (one MultiSpectralProcessor) on the device."""
    OP_NONE, OP_MASK, OP_CALLBACK = 0, 1, 2

    def __init__(self, channels, max_rank):
        h = c_void_p()
        check(lib.mi_spectral_bank_create(byref(h), channels, max_rank))
        self.handle, self.channels = h, channels
        self._cb = None

    def set_rank(self, rank):
        check(lib.mi_spectral_bank_set_rank(self.handle, rank))

    def set_phase(self, phase):
        check(lib.mi_spectral_bank_set_phase(self.handle, phase))

    def set_timing(self, eager):
        """False: SpectralProcessor (a full frame is transformed when the next sample arrives); True: Multi..."""
        check(lib.mi_spectral_bank_set_timing(self.handle, 1 if eager else 0))

    def get(self):
        v = [c_uint32() for _ in range(3)]
        check(lib.mi_spectral_bank_get(self.handle, *[byref(x) for x in v]))
        return dict(zip(("rank", "latency", "remaining"), [x.value for x in v]))

    def bind(self, pyfunc):
        """pyfunc(spectrum_dev_ptr, rank, channels, stream) runs on the host between the two transforms."""
        from .capi import SPECTRAL_FUNC
        if pyfunc is None:
            check(lib.mi_spectral_bank_unbind(self.handle))
            self._cb = None
            return
        self._cb = SPECTRAL_FUNC(lambda obj, subj, spec, rank, ch, st: pyfunc(spec, rank, ch, st))
        check(lib.mi_spectral_bank_bind(self.handle, ctypes.cast(self._cb, c_void_p), None, None))

    def bind_mask(self, mask, stream=None):
        m = np.ascontiguousarray(mask, dtype=np.float32)
        stride = 0 if m.ndim == 1 else m.shape[1]
        check(lib.mi_spectral_bank_bind_mask(self.handle, m.ctypes.data_as(c_void_p), stride, _stream(stream)))

    def bind_channels(self, has_in=None, has_out=None, stream=None):
        a = None if has_in is None else np.ascontiguousarray(has_in, dtype=np.uint8)
        b = None if has_out is None else np.ascontiguousarray(has_out, dtype=np.uint8)
        check(lib.mi_spectral_bank_bind_channels(self.handle, a.ctypes.data_as(c_void_p) if a is not None else None,
                                                 b.ctypes.data_as(c_void_p) if b is not None else None, _stream(stream)))

    def reset(self, stream=None):
        check(lib.mi_spectral_bank_reset(self.handle, _stream(stream)))

    def process(self, out, inp, count, out_stride=None, in_stride=None, stream=None):
        check(lib.mi_spectral_bank_process(self.handle, _ptr(out) if out is not None else None, _ptr(inp), count,
                                           count if out_stride is None else out_stride,
                                           count if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, outs, inps, count, out_stride=None, in_stride=None, stream=None):
        """len(outs) consecutive process() calls issued by one C call (mi_spectral_bank_process_blocks)."""
        n = len(outs)
        assert n == len(inps)
        po = (c_void_p * n)(*[(_ptr(b) if b is not None else None) for b in outs])
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_spectral_bank_process_blocks(self.handle, po, pi, n, count,
                                                  count if out_stride is None else out_stride,
                                                  count if in_stride is None else in_stride, _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_spectral_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AnalyzerBank:
    """lsp::dspu::Analyzer with `channels` channels on the device."""
    SAMPLE_RATE, RATE, WINDOW, ENVELOPE, SHIFT, REACTIVITY, RANK, ACTIVE = range(8)
    CH_FREEZE, CH_ENABLE, CH_DELAY = range(3)

    def __init__(self, channels, max_rank, max_sample_rate, min_rate, max_delay=0):
        h = c_void_p()
        check(lib.mi_analyzer_bank_create(byref(h), channels, max_rank, max_sample_rate, min_rate, max_delay))
        self.handle, self.channels = h, channels

    def configure(self, what, value):
        check(lib.mi_analyzer_bank_configure(self.handle, what, float(value)))

    def channel(self, channel, what, value):
        check(lib.mi_analyzer_bank_channel(self.handle, channel, what, int(value)))

    def process(self, inp, samples, in_stride=None, stream=None):
        check(lib.mi_analyzer_bank_process(self.handle, _ptr(inp) if inp is not None else None, samples,
                                           samples if in_stride is None else in_stride, _stream(stream)))

    def info(self):
        v = [c_uint32() for _ in range(4)]
        check(lib.mi_analyzer_bank_info(self.handle, *[byref(x) for x in v]))
        return dict(zip(("rank", "bins", "period", "step"), [x.value for x in v]))

    def get_spectrum(self, idx, stream=None):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        didx = c_void_p()
        check(lib.mi_dspu_malloc(byref(didx), idx.nbytes))
        check(lib.mi_dspu_copy_h2d(didx, idx.ctypes.data_as(c_void_p), idx.nbytes, _stream(stream)))
        out = DeviceBuffer((self.channels, idx.size))
        check(lib.mi_analyzer_bank_get_spectrum(self.handle, c_void_p(out.ptr), idx.size, didx, idx.size, _stream(stream)))
        res = out.download(stream)
        lib.mi_dspu_free(didx)
        return res

    def allreduce_bins(self, bins, frames, comm, stream=None):
        """Sum `bins` (device [frames][bins]) over the ranks of `comm` in place: RCCL from the library's host side."""
        check(lib.mi_analyzer_bank_allreduce_bins(self.handle, _ptr(bins), int(frames), comm.handle, _stream(stream)))

    def allreduce_bins_begin(self, partial, total, frames, comm, slot, stream=None):
        """The same collective on the communicator's side stream, behind what `stream` has enqueued so far; comm.wait(slot, stream)
        before `total` is read or the slot's buffers are written again (mi_analyzer_bank_allreduce_bins_begin)."""
        check(lib.mi_analyzer_bank_allreduce_bins_begin(self.handle, _ptr(partial), _ptr(total), int(frames), comm.handle, int(slot), _stream(stream)))

    def reduce_bins(self, out, with_envelope=False, stream=None):
        check(lib.mi_analyzer_bank_reduce_bins(self.handle, _ptr(out), int(with_envelope), _stream(stream)))

    def process_reduce(self, inp, samples, out, with_envelope=False, in_stride=None, stream=None):
        """process() + reduce_bins(), the reduction riding on the analysis launch (mi_analyzer_bank_process_reduce)."""
        check(lib.mi_analyzer_bank_process_reduce(self.handle, _ptr(inp) if inp is not None else None, samples,
                                                  samples if in_stride is None else in_stride, _ptr(out),
                                                  int(with_envelope), _stream(stream)))

    def process_reduce_frames(self, inps, samples, out, with_envelope=False, in_stride=None, out_stride=None, stream=None):
        """len(inps) consecutive process_reduce() calls in one C call; frame k's sums go to row k of `out`."""
        n = len(inps)
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        bins = self.info()["bins"]
        check(lib.mi_analyzer_bank_process_reduce_frames(self.handle, pi, n, samples, samples if in_stride is None else in_stride,
                                                         _ptr(out), bins if out_stride is None else out_stride, int(with_envelope),
                                                         _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_analyzer_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dynfilter_sections(ftype, slope, freq, freq2, quality, gain, sample_rate=48000):
    """Host: the digital sections of a dynamic filter at a fixed gain, (n, 5) float32."""
    fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), 1.0, float(quality))
    n = c_uint32()
    arr = (BiquadX1 * 128)()
    check(lib.mi_dynfilter_sections(byref(fp), int(sample_rate), float(gain), arr, 128, byref(n)))
    return np.array([[c.b0, c.b1, c.b2, c.a1, c.a2] for c in arr[:n.value]], dtype=np.float32).reshape(-1, 5)


def dynfilter_freq_chart(freqs, ftype, slope, freq, freq2, quality, gain, sample_rate=48000):
    fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), 1.0, float(quality))
    f = np.ascontiguousarray(freqs, dtype=np.float32)
    c = np.empty(2 * f.size, np.float32)
    check(lib.mi_dynfilter_freq_chart(byref(fp), int(sample_rate), c.ctypes.data_as(c_void_p), f.ctypes.data_as(c_void_p),
                                      float(gain), f.size))
    return c[0::2] + 1j * c[1::2]


class DynFilterBank:
    """`channels` x lsp::dspu::DynamicFilters(filters) on the device (shared settings, per-channel gain curves)."""

    def __init__(self, channels, filters):
        self.channels, self.filters = int(channels), int(filters)
        h = c_void_p()
        check(lib.mi_dynfilter_bank_create(byref(h), self.channels, self.filters))
        self.handle = h

    def set_sample_rate(self, sr):
        check(lib.mi_dynfilter_bank_set_sample_rate(self.handle, int(sr)))

    def set_params(self, fid, ftype, slope, freq, freq2, gain, quality):
        fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), float(gain), float(quality))
        check(lib.mi_dynfilter_bank_set_params(self.handle, int(fid), byref(fp)))

    def get_params(self, fid):
        fp, act = FilterParams(), c_int()
        check(lib.mi_dynfilter_bank_get_params(self.handle, int(fid), byref(fp), byref(act)))
        return dict(nType=fp.nType, nSlope=fp.nSlope, fFreq=fp.fFreq, fFreq2=fp.fFreq2, fGain=fp.fGain, fQuality=fp.fQuality), bool(act.value)

    def set_filter_active(self, fid, active=True):
        check(lib.mi_dynfilter_bank_set_filter_active(self.handle, int(fid), 1 if active else 0))

    def process(self, fid, out, inp, gain, samples, out_stride=None, in_stride=None, gain_stride=None, stream=None):
        check(lib.mi_dynfilter_bank_process(self.handle, int(fid), _ptr(out), _ptr(inp), _ptr(gain) if gain is not None else c_void_p(0),
                                            int(samples), int(out_stride or samples), int(in_stride or samples),
                                            int(gain_stride or samples), _stream(stream)))

    def close(self):
        if self.handle is not None:
            lib.mi_dynfilter_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """mi_dspu_comm_t: an RCCL communicator owned by the library (one process per GPU)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.ID_BYTES)
        check(lib.mi_dspu_comm_unique_id(buf))
        return bytes(buf.raw)

    def __init__(self, unique_id, nranks, rank):
        assert len(unique_id) == Comm.ID_BYTES
        h = c_void_p()
        check(lib.mi_dspu_comm_create(byref(h), ctypes.c_char_p(unique_id), int(nranks), int(rank)))
        self.handle = h

    def info(self):
        n, r = c_int(), c_int()
        check(lib.mi_dspu_comm_info(self.handle, byref(n), byref(r)))
        return n.value, r.value

    def wait(self, slot, stream=None):
        """`stream` waits (on the device) for the collective begun in `slot` (mi_dspu_comm_wait)."""
        check(lib.mi_dspu_comm_wait(self.handle, int(slot), _stream(stream)))

    def close(self):
        if self.handle is not None:
            lib.mi_dspu_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DelayBank:
    """`channels` x lsp::dspu::Delay on the device."""

    def __init__(self, channels, max_size):
        h = c_void_p()
        check(lib.mi_delay_bank_create(byref(h), channels, max_size))
        self.handle, self.channels = h, channels

    def set_delay(self, delay, channel=None):
        check(lib.mi_delay_bank_set_delay(self.handle, 0xFFFFFFFF if channel is None else channel, int(delay)))

    def get(self, channel=0):
        v = [c_uint32() for _ in range(4)]
        check(lib.mi_delay_bank_get(self.handle, channel, *[byref(x) for x in v]))
        return dict(zip(("delay", "size", "head", "tail"), [x.value for x in v]))

    def clear(self, stream=None):
        check(lib.mi_delay_bank_clear(self.handle, _stream(stream)))

    def append(self, inp, count, in_stride=None, stream=None):
        check(lib.mi_delay_bank_append(self.handle, _ptr(inp), count, count if in_stride is None else in_stride, _stream(stream)))

    def process(self, out, inp, count, add=False, gain=None, gain_vec=None, out_stride=None, in_stride=None, stream=None):
        mode = 2 if gain_vec is not None else (1 if gain is not None else 0)
        check(lib.mi_delay_bank_process(self.handle, _ptr(out), _ptr(inp), count,
                                        count if out_stride is None else out_stride,
                                        count if in_stride is None else in_stride, int(add), mode,
                                        0.0 if gain is None else float(gain),
                                        _ptr(gain_vec) if gain_vec is not None else None, count, _stream(stream)))

    def append_rows(self, rows, inp, count, in_stride=None, stream=None):
        """Delay::append for the listed lines only (row r of `inp` belongs to channel rows[r])."""
        rw = np.ascontiguousarray(rows, dtype=np.uint32)
        check(lib.mi_delay_bank_append_rows(self.handle, rw.ctypes.data_as(c_void_p), len(rw), _ptr(inp), count,
                                            count if in_stride is None else in_stride, _stream(stream)))

    def process_rows(self, rows, out, inp, count, add=False, gain=None, gain_vec=None, out_stride=None, in_stride=None, stream=None):
        """Delay::process for the listed lines only: they alone are written and move on."""
        rw = np.ascontiguousarray(rows, dtype=np.uint32)
        mode = 2 if gain_vec is not None else (1 if gain is not None else 0)
        check(lib.mi_delay_bank_process_rows(self.handle, rw.ctypes.data_as(c_void_p), len(rw), _ptr(out), _ptr(inp), count,
                                             count if out_stride is None else out_stride,
                                             count if in_stride is None else in_stride, int(add), mode,
                                             0.0 if gain is None else float(gain),
                                             _ptr(gain_vec) if gain_vec is not None else None, count, _stream(stream)))

    def process_ramping(self, out, inp, new_delays, count, gain=None, gain_vec=None, stream=None):
        nd = np.ascontiguousarray(new_delays, dtype=np.uint32)
        assert nd.shape == (self.channels,)
        mode = 2 if gain_vec is not None else (1 if gain is not None else 0)
        check(lib.mi_delay_bank_process_ramping(self.handle, _ptr(out), _ptr(inp), nd.ctypes.data_as(c_void_p), count,
                                                count, count, mode, 0.0 if gain is None else float(gain),
                                                _ptr(gain_vec) if gain_vec is not None else None, count, _stream(stream)))

    def process_ramping_rows(self, rows, out, inp, new_delays, count, gain=None, gain_vec=None, stream=None):
        """Delay::process_ramping for the listed lines only (row r of the buffers and new_delays[r] belong to line rows[r])."""
        rw = np.ascontiguousarray(rows, dtype=np.uint32)
        nd = np.ascontiguousarray(new_delays, dtype=np.uint32)
        assert nd.shape == rw.shape
        mode = 2 if gain_vec is not None else (1 if gain is not None else 0)
        check(lib.mi_delay_bank_process_ramping_rows(self.handle, rw.ctypes.data_as(c_void_p), len(rw), _ptr(out), _ptr(inp),
                                                     nd.ctypes.data_as(c_void_p), count, count, count, mode,
                                                     0.0 if gain is None else float(gain),
                                                     _ptr(gain_vec) if gain_vec is not None else None, count, _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_delay_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingBank:
    """`channels` x lsp::dspu::RingBuffer on the device."""

    def __init__(self, channels, size, fill=0.0):
        h = c_void_p()
        check(lib.mi_ring_bank_create(byref(h), channels, size, fill))
        self.handle, self.channels = h, channels

    def fill(self, value=0.0, stream=None):
        check(lib.mi_ring_bank_fill(self.handle, value, _stream(stream)))

    def append(self, inp, count, in_stride=None, stream=None):
        n = ctypes.c_size_t()
        check(lib.mi_ring_bank_append(self.handle, _ptr(inp), count, count if in_stride is None else in_stride,
                                      byref(n), _stream(stream)))
        return n.value

    def get(self, out, offset, count, out_stride=None, stream=None):
        n = ctypes.c_size_t()
        check(lib.mi_ring_bank_get(self.handle, _ptr(out), offset, count, count if out_stride is None else out_stride,
                                   byref(n), _stream(stream)))
        return n.value

    def info(self, offset=0):
        v = [c_uint32() for _ in range(3)]
        check(lib.mi_ring_bank_info(self.handle, offset, *[byref(x) for x in v]))
        return dict(zip(("capacity", "head", "tail_position"), [x.value for x in v]))

    def close(self):
        if self.handle:
            lib.mi_ring_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoudnessBank:
    """`meters` x lsp::dspu::LoudnessMeter(channels) sharing one configuration (mi_loudness_bank_*)."""
    WEIGHT_NONE, WEIGHT_A, WEIGHT_B, WEIGHT_C, WEIGHT_D, WEIGHT_K = range(6)

    def __init__(self, meters, channels, max_period_ms=400.0):
        h = c_void_p()
        check(lib.mi_loudness_bank_create(byref(h), meters, channels, float(max_period_ms)))
        self.handle, self.meters, self.channels = h, meters, channels

    def set_sample_rate(self, sr, stream=None):
        check(lib.mi_loudness_bank_set_sample_rate(self.handle, sr, _stream(stream)))

    def set_period(self, ms):
        check(lib.mi_loudness_bank_set_period(self.handle, float(ms)))

    def set_weighting(self, w):
        check(lib.mi_loudness_bank_set_weighting(self.handle, int(w)))

    def set_designation(self, channel, designation):
        check(lib.mi_loudness_bank_set_designation(self.handle, channel, int(designation)))

    def set_link(self, channel, link):
        check(lib.mi_loudness_bank_set_link(self.handle, channel, float(link)))

    def set_active(self, channel, active=True, stream=None):
        check(lib.mi_loudness_bank_set_active(self.handle, channel, 1 if active else 0, _stream(stream)))

    def set_bound(self, channel, bound=True):
        check(lib.mi_loudness_bank_set_bound(self.handle, channel, 1 if bound else 0))

    def clear(self, stream=None):
        check(lib.mi_loudness_bank_clear(self.handle, _stream(stream)))

    def latency(self):
        v = c_uint32()
        check(lib.mi_loudness_bank_latency(self.handle, byref(v)))
        return v.value

    def process(self, out, ch_out, inp, count, out_stride=None, in_stride=None, gain=None, stream=None):
        """gain=None: process(out, count), which also records loudness(); a number: process(out, count, gain)."""
        args = (self.handle, _ptr(out) if out is not None else None, _ptr(ch_out) if ch_out is not None else None, _ptr(inp),
                count, count if out_stride is None else out_stride, count if in_stride is None else in_stride)
        if gain is None:
            check(lib.mi_loudness_bank_process(*args, _stream(stream)))
        else:
            check(lib.mi_loudness_bank_process_gain(*args, float(gain), _stream(stream)))

    def loudness(self, stream=None):
        v = (c_float * self.meters)()
        check(lib.mi_loudness_bank_loudness(self.handle, v, _stream(stream)))
        return np.array(list(v), np.float32)

    def close(self):
        if self.handle:
            lib.mi_loudness_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ILUFSBank:
    """`meters` x lsp::dspu::ILUFSMeter(channels) sharing one configuration (mi_ilufs_bank_*)."""
    DBFS_TO_LUFS_SHIFT_GAIN = 0.923527857225

    def __init__(self, meters, channels, max_int_time=60.0, block_period_ms=400.0):
        h = c_void_p()
        check(lib.mi_ilufs_bank_create(byref(h), meters, channels, float(max_int_time), float(block_period_ms)))
        self.handle, self.meters, self.channels = h, meters, channels

    def set_sample_rate(self, sr, stream=None):
        check(lib.mi_ilufs_bank_set_sample_rate(self.handle, sr, _stream(stream)))

    def set_integration_period(self, seconds, stream=None):
        check(lib.mi_ilufs_bank_set_integration_period(self.handle, float(seconds), _stream(stream)))

    def set_weighting(self, w):
        check(lib.mi_ilufs_bank_set_weighting(self.handle, int(w)))

    def set_designation(self, channel, designation):
        check(lib.mi_ilufs_bank_set_designation(self.handle, channel, int(designation)))

    def set_active(self, channel, active=True):
        check(lib.mi_ilufs_bank_set_active(self.handle, channel, 1 if active else 0))

    def clear(self, stream=None):
        check(lib.mi_ilufs_bank_clear(self.handle, _stream(stream)))

    def process(self, out, inp, count, out_stride=None, in_stride=None, gain=DBFS_TO_LUFS_SHIFT_GAIN, stream=None):
        check(lib.mi_ilufs_bank_process(self.handle, _ptr(out) if out is not None else None, _ptr(inp), count,
                                        count if out_stride is None else out_stride,
                                        count if in_stride is None else in_stride, float(gain), _stream(stream)))

    def loudness(self, stream=None):
        v = (c_float * self.meters)()
        check(lib.mi_ilufs_bank_loudness(self.handle, v, _stream(stream)))
        return np.array(list(v), np.float32)

    def history(self, stream=None):
        """(hist [meters][size], head [meters], count [meters]) as the meters hold them."""
        size = c_uint32()
        check(lib.mi_ilufs_bank_history(self.handle, None, byref(size), None, None, _stream(stream)))
        hist = np.zeros((self.meters, size.value), np.float32)
        head = (c_uint32 * self.meters)()
        count = (c_uint32 * self.meters)()
        check(lib.mi_ilufs_bank_history(self.handle, hist.ctypes.data_as(c_void_p) if size.value else None, byref(size), head, count,
                                        _stream(stream)))
        return hist, np.array(head[:], np.int64), np.array(count[:], np.int64)

    def close(self):
        if self.handle:
            lib.mi_ilufs_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SplitterBank:
    """lsp::dspu::SpectralSplitter for `channels` channels sharing the settings (mi_splitter_bank_*)."""

    def __init__(self, channels, max_rank, handlers):
        h = c_void_p()
        check(lib.mi_splitter_bank_create(byref(h), channels, max_rank, handlers))
        self.handle, self.channels, self.handlers = h, channels, handlers
        self._cbs = {}

    def set_rank(self, rank):
        check(lib.mi_splitter_bank_set_rank(self.handle, rank))

    def set_chunk_rank(self, rank):
        check(lib.mi_splitter_bank_set_chunk_rank(self.handle, rank))

    def set_phase(self, phase):
        check(lib.mi_splitter_bank_set_phase(self.handle, float(phase)))

    def _get(self):
        v = [c_uint32() for _ in range(4)]
        check(lib.mi_splitter_bank_get(self.handle, *[byref(x) for x in v]))
        return [x.value for x in v]

    def rank(self):
        return self._get()[0]

    def chunk_rank(self):
        return self._get()[1]

    def latency(self):
        return self._get()[2]

    def remaining(self):
        return self._get()[3]

    def bind_copy(self, handler, stream=None):
        check(lib.mi_splitter_bank_bind_copy(self.handle, handler, _stream(stream)))

    def bind_mask(self, handler, mask, stream=None):
        """mask: 2^rank gains (shared) or [channels][2^rank], host array."""
        from ctypes import POINTER as _P
        m = np.ascontiguousarray(mask, dtype=np.float32)
        stride = 0 if m.ndim == 1 else m.shape[1]
        check(lib.mi_splitter_bank_bind_mask(self.handle, handler, m.ctypes.data_as(_P(c_float)), stride, _stream(stream)))

    def bind_callback(self, handler, pyfunc, stream=None):
        """pyfunc(out_ptr, in_ptr, rank, channels, stream): device addresses of [channels][2 * 2^rank] floats."""
        from ctypes import cast
        from .capi import SPLITTER_FUNC
        cb = SPLITTER_FUNC(lambda obj, subj, out, inp, rank, ch, st: pyfunc(out, inp, rank, ch, st))
        self._cbs[handler] = cb
        check(lib.mi_splitter_bank_bind_callback(self.handle, handler, cast(cb, c_void_p), None, None, _stream(stream)))

    def unbind(self, handler):
        check(lib.mi_splitter_bank_unbind(self.handle, handler))
        self._cbs.pop(handler, None)

    def clear(self, stream=None):
        check(lib.mi_splitter_bank_clear(self.handle, _stream(stream)))

    def process(self, outs, inp, count, out_stride=None, in_stride=None, stream=None):
        """outs: list of `handlers` device buffers or None (handler without a sink); inp None = silence."""
        arr = (c_void_p * self.handlers)(*[(_ptr(b) if b is not None else None) for b in outs])
        check(lib.mi_splitter_bank_process(self.handle, arr, _ptr(inp) if inp is not None else None, count,
                                           count if out_stride is None else out_stride,
                                           count if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, outs, inps, count, out_stride=None, in_stride=None, stream=None):
        """len(inps) consecutive process() calls in one C call; outs[k]: block k's list of `handlers` buffers (or None)."""
        n = len(inps)
        assert n == len(outs)
        flat = [(_ptr(b) if b is not None else None) for o in outs for b in o]
        assert len(flat) == n * self.handlers
        po = (c_void_p * len(flat))(*flat)
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_splitter_bank_process_blocks(self.handle, po, pi, n, count,
                                                  count if out_stride is None else out_stride,
                                                  count if in_stride is None else in_stride, _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_splitter_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def crossover_fft_mask(hpf, lpf, flatten, gain, sample_rate, rank):
    """FFTCrossover::update_band (FFTCrossover.cpp:459-486): hpf / lpf = (freq, slope) or None."""
    from ctypes import POINTER as _P
    n = 1 << rank
    m = np.empty(n, np.float32)
    p = m.ctypes.data_as(_P(c_float))
    if hpf is not None:
        lib.mi_crossover_hipass_fft_set(p, hpf[0], hpf[1], float(sample_rate), rank)
        if lpf is not None:
            lib.mi_crossover_lopass_fft_apply(p, lpf[0], lpf[1], float(sample_rate), rank)
    elif lpf is not None:
        lib.mi_crossover_lopass_fft_set(p, lpf[0], lpf[1], float(sample_rate), rank)
    else:
        m[:] = np.float32(flatten) * np.float32(gain)
        return m
    return (np.clip(m, np.float32(0.0), np.float32(flatten)) * np.float32(gain)).astype(np.float32)


class CrossoverBank:
    """lsp::dspu::Crossover for `channels` channels sharing the split settings (mi_crossover_bank_*)."""
    MODE_BT, MODE_MT = 0, 1

    def __init__(self, channels, bands):
        h = c_void_p()
        check(lib.mi_crossover_bank_create(byref(h), channels, bands))
        self.handle, self.channels, self.bands = h, channels, bands

    def set_sample_rate(self, sr):
        check(lib.mi_crossover_bank_set_sample_rate(self.handle, sr))

    def set_slope(self, split, slope):
        check(lib.mi_crossover_bank_set_slope(self.handle, split, slope))

    def set_frequency(self, split, freq):
        check(lib.mi_crossover_bank_set_frequency(self.handle, split, float(freq)))

    def set_mode(self, split, mode):
        check(lib.mi_crossover_bank_set_mode(self.handle, split, mode))

    def set_gain(self, band, gain):
        check(lib.mi_crossover_bank_set_gain(self.handle, band, float(gain)))

    def get_split(self, split):
        sl, fr, mo = c_uint32(), c_float(), c_int()
        check(lib.mi_crossover_bank_get_split(self.handle, split, byref(sl), byref(fr), byref(mo)))
        return {"slope": sl.value, "freq": fr.value, "mode": mo.value}

    def get_band(self, band, stream=None):
        g, s0, s1, act = c_float(), c_float(), c_float(), c_int()
        check(lib.mi_crossover_bank_get_band(self.handle, band, byref(g), byref(s0), byref(s1), byref(act), _stream(stream)))
        return {"gain": g.value, "start": s0.value, "end": s1.value, "active": bool(act.value)}

    def process(self, band_out, inp, samples, out_stride=None, in_stride=None, stream=None):
        """band_out: list of `bands` device buffers or None (band without a handler)."""
        arr = (c_void_p * self.bands)(*[(_ptr(b) if b is not None else None) for b in band_out])
        check(lib.mi_crossover_bank_process(self.handle, arr, _ptr(inp), samples,
                                            samples if out_stride is None else out_stride,
                                            samples if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, band_outs, inps, samples, out_stride=None, in_stride=None, stream=None):
        """len(inps) consecutive process() calls in one C call; band_outs[i]: block i's list of `bands` buffers (or None)."""
        n = len(inps)
        assert n == len(band_outs)
        flat = [(_ptr(b) if b is not None else None) for outs in band_outs for b in outs]
        assert len(flat) == n * self.bands
        po = (c_void_p * len(flat))(*flat)
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_crossover_bank_process_blocks(self.handle, po, pi, n, samples,
                                                   samples if out_stride is None else out_stride,
                                                   samples if in_stride is None else in_stride, _stream(stream)))

    def freq_chart(self, band, freqs, stream=None):
        import numpy as np
        f = np.ascontiguousarray(freqs, dtype=np.float32)
        c = np.empty(2 * f.size, np.float32)
        from ctypes import POINTER as _P
        check(lib.mi_crossover_bank_freq_chart(self.handle, band, c.ctypes.data_as(_P(c_float)),
                                               f.ctypes.data_as(_P(c_float)), f.size, _stream(stream)))
        return c[0::2] + 1j * c[1::2]

    def close(self):
        if self.handle:
            lib.mi_crossover_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EqualizerBank:
    """`channels` x lsp::dspu::Equalizer on the device."""
    BYPASS, IIR, FIR, FFT, SPM = range(5)

    def __init__(self, channels, filters, fir_rank):
        h = c_void_p()
        check(lib.mi_equalizer_bank_create(byref(h), channels, filters, fir_rank))
        self.handle, self.channels = h, channels

    def set_params(self, filter_id, ftype, slope=1, freq=1000.0, freq2=1000.0, gain=1.0, quality=0.0, channel=None):
        fp = FilterParams(int(ftype), int(slope), float(freq), float(freq2), float(gain), float(quality))
        check(lib.mi_equalizer_bank_set_params(self.handle, 0xFFFFFFFF if channel is None else channel, filter_id, byref(fp)))

    def set_mode(self, mode):
        check(lib.mi_equalizer_bank_set_mode(self.handle, mode))

    def set_sample_rate(self, sr):
        check(lib.mi_equalizer_bank_set_sample_rate(self.handle, sr))

    def get_latency(self, stream=None):
        v = c_uint32()
        check(lib.mi_equalizer_bank_get_latency(self.handle, byref(v), _stream(stream)))
        return v.value

    def set_smooth(self, smooth):
        """Equalizer::set_smooth: FIR/FFT retunes cross-fade over the block that completes next."""
        check(lib.mi_equalizer_bank_set_smooth(self.handle, 1 if smooth else 0))

    def reset(self, stream=None):
        check(lib.mi_equalizer_bank_reset(self.handle, _stream(stream)))

    def process(self, out, inp, samples, out_stride=None, in_stride=None, stream=None):
        check(lib.mi_equalizer_bank_process(self.handle, _ptr(out), _ptr(inp), samples,
                                            samples if out_stride is None else out_stride,
                                            samples if in_stride is None else in_stride, _stream(stream)))

    def process_blocks(self, outs, inps, samples, out_stride=None, in_stride=None, stream=None):
        """len(outs) consecutive process() calls issued by one C call (mi_equalizer_bank_process_blocks)."""
        n = len(outs)
        assert n == len(inps)
        po = (c_void_p * n)(*[_ptr(b) for b in outs])
        pi = (c_void_p * n)(*[_ptr(b) for b in inps])
        check(lib.mi_equalizer_bank_process_blocks(self.handle, po, pi, n, samples,
                                                   samples if out_stride is None else out_stride,
                                                   samples if in_stride is None else in_stride, _stream(stream)))

    def close(self):
        if self.handle:
            lib.mi_equalizer_bank_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
