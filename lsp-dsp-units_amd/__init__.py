"""
lsp-dsp-units_amd -- MI355X (gfx950) implementation of the lsp-dsp-units
block-streaming hot path.

The product is the C-ABI shared library ``libmi_dspu.so`` built from ``csrc/``
(hand-written HIP kernels + C++ host glue, see ``include/mi_dspu.h``).  This
Python package is only a thin ctypes binding used by the tests and bench.py;
it contains no arithmetic and has NO CPU fallback: if the library is missing
the import fails, and without a HIP device every compute call raises.

Import with ``importlib.import_module("lsp-dsp-units_amd")`` (the directory
name carries the reference's name and is not a Python identifier).
"""
from .capi import LIB_PATH, MiError, check, lib          # noqa: F401
from .units import (AnalyzerBank, BiquadBank, Comm, ConvolverBank, CrossoverBank, DelayBank, DeviceBuffer, DynFilterBank, EqualizerBank,  # noqa: F401
                    ILUFSBank, LoudnessBank,
                    RingBank,
                    SpectralBank, SplitterBank, crossover_fft_mask,
                    design_filter, device_count, last_launch, last_stream_clock, source_sha, dynfilter_freq_chart, dynfilter_sections, filter_freq_chart, make_window, make_window_general)
