"""Which sources a kernel's counters were measured on: the hashes (mi_dspu_source_sha) of the kernel's .hip file and of every
header of csrc/, as the LOADED library was built from them.  The profile summaries under profiles/ carry this as "sources";
bench.py quotes a committed counter only while the library it runs was built from the same sources."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FILE_OF = (("biquad_stream_chain", "biquad.hip"), ("biquad_chain", "biquad.hip"), ("biquad_", "biquad.hip"), ("conv_", "convolver.hip"),
           ("analyzer_", "spectral.hip"), ("bin_", "spectral.hip"), ("stft_", "spectral.hip"), ("splitter_", "splitter.hip"),
           ("loudness_", "loudness.hip"), ("ilufs_", "loudness.hip"), ("dyn", "dynfilter.hip"), ("crossover_", "crossover.hip"),
           ("delay_", "delay.hip"), ("ring_", "delay.hip"))
HEADERS = ("mi_common.h", "fft_device.h", "fft16.h", "fft_wave.h", "ilufs_device.h")


def file_of(kernel):
    k = kernel.replace("void ", "").replace("(anonymous namespace)::", "").lstrip("(")
    for prefix, f in FILE_OF:
        if k.startswith(prefix):
            return f
    return None


def sources_for(kernel_or_file, mi=None):
    """{file: sha16} for the kernel's .hip (or the file named) and the headers, from the loaded library."""
    mi = mi or importlib.import_module("lsp-dsp-units_amd")
    f = kernel_or_file if kernel_or_file.endswith(".hip") else file_of(kernel_or_file)
    out = {}
    for name in ((f,) if f else ()) + HEADERS:
        sha = mi.source_sha(name)
        if sha:
            out[name] = sha
    return out


def current(doc_sources, mi=None):
    """True if every recorded hash equals the loaded library's (and something was recorded at all)."""
    if not doc_sources:
        return False
    mi = mi or importlib.import_module("lsp-dsp-units_amd")
    return all(mi.source_sha(f) == sha for f, sha in doc_sources.items())


if __name__ == "__main__":
    import json
    print(json.dumps(sources_for(sys.argv[1])))
