"""Writes tests/golden/dump_keys.json: for every lsp::dspu class of the hot path, the names its dump() hands to the
IStateDumper, in source order (data for tests/test_cpp_classes.py::test_dump_keys_match_the_reference; run in the build
container, where /root/reference exists:  python tests/golden/make_dump_keys.py)."""
import json
import os
import re

REF = "/root/reference/src/main"
FILES = {"FilterBank": "filters/FilterBank.cpp", "Filter": "filters/Filter.cpp", "Equalizer": "filters/Equalizer.cpp",
         "DynamicFilters": "filters/DynamicFilters.cpp", "Convolver": "util/Convolver.cpp",
         "SpectralProcessor": "util/SpectralProcessor.cpp", "MultiSpectralProcessor": "util/MultiSpectralProcessor.cpp",
         "Crossover": "util/Crossover.cpp", "SpectralSplitter": "util/SpectralSplitter.cpp", "FFTCrossover": "util/FFTCrossover.cpp",
         "LoudnessMeter": "meters/LoudnessMeter.cpp", "ILUFSMeter": "meters/ILUFSMeter.cpp", "Delay": "util/Delay.cpp",
         "RingBuffer": "util/RingBuffer.cpp", "Analyzer": "util/Analyzer.cpp"}


def dump_body(text, cls):
    m = re.search(r"void %s::dump\((?:dspu::)?IStateDumper \*v\) const\s*\{" % cls, text)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i]


def keys(body):
    return re.findall(r'v->\w+\(\s*"(\w+)"', body)


if __name__ == "__main__":
    out = {cls: keys(dump_body(open(os.path.join(REF, f)).read(), cls)) for cls, f in FILES.items()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dump_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: len(v) for k, v in out.items()})
