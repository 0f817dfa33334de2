"""Writes tests/golden/reference_public_api.json: for every reference header the hot path mirrors, the NAMES of its public
member functions and exported free functions (names only: an inventory, not source).  Run here, where /root/reference
exists; tests/test_api_surface.py checks the mirror headers against the committed list."""
import json
import os
import re
import sys

REF = "/root/reference/include/lsp-plug.in/dsp-units/"
HEADERS = ["filters/Filter.h", "filters/FilterBank.h", "filters/Equalizer.h", "filters/DynamicFilters.h", "util/Convolver.h",
           "util/SpectralProcessor.h", "util/MultiSpectralProcessor.h", "util/Analyzer.h", "util/Delay.h", "util/RingBuffer.h",
           "util/Crossover.h", "util/FFTCrossover.h", "util/SpectralSplitter.h", "meters/LoudnessMeter.h", "meters/ILUFSMeter.h",
           "misc/windows.h", "misc/envelope.h", "misc/fft_crossover.h", "misc/broadcast.h"]
KEYWORDS = {"if", "for", "while", "switch", "return", "sizeof", "defined", "operator"}


def public_names(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    names = set()
    # free functions: the declaration that follows an export marker
    for m in re.finditer(r"LSP_DSP_UNITS_PUBLIC\s+[^;{()]*?\b([A-Za-z_]\w*)\s*\(", text):
        if m.group(1) not in KEYWORDS and not m.group(0).lstrip().startswith("LSP_DSP_UNITS_PUBLIC\n            class"):
            names.add(m.group(1))
    # classes: walk the body, keep what sits in public sections
    for cm in re.finditer(r"\bclass\s+(?:LSP_DSP_UNITS_PUBLIC\s+)?(\w+)[^;{]*\{", text):
        cls, depth, i, access, start = cm.group(1), 1, cm.end(), "private", cm.end()
        body = []
        while i < len(text) and depth > 0:
            c = text[i]
            depth += (c == "{") - (c == "}")
            i += 1
        body = text[start:i - 1]
        # drop nested braces (inline bodies, nested types) so that only declarations at class level remain
        flat, d = [], 0
        for c in body:
            if c == "{":
                d += 1
            elif c == "}":
                d -= 1
                flat.append(";")
            elif d == 0:
                flat.append(c)
        for stmt in re.split(r"(public|protected|private)\s*:", "".join(flat)):
            if stmt in ("public", "protected", "private"):
                access = stmt
                continue
            if access != "public":
                continue
            for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\([^;]*?\)\s*(?:const)?\s*(?:;|$|=)", stmt):
                n = m.group(1)
                if n not in KEYWORDS and n != cls and not n.startswith("~"):
                    names.add(n)
    return sorted(names)


def main():
    out = {}
    for h in HEADERS:
        with open(os.path.join(REF, h)) as f:
            out[h] = public_names(f.read())
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_public_api.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("%d headers, %d names -> %s" % (len(out), sum(len(v) for v in out.values()), dst))


if __name__ == "__main__":
    sys.exit(main())
