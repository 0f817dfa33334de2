"""Writes tests/golden/reference_signatures.json: the public, out-of-line member functions and exported free functions of
the reference headers the hot path mirrors, as normalised signature strings in the form c++filt prints them with the
namespace qualifiers removed ("Filter::update(unsigned long, filter_params_t const*)").  An inventory of declarations
(names and parameter types), generated where /root/reference exists; tests/test_api_surface.py demangles what
libmi_dspu.so exports and looks every one of them up."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_api_list import REF, HEADERS, KEYWORDS

BASIC = {"size_t": "unsigned long", "ssize_t": "long", "status_t": "int", "uint32_t": "unsigned int", "int32_t": "int",
         "uint8_t": "unsigned char", "uint64_t": "unsigned long", "int64_t": "long", "wsize_t": "unsigned long",
         "wssize_t": "long"}
# function-pointer typedefs of the headers, expanded as the demangler prints them (filled by main())
FUNC = {}


def norm_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"=.*$", "", t).strip()                       # default value
    # drop the parameter name: the last identifier if what precedes it still names a type
    m = re.match(r"^(.*?[\*&\s])([A-Za-z_]\w*)(\s*\[\s*\])?$", t)
    if m and re.search(r"[A-Za-z_]", m.group(1)) and m.group(2) not in ("const", "int", "float", "double", "char", "long", "short", "unsigned", "bool", "void") \
            and not re.fullmatch(r"\s*(const|unsigned|signed)?\s*", m.group(1)):
        t = m.group(1).strip() + ("*" if m.group(3) else "")
    t = re.sub(r"\b(\w+::)+", "", t)                         # namespace / class qualifiers
    toks = t.replace("*", " * ").replace("&", " & ").split()
    # "const T *" -> "T const*"
    if toks and toks[0] == "const" and len(toks) > 1:
        toks = [toks[1], "const"] + toks[2:]
    out = []
    for k in toks:
        out.append(FUNC.get(k, BASIC.get(k, k)))
    s = " ".join(out)
    if s.endswith(" const") and "*" not in s and "&" not in s:      # top-level const of a by-value parameter does not mangle
        s = s[:-6]
    s = re.sub(r"\s*\*", "*", s)
    s = re.sub(r"\s*&", "&", s)
    s = re.sub(r"\*\s+const", "* const", s)
    return s.strip()


def split_params(p):
    p = p.strip()
    if p in ("", "void"):
        return []
    parts, depth, cur = [], 0, ""
    for c in p:
        if c in "(<":
            depth += 1
        elif c in ")>":
            depth -= 1
        if c == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += c
    parts.append(cur)
    return [norm_type(x) for x in parts]


def signatures(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    sigs = set()
    for m in re.finditer(r"LSP_DSP_UNITS_PUBLIC\s+([^;{()]*?)\b([A-Za-z_]\w*)\s*\(([^;{]*?)\)\s*;", text):
        if m.group(2) in KEYWORDS or "class" in m.group(1):
            continue
        ns = re.findall(r"namespace\s+(\w+)", text[:m.start()])
        scope = ns[-1] if ns and ns[-1] not in ("lsp", "dspu") else ""
        sigs.add(("%s::" % scope if scope else "") + "%s(%s)" % (m.group(2), ", ".join(split_params(m.group(3)))))
    for cm in re.finditer(r"\bclass\s+(?:LSP_DSP_UNITS_PUBLIC\s+)?(\w+)[^;{]*\{", text):
        cls, depth, i = cm.group(1), 1, cm.end()
        while i < len(text) and depth > 0:
            depth += (text[i] == "{") - (text[i] == "}")
            i += 1
        body = text[cm.end():i - 1]
        flat, d = [], 0
        for c in body:                                       # inline bodies become a marker, nested types vanish
            if c == "{":
                d += 1
                if d == 1:
                    flat.append(" @INLINE@ ")
            elif c == "}":
                d -= 1
                if d == 0:
                    flat.append(";")
            elif d == 0:
                flat.append(c)
        access = "private"
        for stmt in re.split(r"\b(public|protected|private)\s*:", "".join(flat)):
            if stmt in ("public", "protected", "private"):
                access = stmt
                continue
            if access != "public":
                continue
            for decl in stmt.split(";"):
                if "@INLINE@" in decl or "= delete" in decl or "inline" in decl.split("(")[0]:
                    continue
                m = re.search(r"([A-Za-z_~]\w*)\s*\((.*)\)\s*(const)?\s*$", decl.strip(), flags=re.S)
                if not m or m.group(1) in KEYWORDS or "operator" in decl:
                    continue
                name = m.group(1)
                sigs.add("%s::%s(%s)%s" % (cls, name, ", ".join(split_params(m.group(2))), " const" if m.group(3) else ""))
    return sorted(sigs)


def main():
    out = {}
    for h in HEADERS:                                        # typedef void (* name)(parameters);
        with open(os.path.join(REF, h)) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        for m in re.finditer(r"typedef\s+(\w[\w\s\*]*?)\(\s*\*\s*(\w+)\s*\)\s*\(([^;]*?)\)\s*;", text):
            FUNC[m.group(2)] = "%s (*)(%s)" % (norm_type(m.group(1) + " x").strip(), ", ".join(split_params(m.group(3))))
    for h in HEADERS:
        with open(os.path.join(REF, h)) as f:
            out[h] = signatures(f.read())
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("%d headers, %d signatures -> %s" % (len(out), sum(len(v) for v in out.values()), dst))


if __name__ == "__main__":
    sys.exit(main())
