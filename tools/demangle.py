#!/usr/bin/env python
"""Readable names for this library's kernels in rocprofv3 CSVs (developer tool).  rocprofv3 / c++filt on this image leave names whose
template arguments hold __bf16 (DF16b) or _Float16 (DF16_) mangled; the kernels' template arguments are integers, booleans and these
types, which is all this decodes.      python tools/demangle.py < in.csv > out.csv"""
import re
import sys

PAT = re.compile(r"_ZN5svhip12_GLOBAL__N_1(\d+)([A-Za-z0-9_]+?)I((?:Li\d+E|Ln\d+E|Lb[01]E|DF16b|DF16_|f)+)E(?:Ev[A-Za-z0-9_]*)")
TOK = re.compile(r"Li(\d+)E|Ln(\d+)E|Lb([01])E|(DF16b)|(DF16_)|(f)")


def pretty(m):
    n = int(m.group(1))
    name = m.group(2)
    if len(name) != n:          # the length prefix says where the name ends
        whole = m.group(2) + "I" + m.group(3)
        name, rest = whole[:n], whole[n:]
        if not rest.startswith("I"):
            return m.group(0)
        args_s = rest[1:]
    else:
        args_s = m.group(3)
    args = []
    for t in TOK.finditer(args_s):
        if t.group(1) is not None:
            args.append(t.group(1))
        elif t.group(2) is not None:
            args.append("-" + t.group(2))
        elif t.group(3) is not None:
            args.append("true" if t.group(3) == "1" else "false")
        elif t.group(4):
            args.append("__bf16")
        elif t.group(5):
            args.append("_Float16")
        else:
            args.append("float")
    return "svhip::%s<%s>" % (name, ", ".join(args))


def demangle(text):
    return PAT.sub(pretty, text)


if __name__ == "__main__":
    sys.stdout.write(demangle(sys.stdin.read()))
