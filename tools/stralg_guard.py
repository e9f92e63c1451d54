#!/usr/bin/env python3
"""The change a stralg maintainer makes to put libstralg_amd.so behind libstralg, as a script.

    python tools/stralg_guard.py <reference checkout> <out dir> [--table]

writes copies of the reference's stralg/*.c and bioinf/*.c (headers are copied unchanged) into
<out dir>/stralg and <out dir>/bioinf in which every function that libstralg_amd.so defines is
wrapped in

    #ifndef STRALG_WITH_MI355X
    ...
    #endif

so that, compiled with -DSTRALG_WITH_MI355X and linked against libstralg_amd.so, each
reference-named symbol is defined exactly once: in libstralg_amd.so when the GPU library
provides it, in libstralg otherwise.  Nothing else in the reference changes.  --table prints
the guarded regions (file, lines, function) as the markdown table of INTEGRATION.md.

The reference's sources are read where they lie; the patched copies are build products of
tests/test_abi.py's link test (written under a temporary directory, never committed).
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "stralg_amd", "libstralg_amd.so")
GUARD = "STRALG_WITH_MI355X"

# a definition starts at column 0 with its return type (possibly on a line of its own) and ends at the first "}"
# at column 0: the layout of every function in the reference
DEF = re.compile(r"^(?!static\b)(?:[A-Za-z_][\w\s\*]*?[\s\*])??([A-Za-z_]\w*)\s*\(")
BARE_TYPE = re.compile(r"^[A-Za-z_][\w\s]*\*?\s*$")


def exported(lib=PRODUCT):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {line.split()[2] for line in out.splitlines() if len(line.split()) == 3 and line.split()[1] in "TW"}


def regions(lines, names):
    """[(first, last, name)] 0-based inclusive line ranges of the definitions of `names` in a C file"""
    found = []
    i = 0
    while i < len(lines):
        line = lines[i]
        m = DEF.match(line) if line[:1].isalpha() or line[:1] == "_" else None
        name = m.group(1) if m else None
        if name and name in names and not line.rstrip().endswith(";"):
            # a prototype that spans lines ends in ");" before any "{": skip those
            j = i
            is_def = False
            while j < len(lines):
                if "{" in lines[j]:
                    is_def = True
                    break
                if lines[j].rstrip().endswith(";"):
                    break
                j += 1
            if is_def:
                end = j
                while end < len(lines) and not lines[end].startswith("}"):
                    end += 1
                first = i
                if i > 0 and BARE_TYPE.match(lines[i - 1]) and not lines[i - 1].strip().endswith(("else", "do")):
                    first = i - 1
                found.append((first, end, name))
                i = end + 1
                continue
        i += 1
    return found


def guard_file(src, dst, names):
    with open(src, encoding="utf-8", errors="surrogateescape") as f:
        lines = f.read().split("\n")
    regs = regions(lines, names)
    out, at = [], 0
    for first, last, _ in regs:
        out += lines[at:first]
        out.append(f"#ifndef {GUARD} /* provided by libstralg_amd.so */")
        out += lines[first:last + 1]
        out.append(f"#endif /* {GUARD} */")
        at = last + 1
    out += lines[at:]
    with open(dst, "w", encoding="utf-8", errors="surrogateescape") as f:
        f.write("\n".join(out))
    return regs


def guard_tree(ref, out_dir, names=None):
    names = names or exported()
    report = []
    for sub in ("stralg", "bioinf"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
        for fn in sorted(os.listdir(os.path.join(ref, sub))):
            src, dst = os.path.join(ref, sub, fn), os.path.join(out_dir, sub, fn)
            if fn.endswith(".c"):
                for first, last, name in guard_file(src, dst, names):
                    report.append((f"{sub}/{fn}", first + 1, last + 1, name))
            elif fn.endswith(".h"):
                shutil.copyfile(src, dst)
    return report


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    report = guard_tree(sys.argv[1], sys.argv[2])
    if "--table" in sys.argv:
        print("| reference file | lines | function now resolved from libstralg_amd.so |")
        print("|---|---|---|")
        for fn, a, b, name in report:
            print(f"| `{fn}` | {a}-{b} | `{name}` |")
    else:
        for fn, a, b, name in report:
            print(f"{fn}:{a}-{b} {name}")


if __name__ == "__main__":
    main()
