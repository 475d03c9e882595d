#!/usr/bin/env python3
"""One-off source transformation of round 6: take the lab out of the product kernels.

The streaming kernels carried their measurement apparatus inside their bodies - ablation switches (`#if defined(VICAN_WABLATE) &&
VICAN_WABLATE == 1` ...) and phase stamps (`WSTAMP(2);`) that compile to nothing in the shipped library.  This script rewrites
the sources as the preprocessor sees them with every lab macro UNDEFINED (the `#else` branches are kept, the stamp calls go, the
phase comments behind them stay as comments), so the product text is the product.  The instrumented text is kept as reverse
patches under tools/lab_patches/ (made with `git diff -R` after this ran): `tools/build_variants.py --patch NAME` applies one to
a scratch copy and builds the diagnostic library from it.

    python tools/strip_lab.py vican_amd/csrc/vican_wsweep.hip ...      # in place
"""
import re
import sys

LAB = {"VICAN_WABLATE", "VICAN_WSTAMP", "VICAN_CGWABLATE", "VICAN_CGWSTAMP", "VICAN_ABLATE", "VICAN_STAMP", "VICAN_CGRSTAMP", "COOP_STAMP",
       "RITZ_STAMP", "VICAN_LRSTAMP", "VICAN_CG_HAND_RELEASE", "VICAN_CG_HAND_ACQUIRE"}
STAMP_CALL = re.compile(r"^(\s*)(?:WSTAMP0|WSTAMP|CSTAMP0|CSTAMP|STAMP0|STAMP|RSTAMP|LSTAMP)\(\s*\d*\s*\);\s*(//.*)?$")
STAMP_DEF = re.compile(r"^\s*#\s*define\s+(?:WSTAMP0|WSTAMP|CSTAMP0|CSTAMP|STAMP0|STAMP|RSTAMP|LSTAMP)\b")


def lab_value(cond):
    """None if the condition does not involve a lab macro; else its truth value with the lab macros undefined."""
    if not any(m in cond for m in LAB):
        return None
    c = cond
    # defined(X) / defined X -> 0 for lab macros
    c = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "0" if m.group(1) in LAB else m.group(0), c)
    c = re.sub(r"defined\s+(\w+)", lambda m: "0" if m.group(1) in LAB else m.group(0), c)
    for m in LAB:
        c = re.sub(r"\b%s\b" % m, "0", c)                      # an undefined identifier evaluates to 0 in #if
    if re.search(r"[A-Za-z_]", c.replace("&&", "").replace("||", "")):
        raise SystemExit("mixed condition, not handled: " + cond)
    c = c.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(c))


def strip(text):
    out, stack = [], []        # stack entries: [kind, emitting_before, taken_already]; kind "lab" or "keep"
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        emitting = all(e[1] for e in stack)
        if m:
            d, rest = m.group(1), re.sub(r"/\*.*?\*/", "", m.group(2)).strip()
            if d in ("ifdef", "ifndef", "if"):
                if d == "if":
                    v = lab_value(rest)
                else:
                    name = rest.split()[0]
                    v = None if name not in LAB else (d == "ifndef")
                if v is None:
                    stack.append(["keep", True, True])
                    if emitting:
                        out.append(line)
                else:
                    stack.append(["lab", v, v])
                continue
            top = stack[-1]
            if d == "elif":
                if top[0] == "keep":
                    if emitting or all(e[1] for e in stack[:-1]):
                        out.append(line)
                else:
                    v = lab_value(rest)
                    if v is None:
                        raise SystemExit("#elif with a non-lab condition inside a lab #if: " + line)
                    top[1] = (not top[2]) and v
                    top[2] = top[2] or v
                continue
            if d == "else":
                if top[0] == "keep":
                    if all(e[1] for e in stack[:-1]):
                        out.append(line)
                else:
                    top[1] = not top[2]
                    top[2] = True
                continue
            if d == "endif":
                stack.pop()
                if top[0] == "keep" and all(e[1] for e in stack):
                    out.append(line)
                continue
        if not emitting:
            continue
        if STAMP_DEF.match(line):
            continue
        c = STAMP_CALL.match(line)
        if c:
            if c.group(2):
                out.append(c.group(1) + c.group(2))            # the phase comment stays
            continue
        out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        if new != src:
            open(path, "w").write(new)
            print("%s: %d -> %d lines" % (path, src.count("\n"), new.count("\n")))
