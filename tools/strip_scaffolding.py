#!/usr/bin/env python3
"""Resolve the experiment switches of a kernel source at their shipped values and drop the dead branches.

    tools/strip_scaffolding.py in.hip out.hip NAME=value ... [-U NAME ...]

Handles #if / #ifdef / #ifndef / #elif / #else / #endif whose condition uses ONLY the named macros (integer
comparisons, !, &&, ||, defined()); every other conditional is kept verbatim.  `#ifndef NAME / #define NAME v /
#endif` default blocks of the named macros are dropped.  Used once in round 4 to take the round-3 timing ablations,
phase stamps and rejected variants (RGFM_HX2P_ABL / _PROF / _PRIO / _QEXP / _CHUNK_EXP / _GNEARLY / _FAST / _TAIL,
RGFM_HX2Q_*) out of the shipped kernels; the frozen originals live in tools/kbench/variants/.
"""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    defs, undef = {}, set()
    it = iter(sys.argv[3:])
    for a in it:
        if a == "-U":
            undef.add(next(it))
        else:
            k, v = a.split("=")
            defs[k] = int(v)
    known = set(defs) | undef

    def evaluate(expr):
        """value of a condition, or None if it mentions anything unknown"""
        e = re.sub(r"//.*", "", expr).strip()
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("1" if m.group(1) in defs else "0") if m.group(1) in known else m.group(0), e)
        names = set(re.findall(r"[A-Za-z_]\w*", e))
        if not names <= known:
            return None
        for n in names:
            e = re.sub(rf"\b{n}\b", str(defs.get(n, 0)), e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        return bool(eval(e))

    lines = open(src).read().split("\n")
    out = []
    # stack entries: [known, emitting_parent, taken_already, currently_emitting]
    stack = []

    def emitting():
        return all(s[3] for s in stack)

    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r"#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                # uses of a profiling macro that is defined empty
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "ifdef":
                nm = rest.split()[0]
                val = (nm in defs) if nm in known else None
            elif kind == "ifndef":
                nm = rest.split()[0]
                val = (nm not in defs) if nm in known else None
                # default block "#ifndef X / #define X v / #endif"
                if nm in known and i + 2 < len(lines) and re.match(rf"#\s*define\s+{nm}\b", lines[i + 1].strip()) and lines[i + 2].strip().startswith("#endif"):
                    i += 3
                    continue
            else:
                val = evaluate(rest)
            if val is None:
                stack.append([False, True, True, True])
                if emitting():
                    out.append(ln)
            else:
                stack.append([True, True, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if emitting():
                    out.append(ln)
            else:
                val = evaluate(rest)
                assert val is not None, ("mixed #elif", ln)
                top[3] = (not top[2]) and val
                top[2] = top[2] or val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if emitting():
                    out.append(ln)
            else:
                top[3] = not top[2]
                top[2] = True
        else:  # endif
            top = stack.pop()
            if not top[0] and emitting():
                out.append(ln)
        i += 1
    assert not stack
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
