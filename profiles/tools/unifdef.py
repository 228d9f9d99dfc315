#!/usr/bin/env python3
"""unifdef.py FILE -DNAME[=VALUE] ... -UNAME ... : resolve the preprocessor conditionals of FILE that depend only on the
given symbols (there is no unifdef in the image).  A conditional whose expression mentions anything else is kept as it
is.  Used once, in round 5, to take the experiment switches that lost out of grt_render_tile.hip / grt_device.h (they
live on as profiles/tools/r04_experiments_removed.patch); the result is checked by comparing the default build's device
assembly before and after (byte-identical apart from the __hip_cuid symbol)."""
import re
import sys


def tri_eval(expr, defs, undefs):
    """True / False when the expression is decided by the known symbols, None otherwise."""
    e = re.sub(r"/\*.*?\*/", " ", expr)
    e = re.sub(r"//.*", " ", e).strip()
    unknown = [False]

    def rep_defined(m):
        n = m.group(1) or m.group(2)
        if n in defs:
            return " 1 "
        if n in undefs:
            return " 0 "
        unknown[0] = True
        return " 0 "

    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", rep_defined, e)

    def rep_ident(m):
        n = m.group(0)
        if n in ("and", "or", "not"):
            return n
        if n in defs:
            return str(defs[n] if defs[n] != "" else 1)
        if n in undefs:
            return "0"
        unknown[0] = True
        return "0"

    e = re.sub(r"\b[A-Za-z_]\w*\b", rep_ident, e)
    if unknown[0]:
        return None
    e = re.sub(r"(\d+)[uUlL]+\b", r"\1", e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    try:
        return bool(eval(e, {"__builtins__": {}}, {}))
    except Exception:
        return None


def process(lines, defs, undefs):
    out = []
    # stack entries: [kind, emitting_parent, state, kept]
    #   kind 'r' = resolved conditional (its directives vanish), 'k' = kept as written
    #   state for 'r': 'taken' (a branch is being emitted), 'done' (a branch was emitted already), 'wait' (none yet)
    stack = []

    def emitting():
        for kind, par, st, _ in stack:
            if kind == "r" and st != "taken":
                return False
        return True

    for ln in lines:
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(ln)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("ifdef", "ifndef", "if"):
            if d == "ifdef":
                v = tri_eval("defined(%s)" % rest.split()[0], defs, undefs)
            elif d == "ifndef":
                v = tri_eval("defined(%s)" % rest.split()[0], defs, undefs)
                v = None if v is None else (not v)
            else:
                v = tri_eval(rest, defs, undefs)
            par = emitting()
            if v is None or not par:
                stack.append(["k", par, None, None])
                if par:
                    out.append(ln)
            else:
                stack.append(["r", par, "taken" if v else "wait", None])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "k":
                if emitting():
                    out.append(ln)
            else:
                if top[2] == "taken":
                    top[2] = "done"
                elif top[2] == "wait":
                    v = tri_eval(rest, defs, undefs)
                    if v is None:  # every branch so far was decided false: the rest becomes a conditional of its own
                        top[0] = "k"
                        if emitting():
                            out.append(ln.replace("elif", "if", 1))
                    else:
                        top[2] = "taken" if v else "wait"
        elif d == "else":
            top = stack[-1]
            if top[0] == "k":
                if emitting():
                    out.append(ln)
            else:
                top[2] = "taken" if top[2] == "wait" else "done"
        else:
            top = stack.pop()
            if top[0] == "k" and top[1] and emitting():
                out.append(ln)
    assert not stack
    return out


def main():
    path = sys.argv[1]
    defs, undefs = {}, set()
    for a in sys.argv[2:]:
        if a.startswith("-D"):
            n, _, v = a[2:].partition("=")
            defs[n] = v
        elif a.startswith("-U"):
            undefs.add(a[2:])
    lines = open(path).read().split("\n")
    open(path, "w").write("\n".join(process(lines, defs, undefs)))


if __name__ == "__main__":
    main()
