#!/usr/bin/env python
"""A small unifdef: resolve the preprocessor conditionals that depend ONLY on a given set of macros (defined with a
value, or known to be undefined) and leave every other conditional alone.  Used in round 5 to turn the adopted experiment
macros into plain code and to move the rejected ones out of csrc/ (tools/experiments/r05_removed_macros.patch).

  tools/unifdef.py -DNAME[=VALUE] ... -UNAME ... file ...      (rewrites the files in place)
"""
import re
import sys


def evaluate(expr, defs, undefs):
    """True / False when the expression only involves known macros, None otherwise."""
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"/\*.*?\*/", "", e)
    known = True

    def sub_defined(m):
        nonlocal known
        n = m.group(1)
        if n in defs:
            return "1"
        if n in undefs:
            return "0"
        known = False
        return "0"
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", sub_defined, e)

    def sub_name(m):
        nonlocal known
        n = m.group(0)
        if n in defs:
            return str(defs[n])
        if n in undefs:
            return "0"
        known = False
        return "0"
    e = re.sub(r"\b[A-Za-z_]\w*\b", sub_name, e)
    if not known:
        return None
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    try:
        return bool(eval(e, {"__builtins__": {}}))
    except Exception:
        return None


def process(text, defs, undefs):
    out = []
    # stack of frames: dict(known=bool, taken=bool (a branch was already emitted), active=bool (current branch emitted))
    stack = []
    emitting = lambda: all(f["active"] for f in stack)
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "if":
                v = evaluate(rest, defs, undefs)
            else:
                name = re.match(r"\s*(\w+)", rest).group(1)
                v = (name in defs) if (name in defs or name in undefs) else None
                if v is not None and kind == "ifndef":
                    v = not v
            if v is None:
                if emitting():
                    out.append(line)
                stack.append(dict(known=False, active=True, taken=True))
            else:
                stack.append(dict(known=True, active=v, taken=v))
        elif kind == "elif":
            f = stack[-1]
            if not f["known"]:
                if emitting():
                    out.append(line)
            else:
                v = evaluate(rest, defs, undefs)
                if f["taken"]:
                    f["active"] = False
                elif v is None:
                    raise SystemExit(f"unifdef: '#elif{rest}' mixes known and unknown macros")
                else:
                    f["active"] = f["taken"] = v
        elif kind == "else":
            f = stack[-1]
            if not f["known"]:
                if emitting():
                    out.append(line)
            else:
                f["active"] = not f["taken"]
                f["taken"] = True
        else:
            f = stack.pop()
            if not f["known"] and emitting():
                out.append(line)
    assert not stack, "unbalanced conditionals"
    return "\n".join(out)


def main():
    defs, undefs, files = {}, set(), []
    for a in sys.argv[1:]:
        if a.startswith("-D"):
            n, _, v = a[2:].partition("=")
            defs[n] = int(v) if v else 1
        elif a.startswith("-U"):
            undefs.add(a[2:])
        else:
            files.append(a)
    for f in files:
        s = open(f).read()
        t = process(s, defs, undefs)
        if t != s:
            open(f, "w").write(t)
            print(f"{f}: {s.count(chr(10)) - t.count(chr(10))} lines fewer")


if __name__ == "__main__":
    main()
