#!/usr/bin/env python
"""
Every TOCVP_* environment knob the package and bench.py read, with its default and where it is read:
    python scripts/list_knobs.py            # table on stdout (KNOBS.md holds the annotated copy)
    python scripts/list_knobs.py --check    # exit 1 when a knob in the sources is missing from KNOBS.md (CPU test)
"""
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PAT = re.compile(r'(?:os\.environ\.get|_knob|knob|getenv)\(\s*"(TOCVP_[A-Z0-9_]+)"(?:\s*,\s*("?[^")]*"?))?')


def scan():
    found = {}
    files = [os.path.join(ROOT, "bench.py")]
    for d, _, names in os.walk(os.path.join(ROOT, "textocvp_amd")):
        files += [os.path.join(d, n) for n in names if n.endswith((".py", ".hip", ".h"))]
    for f in sorted(files):
        for i, line in enumerate(open(f, errors="replace"), 1):
            for m in PAT.finditer(line):
                name, default = m.group(1), (m.group(2) or "").strip('"')
                found.setdefault(name, {"default": default, "where": []})["where"].append(f"{os.path.relpath(f, ROOT)}:{i}")
    return found


if __name__ == "__main__":
    knobs = scan()
    if "--check" in sys.argv:
        doc = open(os.path.join(ROOT, "KNOBS.md")).read()
        missing = [k for k in knobs if f"`{k}`" not in doc]
        stale = [k for k in re.findall(r"`(TOCVP_[A-Z0-9_]+)`", doc) if k not in knobs and f"~~`{k}`~~" not in doc]
        if missing or stale:
            print("missing from KNOBS.md:", missing, "\nin KNOBS.md but not in the sources:", sorted(set(stale)))
            sys.exit(1)
        print(f"{len(knobs)} knobs, all documented")
        sys.exit(0)
    for k, v in sorted(knobs.items()):
        print(f"| `{k}` | `{v['default']}` | {', '.join(v['where'])} |")
