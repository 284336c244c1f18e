#!/usr/bin/env python
""" one-screen view of a bench.py line on stdin: value, per-kernel rooflines, extra legs """
import json
import sys

d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("value_no_overlap"))
for rr in d.get("rooflines", []):
    print("   ", rr.get("kernel", "")[:60], rr.get("avg_launch_ms"), rr.get("frac"))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print("   ", k, v.get("value"), v.get("ms_per_step"))
