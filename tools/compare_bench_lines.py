#!/usr/bin/env python3
"""Every number of two bench.py lines side by side (new / old - 1 beyond 3 %): the check that a change to one leg has not moved another.
usage: python tools/compare_bench_lines.py old.json new.json"""
import json, sys


def flat(d, pre=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat(v, pre + k + "."))
        elif isinstance(v, (int, float)) and not isinstance(v, bool):
            out[pre + k] = float(v)
    return out


a, b = (flat(json.load(open(p))) for p in sys.argv[1:3])
for k in sorted(set(a) & set(b)):
    if any(s in k for s in ("gibs", "value", "ms", "frac")) and a[k] and abs(b[k] / a[k] - 1.0) > 0.03:
        print(f"{k:70s} {a[k]:12.3f} -> {b[k]:12.3f}  {100.0 * (b[k] / a[k] - 1.0):+6.1f} %")
