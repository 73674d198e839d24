#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel name (sum over dispatches)."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = name.split(">(")[0] + (">" if ">(" in name else "")
        name = name.split("(")[0] if ">" not in name else name
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
names = sorted(rows, key=lambda n: -rows[n].get("SQ_WAVE_CYCLES", rows[n].get("GRBM_GUI_ACTIVE", 0)))
counters = sorted({c for n in rows for c in rows[n]})
print("kernel".ljust(50), "calls", *[c.rjust(24) for c in counters])
for n in names[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(n[:50].ljust(50), str(len(calls[n])).rjust(5), *[f"{rows[n].get(c, 0):24.0f}" for c in counters])
