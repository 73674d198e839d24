#!/usr/bin/env python3
"""A few optimizer steps of one workload (tools/workloads.py), for profiler passes:  python3 tools/run_steps.py <workload> [steps] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.workloads import build

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
w = build(name, batch)
w.feed()
for _ in range(steps):
    w.step()
w.synchronize()
print("done", name, steps)
