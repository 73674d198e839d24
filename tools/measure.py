#!/usr/bin/env python3
"""One line per run: throughput of one of BASELINE.json's training steps (tools/workloads.py), for A/B runs under
environment switches.   python tools/measure.py pm_vqvae_mnist [batch] [steps] [warmup]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.workloads import build, measure  # noqa: E402

name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 4
print(json.dumps(measure(build(name, batch), steps, warm)))
