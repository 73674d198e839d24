#!/usr/bin/env python3
"""Is the PM-VAE step limited by the clock the chip holds under sustained load?  Per-step HIP-event times of the same launch
plan (a) back to back, (b) with the host pausing between steps so that the device idles ~70 % of the time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from posterior_matching_amd import ops
from tools.workloads import build


def run(w, n, pause):
    evs = []
    for _ in range(n):
        w.feed()
        with torch.cuda.stream(w.ts.stream):
            e0 = ops.Event(); e0.record()
        w.step()
        with torch.cuda.stream(w.ts.stream):
            e1 = ops.Event(); e1.record()
        evs.append((e0, e1))
        if pause:
            w.synchronize()
            time.sleep(pause)
    w.synchronize()
    ts = sorted(a.elapsed_ms(b) for a, b in evs)
    return ts[len(ts) // 10], ts[len(ts) // 2], ts[-len(ts) // 10]


w = build(sys.argv[1] if len(sys.argv) > 1 else "pm_vae_mnist")
w.feed()
for _ in range(30):
    w.step()
w.synchronize()
for tag, n, pause in (("back to back", 400, 0.0), ("3 ms pause", 200, 0.003), ("back to back", 400, 0.0), ("10 ms pause", 100, 0.010)):
    p10, p50, p90 = run(w, n, pause)
    print(f"{tag:14s} per-step ms: p10 {p10:.4f}  median {p50:.4f}  p90 {p90:.4f}", flush=True)
