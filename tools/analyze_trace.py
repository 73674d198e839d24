#!/usr/bin/env python3
"""Per-queue busy time and critical-path view of a rocprofv3 --kernel-trace CSV of bench.py (concurrent mode).
    python tools/analyze_trace.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ks.sort()
# last `steps` optimizer steps: delimited by adam kernels
adam = [i for i, k in enumerate(ks) if "adam" in k[3]]
lo, hi = adam[-steps - 1], adam[-1]
win = ks[lo + 1:hi + 1]
t0, t1 = win[0][0], win[-1][1]
span = (t1 - t0) / steps / 1e3
busy = defaultdict(float)
for s, e, q, n in win:
    busy[q] += (e - s) / 1e3
print(f"span per step {span:.1f} us; kernels per step {len(win) / steps:.0f}")
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: busy {b / steps:8.1f} us/step ({100 * b / steps / span:4.1f} %)")
# union busy (any queue busy) and concurrency histogram
ev = []
for s, e, q, n in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, t0, defaultdict(float)
for t, d in ev:
    hist[cur] += (t - last) / 1e3
    cur += d
    last = t
for c in sorted(hist):
    print(f"{c} kernels in flight: {hist[c] / steps:8.1f} us/step")
# timeline of the last step: start offset, duration, queue, kernel
if len(sys.argv) > 3:
    last_lo = adam[-2]
    step = ks[last_lo + 1:hi + 1]
    s0 = step[0][0]
    qid = {q: i for i, q in enumerate(sorted({k[2] for k in step}))}
    with open(sys.argv[3], "w") as f:
        for s, e, q, n in step:
            f.write(f"{(s - s0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{qid[q]}  {n[:90]}\n")
