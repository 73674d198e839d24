#!/usr/bin/env python3
"""Timeline of the CONCURRENT PM-VAE step without a profiler: a one-thread pm_stamp kernel behind every launch of the recorded
launch plan stores the device's 100 MHz real-time counter, so each stamp is the END time of the kernel in front of it on its
stream (plus ~2 us of launch boundary).  rocprofv3 --kernel-trace slows the host's launches enough to change how the two
streams interleave (the profiled step is 1.54 ms, the plain one 1.37 ms); the stamps cost one tiny launch per kernel.
    python tools/stamp_timeline.py [out.txt] [workload]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from posterior_matching_amd import ops
from tools.workloads import build


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    name = sys.argv[2] if len(sys.argv) > 2 else "pm_vae_mnist"
    w = build(name)
    w.feed()
    w.step()                     # eager: allocates
    w.synchronize()
    ops.stamps_begin(8192)
    for _ in range(30):          # step 2 records the plan (with the stamps), later steps replay it
        w.feed()
        w.step()
    w.synchronize()
    rows = ops.stamps_end()
    # the recording step issued every stamp once; replays overwrite the same slots: the values are those of the LAST step
    streams = {}
    t0 = min(v for _, _, v in rows if v)
    with open(out, "w") as fp:
        prev = {}
        for n, s, v in rows:
            q = streams.setdefault(s, f"q{len(streams)}")
            t = (v - t0) / 100.0
            fp.write(f"{t:9.1f} us  {q}  +{t - prev.get(s, t):7.1f}  {n}\n")
            prev[s] = t
        fp.write(f"span {(max(v for _, _, v in rows) - t0) / 100.0:.1f} us, {len(rows)} stamped launches\n")


if __name__ == "__main__":
    main()
