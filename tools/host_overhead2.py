"""host time to enqueue ONE step of a workload into idle queues vs the step's device time
   python tools/host_overhead2.py pm_vdvae_mnist 16"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.workloads import build
w = build(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
w.feed()
for _ in range(5): w.step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter(); w.step(); h = time.perf_counter() - t
    torch.cuda.synchronize(); d = time.perf_counter() - t
    hs.append(h); ts.append(d)
kinds = {}
plan = getattr(w.ts, "_plan", None)
for fn, args, name in (plan.calls if plan else []):
    k = name if name in ("wait_stream", "event.record", "wait_event") else "launch"
    kinds[k] = kinds.get(k, 0) + 1
print(f"{sys.argv[1]}: host enqueue {1e3 * statistics.median(hs):.3f} ms, step (sync to sync) {1e3 * statistics.median(ts):.3f} ms; plan {kinds}")
