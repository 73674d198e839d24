import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import optim
from posterior_matching_amd.engine import PMVAETrainStep
from posterior_matching_amd.models import PosteriorMatchingVAE
from tests.ref_configs import pm_vae_mnist
cfg = pm_vae_mnist(); xs=(28,28,1)
m = PosteriorMatchingVAE.from_config(cfg["model"], device="cuda:0"); m.init(xs)
opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(0.0), optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
ts = PMVAETrainStep(m, cfg, opt, 256, xs, use_graph=False)
ts.x.uniform_(); ts.b.fill_(1.0)
for _ in range(5): ts.step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(50): ts.step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/50:.3f} ms/step ; total {1e3*(t2-t0)/50:.3f} ms/step")
# host time to enqueue ONE step into idle queues (median of 30), and what the plan holds
import statistics
ds = []
for _ in range(30):
    torch.cuda.synchronize()
    t = time.perf_counter(); ts.step(); ds.append(time.perf_counter() - t)
torch.cuda.synchronize()
kinds = {}
for fn, args, name in ts._plan.calls:
    k = name if name in ("wait_stream", "event.record", "wait_event") else "launch"
    kinds[k] = kinds.get(k, 0) + 1
print(f"one step into idle queues: host {1e3 * statistics.median(ds):.3f} ms (min {1e3 * min(ds):.3f}); plan entries {kinds}")
