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
