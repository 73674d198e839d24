import sys, torch
sys.path.insert(0, "/root/repo")
from tools.workloads import build
for name, B in (("pm_vqvae_celeb_a", 16), ("vqvae_mnist", 256), ("pm_vae_gas", 128)):
    finals = []
    for _ in range(2):
        w = build(name, B)
        w.feed()
        for _ in range(3):
            w.step()
        w.synchronize()
        st = w.ts.store if hasattr(w.ts, "store") else w.ts.model.store
        finals.append(st.flat_p.clone())
        del w
        torch.cuda.empty_cache()
    d = (finals[0] - finals[1]).abs().max().item()
    print(name, B, "bit-identical" if torch.equal(finals[0], finals[1]) else f"differs (max abs {d:.3e})", flush=True)
