#!/usr/bin/env python3
"""bf16x3 weight gradient of a 48-channel 3x3 layer (padded-tap mode of gather_wgrad_bf16_sub_kernel) against the f32 form:
error per tap and per 16-channel block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posterior_matching_amd import ops
from posterior_matching_amd.ops import LayerGeom

d = torch.device("cuda:0")
torch.manual_seed(0)
for (H, ci, co, k, B) in ((28, 48, 48, 3, 3), (28, 48, 48, 3, 4), (14, 48, 48, 3, 5), (7, 48, 192, 1, 3), (28, 64, 64, 3, 3)):
    g = LayerGeom.conv(H, H, ci, co, k, 1, "SAME")
    x = torch.randn((B, H, H, ci), device=d)
    dy = torch.randn((B, H, H, co), device=d)
    dw0, db0 = torch.zeros(g.weight_shape, device=d), torch.zeros(co, device=d)
    dw1, db1 = torch.zeros(g.weight_shape, device=d), torch.zeros(co, device=d)
    ops.layer_wgrad(g, x, dy, dw0, db0, bf16=False)
    ops.layer_wgrad(g, x, dy, dw1, db1, bf16=True)
    torch.cuda.synchronize()
    ref = dw0.abs().max().item()
    print(f"H{H} {ci}->{co} k{k} B{B}: max rel err {((dw1 - dw0).abs().max() / ref).item():.3e}  bias {((db1 - db0).abs().max() / db0.abs().max()).item():.3e}")
    e = (dw1 - dw0).abs() / ref
    for ky in range(k):
        print("   ", " | ".join(" ".join(f"{e[ky, kx, 16 * cb:16 * cb + 16, :].max().item():.1e}" for cb in range(ci // 16)) for kx in range(k)))
