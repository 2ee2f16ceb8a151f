"""Diagnostic: the hash-grid network alone regressing a smooth target (no rendering): does Adam on its gradients learn?"""
import os, sys, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
torch.manual_seed(0)
for lr in (1e-2, 2e-3):
    net = S.NeRF_TCNN().cuda()
    m, v = torch.zeros_like(net.flat.data), torch.zeros_like(net.flat.data)
    g = torch.Generator(device="cuda").manual_seed(1)
    hist = []
    for it in range(300):
        pts = (torch.rand(256, 16, 3, device="cuda", generator=g) * 4 - 2)
        dirs = torch.nn.functional.normalize(torch.randn(256, 3, device="cuda", generator=g), dim=-1)
        tgt = torch.cat([torch.sin(pts * 2.0), (pts.norm(dim=-1, keepdim=True) < 1.0).float() * 3.0], -1)
        net.flat.grad = None
        out = net.query(pts, dirs)
        loss = ((out - tgt) ** 2).mean()
        loss.backward()
        S.adam_step_(net.flat.data, net.flat.grad, m, v, lr, it + 1)
        net.mark_weights_changed()
        hist.append(float(loss))
    print("lr", lr, "loss", [round(float(np.mean(hist[i:i + 10])), 4) for i in range(0, 300, 30)], flush=True)
