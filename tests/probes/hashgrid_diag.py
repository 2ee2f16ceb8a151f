"""Diagnostic: per-matrix gradient agreement of the hash-grid path with the oracle, forward error statistics."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from oracle import hashgrid_oracle as H
from test_gpu_hashgrid import make, samples

sd, net = make(S, 4)
pts, dirs = samples(5, 41, 16)
rs = np.random.RandomState(6)
d_raw = torch.from_numpy(rs.normal(size=(41, 16, 4)).astype(np.float32))
with torch.no_grad():
    out = net.query(pts.cuda(), dirs.cuda()).cpu()
for emu in (True, False):
    ref = H.run_network(sd, pts, dirs, bf16emu=emu)
    e = (out - ref).abs()
    print("forward vs", "emu" if emu else "fp32", "max", float(e.max()), "rms", float(e.pow(2).mean().sqrt()), "scale", float(ref.abs().max()),
          "per channel max", e.reshape(-1, 4).max(0)[0].tolist(), flush=True)
splits = {"sigma_net.params": [("W1s", 0, 2048), ("W2s", 2048, 3072)],
          "color_net.params": [("W1c", 0, 2048), ("W2c", 2048, 6144), ("W3c", 6144, 7168)]}
for which, emu in (("bf16emu", True), ("fp32", False)):
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.numel()}
    full = dict(sd); full.update(p)
    (H.run_network(full, pts, dirs, bf16emu=emu) * d_raw).sum().backward()
    net.flat.grad = None
    o = net.query(pts.cuda(), dirs.cuda())
    (o * d_raw.cuda()).sum().backward()
    got = net.named_views(net.flat.grad)
    print(which, "kernel grads finite:", {k: bool(torch.isfinite(v).all()) for k, v in got.items()}, flush=True)
    for k, parts in splits.items():
        for name, lo, hi in parts:
            a, b = got[k][lo:hi].cpu().double(), p[k].grad[lo:hi].double()
            print(which, name, "rel", float((a - b).norm() / b.norm()), "cos", float((a @ b) / (a.norm() * b.norm())),
                  "norms", float(a.norm()), float(b.norm()), flush=True)
    w3 = got["color_net.params"][6144:7168].cpu().reshape(16, 64)
    r3 = p["color_net.params"].grad[6144:7168].reshape(16, 64)
    print(which, "W3c rows rel", [round(float((w3[i] - r3[i]).norm() / r3[i].norm().clamp_min(1e-20)), 4) for i in range(4)], flush=True)
