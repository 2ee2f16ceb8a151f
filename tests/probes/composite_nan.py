"""Diagnostic: NaN in d raw[..., 3] of the compositing backward for a tiny positive density behind an opaque sample."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
from oracle import nerf_oracle as O
ops = S.ops
dev = torch.device("cuda")
for S_ in (8, 64, 128):
    for tiny in (3.7e-8, 3e-6, 1e-3, 1e-12, 1e-30, 1e-38, 1e-40):
        for opaque in (50.0, 5.0, 0.0):
            raw = torch.full((1, S_, 4), -0.05)
            raw[..., :3] = torch.linspace(-1, 1, S_)[None, :, None]
            raw[0, 2, 3] = opaque
            raw[0, S_ // 2 + 1, 3] = tiny
            z = torch.linspace(2.0, 6.0, S_)[None]
            rays = torch.zeros(1, 11); rays[0, 5] = -1.0; rays[0, 10] = -1.0
            tgt = torch.tensor([[0.2, 0.4, 0.6]])
            loss = torch.zeros(2, device=dev)
            out = ops.composite_train(raw.to(dev), z.to(dev), rays.to(dev), tgt.to(dev), loss[0:1], None, noise=None, noise_std=0.0,
                                      seed=1, offset=1, white_bkgd=False)
            d = out[5][0, :, 3].cpu()
            rr = raw.clone().requires_grad_(True)
            o = O.raw2outputs(rr, z, rays[:, 3:6])
            l = O.img2mse(o[0], tgt)
            l.backward()
            ref = rr.grad[0, :, 3]
            bad = ~torch.isfinite(d)
            print(f"S {S_:3d} tiny {tiny:8.1e} opaque {opaque:4.1f}: kernel non-finite at {torch.nonzero(bad).flatten().tolist()}, "
                  f"oracle non-finite at {torch.nonzero(~torch.isfinite(ref)).flatten().tolist()}, max|diff| {float((d - ref)[~bad].abs().max()):.3g}")
