"""Diagnostic: is the sudden 'all-empty' collapse of noise-free hash-grid training a NaN / Inf event or a dead-ReLU state?
Trains create_nerf_tcnn networks (torch seed argv[1], default 1) at lr 1e-2 without density noise and reports, every 25
iterations, the loss, whether parameters / gradients are finite, and the largest gradient and sigma-network output."""
import math, os, sys, importlib, contextlib, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def camera(a):
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm()
    x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
    return torch.cat([torch.stack([x, torch.linalg.cross(z, x), z], 1), eye[:, None]], 1).to(dev)


rays_all, tgt_all = [], []
for k in range(6):
    ro, rd = S.get_rays(HH, WW, FOCAL, camera(2 * math.pi * k / 6))
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
    tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), False))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)
torch.manual_seed(seed)
with contextlib.redirect_stdout(io.StringIO()):
    kw_train, kw_test, *_ = S.create_nerf_tcnn(_args(lrate=1e-2, raw_noise_std=0.0), device=dev)
kw_train.update(near=NEAR, far=FAR)
tr = RenderTrainer(kw_train, lrate=1e-2, lrate_decay=250)
g = torch.Generator().manual_seed(1)
for it in range(700):
    sel = torch.randint(0, rays_all.shape[1], (512,), generator=g).to(dev)
    loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
    bad = [not bool(torch.isfinite(n.flat).all()) for n in tr.nets]
    gbad = [not bool(torch.isfinite(n.flat.grad).all()) for n in tr.nets]
    if it % 25 == 24 or any(bad) or any(gbad) or not math.isfinite(float(loss)):
        gm = [float(n.flat.grad.abs().max()) for n in tr.nets]
        print(f"it {it + 1}: loss {float(loss):.5f} psnr {float(-10 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))):.2f} "
              f"params non-finite {bad} grads non-finite {gbad} max|grad| {gm}", flush=True)
    if any(bad):
        break
