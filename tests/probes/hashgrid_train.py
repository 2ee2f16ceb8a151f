"""Diagnostic: training trajectories of the hash-grid path on the analytic sphere for a few settings."""
import os, sys, math, contextlib, io, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")
rays_all, tgt_all = [], []
for k in range(6):
    a = 2 * math.pi * k / 6
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm()
    x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
    y = torch.linalg.cross(z, x)
    c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).to(dev)
    ro, rd = S.get_rays(HH, WW, FOCAL, c2w)
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
    tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), False))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)
for lr, noise, iters in ((1e-2, 0.0, 2000), (3e-3, 0.0, 2000), (1e-2, 1.0, 2000), (3e-2, 0.0, 2000)):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, *_ = S.create_nerf_tcnn(_args(lrate=lr, raw_noise_std=noise), device=dev)
    kw_train.update(near=NEAR, far=FAR)
    tr = RenderTrainer(kw_train, lrate=lr, lrate_decay=250)
    g = torch.Generator().manual_seed(1)
    ps = []
    for it in range(iters):
        sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
        loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
        ps.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))
    import time
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(20):
        sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
        tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
    torch.cuda.synchronize(); print("ms/step", (time.time() - t0) / 20 * 1e3)
    n = tr.nets[1]
    v = n.named_views(n.flat.detach())
    print(f"lr {lr} noise {noise}: psnr", [round(float(np.mean(ps[i:i + 10])), 2) for i in range(0, iters, 200)],
          "grid absmax", float(v["encoder.params"].abs().max()), "sigma w absmax", float(v["sigma_net.params"].abs().max()), flush=True)
