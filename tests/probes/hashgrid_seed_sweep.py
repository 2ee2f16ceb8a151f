"""Diagnostic: outcome of hash-grid training on the analytic sphere as a function of the initialisation seed and of
raw_noise_std (ReLU density can start dead: sigma <= 0 everywhere gives no gradient at all)."""
import math, os, sys, importlib, contextlib, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")


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
import sys
NOISES = (0.0, 1.0) if len(sys.argv) < 2 else tuple(float(x) for x in sys.argv[1].split(','))
LRS = (1e-2, 5e-4) if len(sys.argv) < 3 else tuple(float(x) for x in sys.argv[2].split(','))
for noise in NOISES:
    for lr in LRS:
        for seed in range(6):
            torch.manual_seed(seed)
            with contextlib.redirect_stdout(io.StringIO()):
                kw_train, kw_test, *_ = S.create_nerf_tcnn(_args(lrate=lr, raw_noise_std=noise), device=dev)
            kw_train.update(near=NEAR, far=FAR); kw_test.update(near=NEAR, far=FAR)
            tr = RenderTrainer(kw_train, lrate=lr, lrate_decay=250)
            g = torch.Generator().manual_seed(1)
            ps = []
            for it in range(600):
                sel = torch.randint(0, rays_all.shape[1], (512,), generator=g).to(dev)
                loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
                ps.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))
            with torch.no_grad():
                rgb, disp, acc, depth, ex = S.render(HH, WW, FOCAL, chunk=32768, c2w=camera(0.0), **kw_test)
            ro, rd = S.get_rays(HH, WW, FOCAL, camera(0.0))
            seen = float(-10.0 * torch.log10(torch.mean((rgb - sphere_scene(ro, rd, False)) ** 2)))
            print(f"noise {noise} lr {lr} seed {seed}: start {np.mean(ps[:5]):.2f} it100 {np.mean(ps[90:110]):.2f} "
                  f"it300 {np.mean(ps[290:310]):.2f} last50 {np.mean(ps[-50:]):.2f} dB; training view {seen:.2f} dB, opacity {float(acc.mean()):.3f}", flush=True)
