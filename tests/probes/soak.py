"""Soak: long training runs on the analytic sphere with finiteness checks (every 500 iterations) — the kind of run that
would have exposed the compositing-backward NaN of round 2 at once.  usage: soak.py [mlp|hash] [iters] [noise]"""
import os, sys, math, importlib, contextlib, io, argparse, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
kind = sys.argv[1] if len(sys.argv) > 1 else "mlp"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device("cuda")


def camera(a):
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm()
    x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
    return torch.cat([torch.stack([x, torch.linalg.cross(z, x), z], 1), eye[:, None]], 1).to(dev)


rays_all, tgt_all = [], []
for k in range(12):
    ro, rd = S.get_rays(HH, WW, FOCAL, camera(2 * math.pi * k / 12))
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
    tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), False))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    if kind == "hash":
        kw, kwt, *_ = S.create_nerf_tcnn(_args(lrate=1e-2, raw_noise_std=noise), device=dev)
        lr = 1e-2
    else:
        a = _args(lrate=5e-4, raw_noise_std=noise, precision="bf16")
        kw, kwt, *_ = S.create_nerf(a, device=dev)
        lr = 5e-4
kw.update(near=NEAR, far=FAR); kwt.update(near=NEAR, far=FAR)
tr = RenderTrainer(kw, lrate=lr, lrate_decay=250)
g = torch.Generator().manual_seed(1)
ps = []
for it in range(iters):
    sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
    loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
    if it % 50 == 0:
        ps.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))
    if it % 500 == 499:
        fin = all(bool(torch.isfinite(n.flat).all()) for n in tr.nets) and math.isfinite(float(loss))
        print(f"{kind} noise {noise} it {it + 1}: psnr {np.mean(ps[-10:]):.2f} finite {fin} lr {tr.current_lr():.2e}", flush=True)
        if not fin:
            sys.exit(1)
held = []
for a in (math.pi / 12, 3 * math.pi / 12):
    with torch.no_grad():
        rgb, disp, acc, depth, ex = S.render(HH, WW, FOCAL, chunk=32768, c2w=camera(a), **kwt)
    ro, rd = S.get_rays(HH, WW, FOCAL, camera(a))
    held.append(float(-10.0 * torch.log10(torch.mean((rgb - sphere_scene(ro, rd, False)) ** 2))))
print("held-out views (between training cameras):", [round(h, 2) for h in held])
