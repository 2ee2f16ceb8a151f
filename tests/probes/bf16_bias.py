"""Diagnostic: is the bf16 gradient biased against the fp32 gradient of the SAME weights and batch?  Train fp32 for a while
on the analytic sphere, then evaluate K batches in both precisions: per-tensor relative error of one batch, and of the mean
over K batches (an unbiased error averages out ~1/sqrt(K); a systematic one stays)."""
import os, sys, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
import test_gpu_train as T
iters = int(os.environ.get("ITERS", 1500))
K = int(os.environ.get("K", 24))

# train in fp32 (the function builds its own trainer; re-create the pieces here to keep the nets)
import argparse, tempfile, contextlib, io, math
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=64, N_samples=64,
    alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4,
    basedir=tempfile.mkdtemp(), expname="", ft_path=None, no_reload=True, perturb=1.0, white_bkgd=False, raw_noise_std=1.0,
    dataset_type="llff", no_ndc=True, lindisp=False, sigma_loss=False, no_coarse=False, precision="fp32")
with contextlib.redirect_stdout(io.StringIO()):
    kw, kwt, *_ = S.create_nerf(args, device=dev)
kw.update(near=T.NEAR, far=T.FAR)
tr = RenderTrainer(kw, lrate=5e-4)
rays_all, tgt_all = [], []
for k in range(6):
    a = 2 * math.pi * k / 6
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm(); x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm(); y = torch.linalg.cross(z, x)
    c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).to(dev)
    ro, rd = S.get_rays(T.H, T.W, T.FOCAL, c2w)
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)); tgt_all.append(T.sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), False))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)
g = torch.Generator().manual_seed(123)
for it in range(iters):
    sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
    loss, rgb = tr.step(T.H, T.W, T.FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
print("trained", iters, "its, psnr", float(-10 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))
nets = tr.nets
snap = [n.flat.detach().clone() for n in nets]
Nc, Nf = 64, 64
acc = {p: [torch.zeros_like(snap[0]).double(), torch.zeros_like(snap[1]).double()] for p in ("fp32", "bf16")}
single = None
for b in range(K):
    sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
    rnd = dict(t_rand=torch.rand(1024, Nc, device=dev), u=torch.rand(1024, Nf, device=dev),
               noise_c=torch.randn(1024, Nc, device=dev), noise_f=torch.randn(1024, Nc + Nf, device=dev))
    grads = {}
    for prec in ("fp32", "bf16"):
        for n, s in zip(nets, snap):
            n.set_precision(prec)
            with torch.no_grad():
                n.flat.copy_(s)
            n.mark_weights_changed()
        tr.m = [torch.zeros_like(s) for s in snap]; tr.v = [torch.zeros_like(s) for s in snap]
        tr.step(T.H, T.W, T.FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel], randoms=rnd)
        grads[prec] = [n.flat.grad.double().clone() for n in nets]
        for i in range(2):
            acc[prec][i] += grads[prec][i]
    if b == 0:
        single = grads
for i, name in enumerate(("coarse", "fine")):
    views32 = nets[i].named_views(single["fp32"][i]); views16 = nets[i].named_views(single["bf16"][i])
    m32 = nets[i].named_views(acc["fp32"][i] / K); m16 = nets[i].named_views(acc["bf16"][i] / K)
    for k in views32:
        if not k.endswith("weight"):
            continue
        a, b = views16[k].reshape(-1), views32[k].reshape(-1)
        ma, mb = m16[k].reshape(-1), m32[k].reshape(-1)
        print(f"{name:6s} {k:24s} one batch: rel {float((a - b).norm() / b.norm()):.3f} norm ratio {float(a.norm() / b.norm()):.3f} | "
              f"mean of {K}: rel {float((ma - mb).norm() / mb.norm()):.3f} norm ratio {float(ma.norm() / mb.norm()):.3f}", flush=True)
