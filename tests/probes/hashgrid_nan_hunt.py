"""Diagnostic: find the first non-finite tensor in the training step at which noise-free hash-grid training (torch seed 4,
lr 1e-2) produced a NaN gradient (tests/probes/hashgrid_collapse.py: iteration 153)."""
import math, os, sys, importlib, contextlib, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
ops = S.ops
dev = torch.device("cuda")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4



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
    kw, kw_test, *_ = S.create_nerf_tcnn(_args(lrate=1e-2, raw_noise_std=0.0), device=dev)
kw.update(near=NEAR, far=FAR)
tr = RenderTrainer(kw, lrate=1e-2, lrate_decay=250)
g = torch.Generator().manual_seed(1)
found = False
for it in range(700):
    sel = torch.randint(0, rays_all.shape[1], (512,), generator=g).to(dev)
    saved = [n.flat.detach().clone() for n in tr.nets]
    saved_draws = tr._draws
    batch_rays, target = rays_all[:, sel].contiguous(), tgt_all[sel]
    tr.step(HH, WW, FOCAL, batch_rays, target)
    if not all(bool(torch.isfinite(n.flat.grad).all()) for n in tr.nets):
        print(f"non-finite gradient at iteration {it + 1}; replaying that step from the saved state", flush=True)
        for n, s_ in zip(tr.nets, saved):
            with torch.no_grad():
                n.flat.copy_(s_)
            n.mark_weights_changed()
        found = True
        break
if not found:
    print("no non-finite gradient in 700 iterations")
    sys.exit(0)

def rep(name, t):
    t = t.detach().float()
    fin = torch.isfinite(t)
    print(f"{name:28s} finite {bool(fin.all())}  non-finite {int((~fin).sum())}  max|finite| {float(t[fin].abs().max()) if fin.any() else float('nan'):.4g}", flush=True)
    return bool(fin.all())


Nc, Nf = kw['N_samples'], kw['N_importance']
net_c, net_f = kw['network_fn'], kw['network_fine']
rays = ops.pack_rays(batch_rays[0], batch_rays[1], HH, WW, FOCAL, ndc=False, near=NEAR, far=FAR, use_viewdirs=True)
vd = rays[:, -3:]
loss = torch.zeros(2, device=dev)
seed_, d = tr._seed, saved_draws
z_c = ops.sample_coarse_rng(rays, Nc, False, seed_, d + 1)
raw_c, sv_c = ops.mlp_train_forward(net_c, rays, z_c, vd)
out_c = ops.composite_train(raw_c, z_c, rays, target, loss[0:1], None, noise=None, noise_std=0.0, seed=seed_, offset=d + 2, white_bkgd=False)
z_f = ops.sample_fine_rng(z_c, out_c[4], Nf, seed_, d + 3)
raw_f, sv_f = ops.mlp_train_forward(net_f, rays, z_f, vd)
out_f = ops.composite_train(raw_f, z_f, rays, target, loss[0:1], loss[1:2], noise=None, noise_std=0.0, seed=seed_, offset=d + 4, white_bkgd=False)
for name, t in (("z_c", z_c), ("raw_c", raw_c), ("weights_c", out_c[4]), ("d_raw_c", out_c[5]), ("z_f", z_f), ("raw_f", raw_f),
                ("rgb_f", out_f[0]), ("weights_f", out_f[4]), ("d_raw_f", out_f[5]), ("loss", loss)):
    rep(name, t)
dr = out_f[5]
bad = ~torch.isfinite(dr).all(-1)
if bool(bad.any()):
    r, s = torch.nonzero(bad)[0].tolist()
    print("first bad d_raw at ray", r, "sample", s)
    print(" raw_f[ray, s-2:s+3]   ", raw_f[r, max(0, s - 2):s + 3].tolist())
    print(" z_f[ray, s-2:s+3]     ", z_f[r, max(0, s - 2):s + 3].tolist())
    print(" weights_f[ray, s-2:s+3]", out_f[4][r, max(0, s - 2):s + 3].tolist())
    print(" d_raw_f[ray, s-2:s+3] ", dr[r, max(0, s - 2):s + 3].tolist())
g_f = ops.mlp_train_backward(net_f, sv_f, out_f[5])
views = net_f.named_views(g_f)
for k, v in views.items():
    if v.numel():
        rep("grad " + k, v)
print("d_raw_f |sigma| max", float(dr[..., 3].abs().max()), " rgb max", float(dr[..., :3].abs().max()))
