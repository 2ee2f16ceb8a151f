"""Diagnostic (CPU, oracle only): does the reference's recipe for the hash-grid network (ReLU density through raw2outputs,
torch Adam eps 1e-8, bound 100) learn the analytic sphere when every operation is the oracle's fp32 torch restatement?
Answers whether the 'uniform fog' outcome of tests/probes/hashgrid_train.py belongs to the recipe or to the HIP kernels.
usage: python tests/probes/hashgrid_oracle_train.py [lr] [noise] [iters] [n_rand] [torch init seed]"""
import math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O, hashgrid_oracle as H

HH, WW, FOCAL, NEAR, FAR = 96, 128, 230.0, 2.0, 6.0
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 600
n_rand = int(sys.argv[4]) if len(sys.argv) > 4 else 512
init_seed = int(sys.argv[5]) if len(sys.argv) > 5 else None     # torch seed of create_nerf_tcnn's initialisation


def sphere_scene(rays_o, rays_d):
    d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    b = (rays_o * d).sum(-1)
    c = (rays_o * rays_o).sum(-1) - 1.0
    disc = b * b - c
    t = -b - torch.sqrt(disc.clamp(min=0))
    col = 0.5 + 0.5 * (rays_o + d * t[..., None])
    return torch.where((disc > 0)[..., None], col, torch.zeros_like(col))


rays_all, tgt_all = [], []
for k in range(6):
    a = 2 * math.pi * k / 6
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm()
    x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
    y = torch.linalg.cross(z, x)
    c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1)
    ro, rd = O.get_rays(HH, WW, FOCAL, c2w)
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
    tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3)))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)

torch.manual_seed(0)
sds = []
if init_seed is not None:                      # the initial parameters tests/probes/hashgrid_seed_sweep.py starts from
    import spin_nerf_amd as S
    torch.manual_seed(init_seed)
    inits = [S.NeRF_TCNN().state_dict(), S.NeRF_TCNN().state_dict()]
    torch.manual_seed(0)
else:
    inits = [H.init_params(1), H.init_params(2)]
for sd in inits:
    sds.append({k: (v.clone().requires_grad_(True) if v.numel() else v) for k, v in sd.items()})
params = [v for sd in sds for v in sd.values() if v.requires_grad]
opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
mlp = lambda sd, x, **_: H.nerf_tcnn_forward(sd, x)
g = torch.Generator().manual_seed(1)
ps, t0 = [], time.time()
for it in range(iters):
    sel = torch.randint(0, rays_all.shape[1], (n_rand,), generator=g)
    rnd = dict(t_rand=torch.rand(n_rand, 64), u=torch.rand(n_rand, 64),
               noise_c=torch.randn(n_rand, 64) * noise if noise else None,
               noise_f=torch.randn(n_rand, 128) * noise if noise else None)
    r = O.render(HH, WW, FOCAL, rays=rays_all[:, sel], sd_coarse=sds[0], sd_fine=sds[1], randoms=rnd, N_samples=64,
                 N_importance=64, perturb=1.0, white_bkgd=False, lindisp=False, use_viewdirs=True, ndc=False, near=NEAR,
                 far=FAR, i_embed=-1, mlp=mlp)
    mse = O.img2mse(r[0], tgt_all[sel])
    loss = mse + O.img2mse(r[4]["rgb0"], tgt_all[sel])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    for gp in opt.param_groups:
        gp["lr"] = lr * 0.1 ** ((it + 1) / 250000)
    ps.append(float(-10 * torch.log10(mse)))
    if it % 50 == 49:
        print(f"it {it + 1}: psnr {np.mean(ps[-50:]):.2f} acc {float(r[2].mean()):.3f} ({(time.time() - t0) / (it + 1):.2f} s/it) "
              f"grid absmax {float(sds[1]['encoder.params'].abs().max()):.3g}", flush=True)
