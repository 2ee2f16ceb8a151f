"""Diagnostic (GPU): the HIP hash-grid path on the analytic sphere with exactly the set-up of hashgrid_oracle_train.py
(same initial parameters, ray selection, sample counts, optimizer = torch Adam on the flat buffers), for a side-by-side
PSNR / accumulated-opacity trajectory.  usage: python tests/probes/hashgrid_train_gpu.py [lr] [noise] [iters] [n_rand] [mode]
mode: torch (autograd + torch.optim.Adam) | trainer (RenderTrainer.step)"""
import math, os, sys, time, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
from oracle import hashgrid_oracle as H        # initial parameters only

HH, WW, FOCAL, NEAR, FAR = 96, 128, 230.0, 2.0, 6.0
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 800
n_rand = int(sys.argv[4]) if len(sys.argv) > 4 else 512
mode = sys.argv[5] if len(sys.argv) > 5 else "torch"
dev = torch.device("cuda")


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
    c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).to(dev)
    ro, rd = S.get_rays(HH, WW, FOCAL, c2w)
    rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
    tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3)))
rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)

nets = []
for seed in (1, 2):
    net = S.NeRF_TCNN().to(dev)
    net.load_state_dict(H.init_params(seed))
    nets.append(net)


def q(inputs, viewdirs, network_fn):
    return S.run_network(inputs, viewdirs, network_fn)
q._snr_fused = True
kw = dict(network_query_fn=q, perturb=1.0, N_importance=64, network_fine=nets[1], N_samples=64, network_fn=nets[0],
          use_viewdirs=True, white_bkgd=False, raw_noise_std=noise, ndc=False, lindisp=False, near=NEAR, far=FAR)
g = torch.Generator().manual_seed(1)
torch.manual_seed(0)
ps, accs, t0 = [], [], time.time()
if mode == "trainer":
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    tr = RenderTrainer(kw, lrate=lr, lrate_decay=250)
else:
    opt = torch.optim.Adam([n.flat for n in nets], lr=lr, betas=(0.9, 0.999))
for it in range(iters):
    sel = torch.randint(0, rays_all.shape[1], (n_rand,), generator=g).to(dev)
    if mode == "trainer":
        loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
        acc = torch.zeros(1)
    else:
        rgb, disp, acc, depth, ex = S.render(HH, WW, FOCAL, rays=rays_all[:, sel].contiguous(), **kw)
        loss = S.img2mse(rgb, tgt_all[sel]) + S.img2mse(ex["rgb0"], tgt_all[sel])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        for gp in opt.param_groups:
            gp["lr"] = lr * 0.1 ** ((it + 1) / 250000)
    ps.append(float(-10 * torch.log10(torch.mean((rgb.detach() - tgt_all[sel]) ** 2))))
    accs.append(float(acc.mean()))
    if it % 50 == 49:
        v = nets[1].named_views(nets[1].flat.detach())
        print(f"it {it + 1}: psnr {np.mean(ps[-50:]):.2f} acc {np.mean(accs[-50:]):.3f} ({(time.time() - t0) / (it + 1) * 1e3:.1f} ms/it) "
              f"grid absmax {float(v['encoder.params'].abs().max()):.3g}", flush=True)
