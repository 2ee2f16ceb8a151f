"""Diagnostic: oracle train-step throughput vs torch thread count on this host."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O
print("cpu_count", os.cpu_count())
H, W, focal = 378, 504, 400.0
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    n = 512
    sd_c, sd_f = O.init_nerf_params(seed=0), O.init_nerf_params(seed=1)
    params = [p.requires_grad_(True) for sd in (sd_c, sd_f) for p in sd.values()]
    opt = O.AdamState(params, lr=5e-4)
    ro, rd = O.get_rays(H, W, focal, torch.eye(4)[:3, :4])
    sel = torch.randperm(H * W)[:n]
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    kw = dict(H=H, W=W, focal=focal, chunk=32768, ndc=False, near=1.2, far=9.0, use_viewdirs=True, N_samples=64,
              N_importance=128, perturb=1.0, white_bkgd=True, lindisp=True)
    ts = []
    for i in range(3):
        rnd = dict(t_rand=torch.rand(n, 64), u=torch.rand(n, 128), noise_c=torch.randn(n, 64), noise_f=torch.randn(n, 192))
        t0 = time.perf_counter(); O.train_step(sd_c, sd_f, opt, rays, torch.rand(n, 3), kw, randoms=rnd); ts.append(time.perf_counter() - t0)
    print(nt, "threads:", [round(t, 2) for t in ts], "rays/s", round(n / min(ts[1:]), 1), flush=True)
