"""Diagnostic: per-kernel time of a hash-grid training step (bench-shaped: 1024 rays x (64 + 128) samples)."""
import os, sys, importlib, contextlib, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_hashgrid import _args
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    kw, kwt, *_ = S.create_nerf_tcnn(_args(N_importance=128, lrate=5e-4, raw_noise_std=1.0, white_bkgd=True, lindisp=True), device=dev)
kw.update(near=1.2, far=9.0)
tr = RenderTrainer(kw, lrate=5e-4)
H, W, f = 378, 504, 400.0
ro, rd = S.get_rays(H, W, f, torch.eye(4)[:3, :4].to(dev))
ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
g = torch.Generator().manual_seed(5)
batches = []
for _ in range(4):
    sel = torch.randperm(H * W, generator=g)[:1024].to(dev)
    batches.append((torch.stack([ro[sel], rd[sel]], 0).contiguous(), torch.rand(1024, 3, generator=g).to(dev)))
for i in range(3):
    tr.step(H, W, f, *batches[i % 4])
S._lib.prof_enable(True); S._lib.prof_read()
import time
torch.cuda.synchronize(); t0 = time.time()
n = 10
for i in range(n):
    tr.step(H, W, f, *batches[i % 4])
torch.cuda.synchronize(); dt = (time.time() - t0) / n * 1e3
p = S._lib.prof_read()
print("ms/step", round(dt, 3), {k: round(v[0] / n, 4) for k, v in p.items()})
