"""Diagnostic: a noise-free fp32 run (seed 2) renders its held-out view black (4 dB) while training at 28 dB — why?"""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
import spin_nerf_amd as S
keep = {}
orig_render = S.render
def spy(*a, **k):
    out = orig_render(*a, **k)
    if k.get("c2w") is not None:
        keep["kw"] = k
    return out
S.render = spy
p, t = T.train("fp32", 1200, seed=2, noise=0.0)
print("train last400", np.mean(p[-400:]), "held-out", t)
k = dict(keep["kw"]); c2w = k.pop("c2w")
H, W, F = T.H, T.W, T.FOCAL
with torch.no_grad():
    rgb, disp, acc, depth, ex = orig_render(H, W, F, c2w=c2w, retraw=True, **{kk: v for kk, v in k.items() if kk != "chunk"}, chunk=32768)
print("acc mean", float(acc.mean()), "rgb max", float(rgb.max()), "raw finite", bool(torch.isfinite(ex["raw"]).all()),
      "sigma max", float(ex["raw"][..., 3].max()), "rgb0 max", float(ex["rgb0"].max()), "acc0 mean", float(ex["acc0"].mean()))
for a in (0.0, 2 * math.pi / 6, 2 * math.pi * 0.5 / 6, 2 * math.pi * 0.25 / 6):
    eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
    z = eye / eye.norm(); x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm(); y = torch.linalg.cross(z, x)
    c = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).cuda()
    with torch.no_grad():
        rgb, disp, acc, depth, ex = orig_render(H, W, F, c2w=c, **{kk: v for kk, v in k.items()})
    ro, rd = S.get_rays(H, W, F, c)
    ps = float(-10.0 * torch.log10(torch.mean((rgb - T.sphere_scene(ro, rd, False)) ** 2)))
    print(f"angle {a:.3f}: psnr {ps:.2f} acc mean {float(acc.mean()):.3f} acc0 mean {float(ex['acc0'].mean()):.3f}")
