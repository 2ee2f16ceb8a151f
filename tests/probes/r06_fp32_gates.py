"""Round 6, VERDICT r05 "next round" item 4: what the fp32 render() path MEASURES on every reference fixture, so that the gates
of tests/test_gpu_render.py and tests/test_path.py can be per-fixture measured values x a stated margin instead of blanket
2e-4 / `mostly_close` fractions.  Prints one JSON object per (fixture, pytest-hook) — the arithmetic is deterministic
(profiles/r06_determinism.txt), so these are the numbers every box computes.

    python tests/probes/r06_fp32_gates.py > gpurun_out/r06_fp32_gates.jsonl
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S                     # noqa: E402
from helpers import load, RENDER_CASES        # noqa: E402
import test_gpu_render as R                   # noqa: E402


def npy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def frac_within(a, b, atol, rtol):
    a, b = npy(a), npy(b)
    nan = np.isnan(b)
    d = np.abs(a - b)[~nan]
    ok = d <= atol + rtol * np.abs(b[~nan])
    return dict(frac=float(ok.mean()), n=int(ok.size), n_out=int((~ok).sum()), max=float(d.max(initial=0.0)))


def maxabs(a, b):
    a, b = npy(a).reshape(-1), npy(b).reshape(-1)
    ok = np.isfinite(b)
    return float(np.abs(a[ok] - b[ok]).max(initial=0.0))


def maxrel(a, b, floor=1e-12):
    a, b = npy(a).reshape(-1), npy(b).reshape(-1)
    ok = np.isfinite(b)
    return float((np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), floor)).max(initial=0.0))


for name in RENDER_CASES:
    g = load(name)
    for hook in (True, False):
        net_c, net_f, kw = R.build(S, g)
        with torch.no_grad():
            rgb, disp, acc, depth, ex = R.run(S, g, kw, hook)
        n = g["rgb"].reshape(-1, 3).shape[0]
        rec = dict(fixture=name, hook=hook, n_rays=n, Nf=int(g["Nf"]))
        if int(g["Nf"]) > 0:
            rec["coarse"] = dict(rgb0=maxabs(ex["rgb0"], g["x_rgb0"]), acc0=maxabs(ex["acc0"], g["x_acc0"]),
                                 disp0_abs=maxabs(ex["disp0"], g["x_disp0"]), disp0_rel=maxrel(ex["disp0"], g["x_disp0"]),
                                 rgb0_scale=float(np.abs(g["x_rgb0"]).max()))
            rec["fine"] = dict(
                z_vals=frac_within(ex["z_vals"], g["x_z_vals"], 1e-4, 1e-5), weights=frac_within(ex["weights"], g["x_weights"], 2e-4, 0),
                rgb=frac_within(rgb, g["rgb"], 2e-4, 0), acc=frac_within(acc, g["acc"], 2e-4, 0),
                depth=frac_within(depth, g["depth"], 2e-4, 1e-3), disp=frac_within(disp, g["disp"], 2e-4, 1e-3),
                z_std=frac_within(ex["z_std"], g["x_z_std"], 2e-4, 1e-3))
            # the same at the SURVEY 8(d) gates (rgb / acc 1e-5, depth / disp rtol 1e-4, weights / z 1e-4)
            rec["fine_survey"] = dict(
                z_vals=frac_within(ex["z_vals"], g["x_z_vals"], 1e-4, 0), weights=frac_within(ex["weights"], g["x_weights"], 1e-4, 0),
                rgb=frac_within(rgb, g["rgb"], 1e-5, 0), acc=frac_within(acc, g["acc"], 1e-5, 0),
                depth=frac_within(depth, g["depth"], 1e-5, 1e-4), disp=frac_within(disp, g["disp"], 1e-5, 1e-4))
            zr = g["x_z_vals"].reshape(n, -1)
            dz = np.abs(npy(ex["z_vals"]).reshape(n, -1) - zr).max(-1)
            tight = dz <= 2e-6 * np.maximum(1.0, np.abs(zr).max(-1))
            rec["rays_with_identical_z"] = dict(frac=float(tight.mean()), n=int(tight.sum()))
            # the maps of the rays whose samples are the reference's: what the fine stage itself is good to
            t = tight
            if t.any():
                rec["tight_rays"] = dict(rgb=maxabs(npy(rgb).reshape(n, 3)[t], g["rgb"].reshape(n, 3)[t]),
                                         acc=maxabs(npy(acc).reshape(n)[t], g["acc"].reshape(n)[t]),
                                         depth=maxabs(npy(depth).reshape(n)[t], g["depth"].reshape(n)[t]),
                                         z_std=maxabs(npy(ex["z_std"]).reshape(n)[t], g["x_z_std"].reshape(n)[t]))
        else:
            rec["coarse_only"] = dict(rgb=maxabs(rgb, g["rgb"]), acc=maxabs(acc, g["acc"]), depth_rel=maxrel(depth, g["depth"], 1e-3),
                                      disp_rel=maxrel(disp, g["disp"], 1e-3), z_vals=maxabs(ex["z_vals"], g["x_z_vals"]),
                                      weights=maxabs(ex["weights"], g["x_weights"]), raw=maxabs(ex["raw"], g["x_raw"]))
        print(json.dumps(rec), flush=True)

# tests/test_path.py:85 — frame 0 of render_path against the c2w fixture
g = load("render_c2w_fine_vd")
_, _, kw = R.build(S, g)
H, W, f, chunk = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["chunk"])
from helpers import T   # noqa: E402
c2w = T(g["c2w"]).cuda()
rgbs, disps, _ = S.render_path(torch.stack([c2w, c2w], 0), (H, W, f), chunk, kw)
ref = g["rgb"].reshape(H, W, 3)
d = np.abs(rgbs[0] - ref)
print(json.dumps(dict(fixture="render_path frame 0 vs render_c2w_fine_vd", n=int(d.size), over_1e4=int((d > 1e-4).sum()),
                      over_1e5=int((d > 1e-5).sum()), max=float(d.max()))), flush=True)
