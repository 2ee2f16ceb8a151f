"""Diagnostic (not a test): the numbers the gates of tests/test_gpu_render.py are calibrated from.

For every render fixture and both precisions:
  * coarse stage (no resampling upstream): max |err| of rgb0 / acc0 and max rel err of disp0 (or of the final maps when
    the case has no fine stage);
  * fine stage teacher-forced on the fixture's z_vals: max |err| of raw (absolute and relative to max |raw|), then the maps
    composited from the kernel's OWN raw;
  * free-running: per ray, the largest z displacement `dz` and the map errors; fraction of rays with dz <= 2e-6 (bit-level
    agreement) and the worst map error among those; the worst (error - tight gate) / dz ratio among the others.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from helpers import load, T, RENDER_CASES, chunked_pytest_randoms
from test_gpu_render import build, run, pack_rays


def mx(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-12)).max())


for name in RENDER_CASES:
    g = load(name)
    fine = int(g["Nf"]) > 0
    for prec in ("fp32", "bf16"):
        net_c, net_f, kw = build(S, g, prec)
        with torch.no_grad():
            rgb, disp, acc, depth, ex = run(S, g, kw, True)
        out = {"case": name, "prec": prec}
        n = g["rgb"].reshape(-1, 3).shape[0]
        c = lambda t: t.detach().cpu().numpy()
        if fine:
            out["coarse"] = dict(rgb0=mx(c(ex["rgb0"]), g["x_rgb0"]), acc0=mx(c(ex["acc0"]), g["x_acc0"]),
                                 disp0_rel=rel(c(ex["disp0"]), g["x_disp0"]))
            # teacher-forced fine stage
            rays = pack_rays(S, g)
            z = T(g["x_z_vals"]).reshape(n, -1).cuda()
            vd = bool(g["vd"])
            with torch.no_grad():
                raw = net_f.query_rays(rays, z, rays[:, -3:] if vd else None)
            ref_raw = g["x_raw"].reshape(n, z.shape[1], -1)
            noise = None
            if float(g["noise_std"]) > 0:
                rnd = chunked_pytest_randoms(n, int(g["chunk"]), 64, int(g["Nf"]), float(g["perturb"]), float(g["noise_std"]))
                noise = rnd["noise_f"].cuda()
            with torch.no_grad():
                r2, d2, a2, w2, dp2, _ = S.raw2outputs(raw, z, rays[:, 3:6], white_bkgd=bool(g["white"]), noise=noise,
                                                       rays=rays)
            out["teacher"] = dict(raw=mx(c(raw), ref_raw), raw_scale=float(np.abs(ref_raw).max()),
                                  rgb=mx(c(r2), g["rgb"].reshape(n, 3)), acc=mx(c(a2), g["acc"].reshape(n)),
                                  weights=mx(c(w2), g["x_weights"].reshape(n, -1)),
                                  depth_rel=rel(c(dp2), g["depth"].reshape(n)), depth=mx(c(dp2), g["depth"].reshape(n)),
                                  disp_rel=rel(c(d2), g["disp"].reshape(n)))
            # free-running attribution
            zz, zr = c(ex["z_vals"]).reshape(n, -1), g["x_z_vals"].reshape(n, -1)
            dz = np.abs(zz - zr).max(-1)
            e_rgb = np.abs(c(rgb).reshape(n, 3) - g["rgb"].reshape(n, 3)).max(-1)
            e_acc = np.abs(c(acc).reshape(n) - g["acc"].reshape(n))
            e_depth = np.abs(c(depth).reshape(n) - g["depth"].reshape(n))
            rd = g["disp"].reshape(n)
            e_disp = np.abs(c(disp).reshape(n) - rd) / np.maximum(np.abs(rd), 1e-12)
            e_zstd = np.abs(c(ex["z_std"]).reshape(n) - g["x_z_std"].reshape(n))
            tight = dz <= 2e-6 * np.maximum(1.0, np.abs(zr).max(-1))
            free = dict(frac_rays_bit_equal_z=float(tight.mean()), dz_max=float(dz.max()))
            for key, e in (("rgb", e_rgb), ("acc", e_acc), ("depth", e_depth), ("disp_rel", e_disp), ("z_std", e_zstd)):
                free[key + "_tight_max"] = float(e[tight].max()) if tight.any() else None
                free[key + "_max"] = float(e.max())
                if (~tight).any():
                    free[key + "_per_dz"] = float((e[~tight] / dz[~tight]).max())
            out["free"] = free
        else:
            out["coarse"] = dict(rgb=mx(c(rgb), g["rgb"]), acc=mx(c(acc), g["acc"]), depth_rel=rel(c(depth), g["depth"]),
                                 disp_rel=rel(c(disp), g["disp"]), weights=mx(c(ex["weights"]), g["x_weights"]),
                                 raw=mx(c(ex["raw"]), g["x_raw"]), raw_scale=float(np.abs(g["x_raw"]).max()))
        print(json.dumps(out), flush=True)
