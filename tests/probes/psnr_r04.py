"""bf16 vs fp32 training quality, PRE-REGISTERED.  Everything below was fixed before the first run:

SCENE (analytic, chosen so that no run can end in an "all-empty" or "fog" basin: EVERY ray ends on an opaque surface and
16 cameras see every surface from several sides): five shaded, textured spheres inside a large textured enclosing sphere
(radius 5) seen from the inside; 16 training cameras on two rings (radius 3, heights +0.8 / -0.6) looking at the origin,
4 held-out cameras between them (radius 3.1, height 0.1); 96 x 128 pixels, focal 110, near 0.4, far 9; black "background"
never occurs.  64 coarse + 128 fine samples, viewdirs, raw_noise_std = 1, perturb = 1, 1024 rays per step, lr 5e-4 with the
reference's decay (run_nerf.py:1616-1622), ITERS = 2500 steps.

PAIRS: SEEDS = 16 initialisation seeds; for each, one fp32 run (exact-fp32 MFMA mode: the reference's arithmetic) and one
bf16 run from the SAME initial weights, ray batches and in-kernel random draws.

STATISTIC (primary): paired difference bf16 - fp32 of the HELD-OUT PSNR (mean over the 4 held-out views of a deterministic
full-frame render), over ALL pairs — no exclusions of any kind —, reported as mean +- standard error.  Secondary: the same
for the mean training-batch PSNR of the last 500 steps.  A pair whose two runs differ by more than 3 dB held-out is a
"basin flip": counted and reported, and still part of the mean.
BASELINE.json asks for +-0.1 dB on the statue scene (not in the container): the claim this experiment can support is
|mean| <= 0.1 dB at a standard error <= 0.1 dB on THIS scene; anything else is reported as measured.

EXTENSION (the one decision taken after seeing data, recorded here before its run): the first 16 pairs gave a standard error
of 0.116 dB on the primary statistic, above the 0.1 dB the plan asked for; SEED0=16 runs 16 more seeds (16..31) with nothing
else changed, and the pooled 32-pair statistic is reported BESIDE the 16-pair one (profiles/r04_psnr_heldout.txt).

ROUND 5 BLOCK (decided by the round-4 review, recorded here before its run): SEED0=32 SEEDS=32 — seeds 32..63, same script,
same statistic, no exclusions; reported per 16-pair block and pooled over all 64 pairs (profiles/r05_psnr_heldout.txt).  If the
pooled held-out difference is below -0.1 dB at 2 standard errors, README and DESIGN section 2 say so.

ROUND 6 BLOCK (recorded here before its run): round 6 changes the bf16 ARITHMETIC — the encodings and the weight columns they meet
are fp16 (csrc/mlp_layout.h: EncF16).  The same 64 seeds (0..63) are re-run in bf16 with nothing else changed: REUSE_FP32=1
SEED0=0 SEEDS=64.  The fp32 halves of the pairs are NOT re-run: the exact-fp32 path is unchanged and bit-reproducible
(profiles/r06_determinism.txt), so each pair takes its fp32 numbers from the per-seed lines of profiles/r04_psnr_heldout.txt /
r05_psnr_heldout.txt — and to hold that assumption to account, CHECK_FP32 (default seeds 0, 16, 32, 48) ARE re-run in fp32 and must
reproduce their recorded lines to the printed precision, or the script stops.  Statistic as before: paired held-out difference
bf16 - fp32 over ALL 64 pairs, no exclusions; beside it the paired difference new bf16 - old bf16.  Reported whatever it is
(profiles/r06_psnr_heldout.txt).
"""
import argparse
import contextlib
import importlib
import io
import math
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

H, W, FOCAL, NEAR, FAR = 96, 128, 110.0, 0.4, 9.0
ITERS = int(os.environ.get("ITERS", "2500"))
SEEDS = int(os.environ.get("SEEDS", "16"))
SEED0 = int(os.environ.get("SEED0", "0"))

SPHERES = [  # centre, radius, base colour
    ((0.0, 0.0, 0.0), 0.9, (0.9, 0.3, 0.2)),
    ((1.3, 0.4, 0.2), 0.5, (0.2, 0.8, 0.3)),
    ((-1.1, -0.3, 0.9), 0.6, (0.2, 0.4, 0.9)),
    ((0.2, 0.9, -1.2), 0.45, (0.9, 0.8, 0.2)),
    ((-0.6, -0.8, -1.0), 0.55, (0.7, 0.3, 0.8)),
]
R_WALL = 5.0


def scene(rays_o, rays_d):
    """nearest hit among the five spheres, else the enclosing sphere from the inside; Lambert-like shading + a smooth texture"""
    d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    o = rays_o
    t_best = torch.full(o.shape[:-1], float("inf"), device=o.device)
    col = torch.zeros_like(o)
    light = torch.tensor([0.5, 0.8, 0.3], device=o.device)
    light = light / light.norm()
    for c, r, base in SPHERES:
        c = torch.tensor(c, device=o.device)
        oc = o - c
        b = (oc * d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        t = -b - torch.sqrt(disc.clamp(min=0))
        hit = (disc > 0) & (t > 0) & (t < t_best)
        p = o + d * t[..., None]
        n = (p - c) / r
        shade = 0.55 + 0.45 * (n * light).sum(-1, keepdim=True).clamp(min=-1, max=1)
        tex = 0.85 + 0.15 * torch.sin(6.0 * p[..., :1]) * torch.sin(6.0 * p[..., 1:2] + 1.0)
        cc = torch.tensor(base, device=o.device) * shade * tex
        col = torch.where(hit[..., None], cc, col)
        t_best = torch.where(hit, t, t_best)
    # the wall: |o + t d| = R, the far root
    b = (o * d).sum(-1)
    disc = b * b - ((o * o).sum(-1) - R_WALL * R_WALL)
    t = -b + torch.sqrt(disc.clamp(min=0))
    p = (o + d * t[..., None]) / R_WALL
    wall = 0.5 + 0.3 * torch.stack([torch.sin(3.0 * p[..., 0] + 0.5), torch.sin(4.0 * p[..., 1] + 1.5),
                                    torch.sin(3.5 * p[..., 2] + 2.5)], -1) * (0.8 + 0.2 * torch.sin(5.0 * p[..., 1:2] * p[..., :1] * 3.0))
    return torch.where(torch.isinf(t_best)[..., None], wall.clamp(0, 1), col.clamp(0, 1))


def look_at(eye):
    z = eye / eye.norm()
    x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    return torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1)


def cameras():
    tr, ho = [], []
    for k in range(8):
        a = 2 * math.pi * k / 8
        tr.append(look_at(torch.tensor([3 * math.sin(a), 0.8, 3 * math.cos(a)])))
        a2 = a + math.pi / 8
        tr.append(look_at(torch.tensor([3 * math.sin(a2), -0.6, 3 * math.cos(a2)])))
    for k in range(4):
        a = 2 * math.pi * (k + 0.5) / 4 + 0.2
        ho.append(look_at(torch.tensor([3.1 * math.sin(a), 0.1, 3.1 * math.cos(a)])))
    return tr, ho


def run(precision, seed):
    S = importlib.import_module("spin-nerf_amd")
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    args = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=128, N_samples=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, basedir=tempfile.mkdtemp(),
        expname="", ft_path=None, no_reload=True, perturb=1.0, white_bkgd=False, raw_noise_std=1.0, dataset_type="llff",
        no_ndc=True, lindisp=False, sigma_loss=False, no_coarse=False, precision=precision)
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, *_ = S.create_nerf(args, device=dev)
    kw_train.update(near=NEAR, far=FAR)
    kw_test.update(near=NEAR, far=FAR)
    tr = RenderTrainer(kw_train, lrate=5e-4, lrate_decay=250)
    cams_tr, cams_ho = cameras()
    rays_all, tgt_all = [], []
    for c2w in cams_tr:
        ro, rd = S.get_rays(H, W, FOCAL, c2w.to(dev))
        rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
        tgt_all.append(scene(ro.reshape(-1, 3), rd.reshape(-1, 3)))
    rays_all = torch.cat(rays_all, 1)
    tgt_all = torch.cat(tgt_all, 0)
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)   # its own ray batches per seed, shared by the pair
    ps = []
    for it in range(ITERS):
        sel = torch.randint(0, rays_all.shape[1], (1024,), generator=g).to(dev)
        loss, rgb = tr.step(H, W, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
        if it >= ITERS - 500:
            ps.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))

    def view_psnr(c2w):
        c2w = c2w.to(dev)
        with torch.no_grad():
            rgb, *_ = S.render(H, W, FOCAL, chunk=32768, c2w=c2w, **kw_test)
        ro, rd = S.get_rays(H, W, FOCAL, c2w)
        return float(-10.0 * torch.log10(torch.mean((rgb - scene(ro, rd)) ** 2)))
    return float(np.mean(ps)), float(np.mean([view_psnr(c) for c in cams_ho])), view_psnr(cams_tr[0])


def recorded():
    """per-seed numbers of the earlier blocks: {seed: {"fp32": (train, heldout, cam), "bf16": (...)}}"""
    import re
    out = {}
    pat = re.compile(r"seed\s+(\d+): last-500 train fp32 ([\d.]+) bf16 ([\d.]+) \| held-out \(4 views\) fp32 ([\d.]+) bf16 ([\d.]+) \| "
                     r"training camera fp32 ([\d.]+) bf16 ([\d.]+)")
    for f in ("r04_psnr_heldout.txt", "r05_psnr_heldout.txt"):
        for l in open(os.path.join(ROOT, "profiles", f)):
            m = pat.match(l.strip())
            if m:
                v = [float(x) for x in m.groups()[1:]]
                out[int(m.group(1))] = {"fp32": (v[0], v[2], v[4]), "bf16": (v[1], v[3], v[5])}
    return out


def main():
    rows = []
    reuse = os.environ.get("REUSE_FP32") == "1"
    rec = recorded() if reuse else {}
    if reuse:
        for seed in [int(x) for x in os.environ.get("CHECK_FP32", "0,16,32,48").split(",") if x != ""]:
            got = run("fp32", seed)
            want = rec[seed]["fp32"]
            ok = all(abs(a - b) < 0.0051 for a, b in zip(got, want))
            print(f"fp32 check, seed {seed}: re-run {got[0]:.2f} {got[1]:.2f} {got[2]:.2f}  recorded {want[0]:.2f} {want[1]:.2f} {want[2]:.2f}  "
                  f"{'reproduces' if ok else 'DIFFERS'}", flush=True)
            if not ok:
                raise SystemExit("the fp32 path no longer reproduces its recorded PSNR: re-run the pairs in full (REUSE_FP32=0)")
    old_bf16 = []
    for seed in range(SEED0, SEED0 + SEEDS):
        if reuse:
            r = {"fp32": rec[seed]["fp32"], "bf16": run("bf16", seed)}
            old_bf16.append(rec[seed]["bf16"])
        else:
            r = {p: run(p, seed) for p in ("fp32", "bf16")}
        rows.append(r)
        print(f"seed {seed:2d}: last-500 train fp32 {r['fp32'][0]:.2f} bf16 {r['bf16'][0]:.2f} | held-out (4 views) fp32 {r['fp32'][1]:.2f} "
              f"bf16 {r['bf16'][1]:.2f} | training camera fp32 {r['fp32'][2]:.2f} bf16 {r['bf16'][2]:.2f}", flush=True)
    for i, name in ((1, "held-out views (PRIMARY)"), (0, "last-500 training batches"), (2, "deterministic render of a training camera")):
        d = np.array([r["bf16"][i] - r["fp32"][i] for r in rows])
        print(f"{name}: fp32 {np.mean([r['fp32'][i] for r in rows]):.2f} dB, bf16 {np.mean([r['bf16'][i] for r in rows]):.2f} dB, "
              f"paired difference over ALL {len(d)} pairs {d.mean():+.3f} +- {d.std(ddof=1) / np.sqrt(len(d)):.3f} dB "
              f"(sd of a pair {d.std(ddof=1):.2f}, largest |difference| {np.abs(d).max():.2f})")
    if old_bf16:
        for i, name in ((1, "held-out views"), (0, "last-500 training batches")):
            d = np.array([r["bf16"][i] - o[i] for r, o in zip(rows, old_bf16)])
            print(f"NEW bf16 arithmetic against the recorded bf16 runs, {name}: paired difference over {len(d)} seeds {d.mean():+.3f} +- "
                  f"{d.std(ddof=1) / np.sqrt(len(d)):.3f} dB (sd {d.std(ddof=1):.2f})")
    flips = [i for i, r in enumerate(rows) if abs(r["bf16"][1] - r["fp32"][1]) > 3.0]
    print(f"basin flips (held-out difference > 3 dB; included in the means above): {flips}")


if __name__ == "__main__":
    main()
