"""GPU: end-to-end training on an analytic scene — the bf16 path must reach the PSNR of the fp32
parity path at equal iterations (BASELINE.json: "matched PSNR (+-0.1 dB)"; the gate here allows for
run-to-run chaos of two different arithmetic paths), and both must actually learn the scene."""
import argparse
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, FOCAL, NEAR, FAR = 96, 128, 230.0, 2.0, 6.0


def sphere_scene(rays_o, rays_d, white=True):
    """analytic target: a unit sphere at the origin shaded by its normal, white or black background"""
    d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    b = (rays_o * d).sum(-1)
    c = (rays_o * rays_o).sum(-1) - 1.0
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp(min=0))
    n = rays_o + d * t[..., None]
    col = 0.5 + 0.5 * n
    return torch.where(hit[..., None], col, torch.ones_like(col) if white else torch.zeros_like(col))


def train(precision, iters, n_rand=1024, seed=0, white=False, noise=1.0, n_fine=64):
    S = importlib.import_module("spin-nerf_amd")
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    import tempfile
    args = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=n_fine, N_samples=64,
        alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536,
        lrate=5e-4, basedir=tempfile.mkdtemp(), expname="", ft_path=None, no_reload=True, perturb=1.0,
        white_bkgd=white, raw_noise_std=noise, dataset_type="llff", no_ndc=True, lindisp=False, sigma_loss=False,
        no_coarse=False, precision=precision)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, *_ = S.create_nerf(args, device=dev)
    kw_train.update(near=NEAR, far=FAR)
    kw_test.update(near=NEAR, far=FAR)
    tr = RenderTrainer(kw_train, lrate=5e-4, lrate_decay=250)
    # 6 cameras on a ring looking at the origin
    rays_all, tgt_all = [], []
    for k in range(6):
        a = 2 * math.pi * k / 6
        eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
        z = eye / eye.norm()
        x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
        y = torch.linalg.cross(z, x)
        c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).to(dev)
        ro, rd = S.get_rays(H, W, FOCAL, c2w)
        rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
        tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), white))
    rays_all = torch.cat(rays_all, 1)
    tgt_all = torch.cat(tgt_all, 0)
    g = torch.Generator(device="cpu").manual_seed(123 + seed)   # (its own ray batches per seed)
    psnrs = []
    for it in range(iters):
        sel = torch.randint(0, rays_all.shape[1], (n_rand,), generator=g).to(dev)
        loss, rgb = tr.step(H, W, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
        mse = torch.mean((rgb - tgt_all[sel]) ** 2)
        psnrs.append(float(-10.0 * torch.log10(mse)))
    # deterministic full-frame renders (perturb = 0, no density noise): a training camera and a held-out view
    def view_psnr(a):
        eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
        z = eye / eye.norm(); x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm(); y = torch.linalg.cross(z, x)
        c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1).to(dev)
        with torch.no_grad():
            rgb, *_ = S.render(H, W, FOCAL, chunk=32768, c2w=c2w, **kw_test)
        ro, rd = S.get_rays(H, W, FOCAL, c2w)
        return float(-10.0 * torch.log10(torch.mean((rgb - sphere_scene(ro, rd, white)) ** 2)))
    train.seen_view_psnr = view_psnr(0.0)
    return psnrs, view_psnr(2 * math.pi * 0.5 / 6)


@pytest.mark.timeout(1500)
def test_bf16_training_matches_fp32_psnr():
    """Black background + raw_noise_std=1 (the reference's config value).  The noise is a regulariser, not a
    necessity: noise-free runs reach 28 dB on the training views in both precisions (tests/probes/mlp_noise_free.py) but
    may fill unobserved directions with view-dependent fog.  (Until round 2 noise-free runs died into an all-empty
    state: a 0 * NaN in the compositing backward of rays that hit nothing — fixed, test_gpu_kernels.py:
    test_composite_backward_of_a_ray_that_hits_nothing_is_finite.)

    The comparison is PAIRED by seed (same initial weights, ray batches and in-kernel draws for both precisions) and made
    on the mean of the last 400 of 1200 steps.  Measured on MI355X over 12 seeds (round 3): paired difference bf16 - fp32
    -0.00 dB, standard deviation of a pair 0.17 dB (largest 0.34).  The pre-registered experiment of round 4 (a scene in
    which every ray ends on a surface, 16 training + 4 held-out cameras, 64 + 128 samples, 2500 steps; statistic, seeds and
    "no exclusions" fixed before the run: tests/probes/psnr_r04.py, profiles/r04_psnr_heldout.txt) gave, over ALL pairs,
    held-out -0.053 +- 0.069 dB (32 pairs; the planned 16: +0.107 +- 0.116) and last-500-step training PSNR -0.024 +- 0.026 dB,
    no basin flip in either precision.  Gates here: the mean of EIGHT pairs within 0.15 dB (2.5 standard errors at the measured
    0.17 dB per pair), every pair within 0.8 dB, both paths above 24 dB — BASELINE.json asks for +-0.1 dB on the statue scene,
    which is not in the container: that claim stays untested."""
    iters, seeds = 1200, (0, 1, 2, 3, 4, 5, 6, 7)
    diffs = []
    for seed in seeds:
        p32, t32 = train("fp32", iters, seed=seed)
        s32 = train.seen_view_psnr
        p16, t16 = train("bf16", iters, seed=seed)
        s16 = train.seen_view_psnr
        tail32, tail16 = float(np.mean(p32[-400:])), float(np.mean(p16[-400:]))
        print(f"seed {seed}: train PSNR (last 400 of {iters}): fp32 {tail32:.2f} dB, bf16 {tail16:.2f} dB; deterministic full-frame "
              f"render of a training camera: fp32 {s32:.2f} dB, bf16 {s16:.2f} dB; start {np.mean(p32[:5]):.2f} dB")
        assert s32 > 22.0 and s16 > 22.0, "the inference path does not reproduce what was trained"
        assert tail32 > 24.0, "fp32 path did not learn the scene"
        assert tail16 > 24.0, "bf16 path did not learn the scene"
        assert abs(tail16 - tail32) < 0.8, (seed, tail16, tail32)
        diffs.append(tail16 - tail32)
    print(f"paired differences bf16 - fp32: {[round(d, 3) for d in diffs]}, mean {np.mean(diffs):+.3f} dB")
    assert abs(float(np.mean(diffs))) < 0.15, diffs
