"""Shared test helpers: fixture loading and the reference's per-chunk pytest=True randoms."""
import os

import numpy as np
import torch

from oracle import nerf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def chunked_pytest_randoms(n_rays, chunk, Nc, Nf, perturb, noise_std):
    """Under pytest=True every render_rays chunk re-seeds numpy (run_nerf.py:663-666), so the
    full-batch randoms are the per-chunk seed-0 draws concatenated."""
    parts = {"t_rand": [], "u": [], "noise_c": [], "noise_f": []}
    for s in range(0, n_rays, chunk):
        n = min(chunk, n_rays - s)
        r = O.pytest_randoms(n, Nc, Nf, perturb, noise_std)
        for k in parts:
            parts[k].append(r[k])
    return {k: (torch.cat(v, 0) if v[0] is not None else None) for k, v in parts.items()}


def _bf16_bits_to_f32(a):
    return torch.from_numpy((np.asarray(a).astype(np.uint32) << 16).view(np.float32).copy())


def render_case_nets(g):
    vd, och, Nf = bool(g["vd"]), int(g["och"]), int(g["Nf"])
    if any(k.startswith("wc_") for k in g):
        # networks the REFERENCE trained (tests/golden/make_golden_trained.py): stored as bf16 bit patterns — the reference
        # rendered the fixture with exactly these (bf16-representable) weights
        return ({k[3:]: _bf16_bits_to_f32(v) for k, v in g.items() if k.startswith("wc_")},
                {k[3:]: _bf16_bits_to_f32(v) for k, v in g.items() if k.startswith("wf_")})
    sd_c = O.make_wild_params(seed=11, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0)
    sd_f = O.make_wild_params(seed=12, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0) if Nf > 0 else None
    return sd_c, sd_f


RENDER_CASES = ["render_ndc_fine_vd", "render_lindisp_fine_vd", "render_lindisp_fine_vd_detach",
                "render_ndc_coarse_vd", "render_noperturb_fine_vd_alpha", "render_ndc_fine_novd",
                "render_c2w_fine_vd", "render_trained_fine_vd", "render_trained_black_vd"]
# networks the REFERENCE trained (tests/golden/make_golden_trained.py, seeded and regenerable): the sphere in front of a white
# background with density noise (every ray ends opaque: acc == 1), and in front of a black one without noise (rays that miss
# or graze the sphere: acc in [0, 1), 28 of 48 rays below 0.99)
TRAINED_CASES = ["render_trained_fine_vd", "render_trained_black_vd"]


def fixture_loss(g, mse, rgb, rgb0, disp):
    """the loss the fixture's gradients belong to: img2mse(rgb) [+ img2mse(rgb0)] + 0.1 img2mse(disp, 0) — without the
    disparity term where the fixture says so (a ray that hits nothing has disp = 1 / (0 / 0) = NaN in the reference)"""
    loss = mse(rgb)
    if rgb0 is not None:
        loss = loss + mse(rgb0)
    if int(g.get("disp_loss", 1)):
        loss = loss + 0.1 * (disp ** 2).mean()
    return loss
R2O_CASES = ["r2o_s64", "r2o_s192_white_noise", "r2o_s192_detach", "r2o_s64_zero_sigma",
             "r2o_s64_huge_sigma", "r2o_s5"]
PDF_CASES = ["pdf_rand", "pdf_det", "pdf_delta", "pdf_delta_det", "pdf_uniform", "pdf_zeros", "pdf_small"]
MLP_CASES = ["mlp_default_vd", "mlp_default_novd", "mlp_wild_vd", "mlp_wild_novd"]


def mlp_case_params(g):
    vd = bool(g["use_viewdirs"])
    kw = dict(use_viewdirs=vd, output_ch=4 if vd else 5, input_ch_views=27 if vd else 0)
    return O.make_wild_params(seed=1, **kw) if int(g["wild"]) else O.init_nerf_params(seed=0, **kw)


# ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), numpy restatement of the
# generator the production kernels draw from (render_ops.hip: rng_uniform / rng_normal) ------------------------------
def philox4x32_10(ctr, key):
    """ctr uint32 [..., 4], key (k0, k1) -> uint32 [..., 4]"""
    c = np.array(ctr, dtype=np.uint64)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[..., 0], M1 * c[..., 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = np.stack([hi1 ^ c[..., 1] ^ k0, lo1, hi0 ^ c[..., 3] ^ k1, lo0], -1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c.astype(np.uint32)


def philox_uniform(n, seed, offset):
    """element i of a call: counter (i, 0, offset_lo, offset_hi), key (seed_lo, seed_hi); U[0,1) from 24 bits of word 0"""
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i & np.uint64(0xFFFFFFFF), i >> np.uint64(32), np.full(n, offset & 0xFFFFFFFF, np.uint64),
                    np.full(n, offset >> 32, np.uint64)], -1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    return ((w[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)), w
