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


def render_case_nets(g):
    vd, och, Nf = bool(g["vd"]), int(g["och"]), int(g["Nf"])
    sd_c = O.make_wild_params(seed=11, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0)
    sd_f = O.make_wild_params(seed=12, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0) if Nf > 0 else None
    return sd_c, sd_f


RENDER_CASES = ["render_ndc_fine_vd", "render_lindisp_fine_vd", "render_lindisp_fine_vd_detach",
                "render_ndc_coarse_vd", "render_noperturb_fine_vd_alpha", "render_ndc_fine_novd",
                "render_c2w_fine_vd"]
R2O_CASES = ["r2o_s64", "r2o_s192_white_noise", "r2o_s192_detach", "r2o_s64_zero_sigma",
             "r2o_s64_huge_sigma", "r2o_s5"]
PDF_CASES = ["pdf_rand", "pdf_det", "pdf_delta", "pdf_delta_det", "pdf_uniform", "pdf_zeros", "pdf_small"]
MLP_CASES = ["mlp_default_vd", "mlp_default_novd", "mlp_wild_vd", "mlp_wild_novd"]


def mlp_case_params(g):
    vd = bool(g["use_viewdirs"])
    kw = dict(use_viewdirs=vd, output_ch=4 if vd else 5, input_ch_views=27 if vd else 0)
    return O.make_wild_params(seed=1, **kw) if int(g["wild"]) else O.init_nerf_params(seed=0, **kw)
