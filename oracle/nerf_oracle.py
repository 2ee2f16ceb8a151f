"""CPU oracle for the SPIn-NeRF volumetric-render hot path.

TEST INFRASTRUCTURE ONLY.  This file is a clean-room restatement (plain PyTorch
CPU ops, fp32 or fp64) of the arithmetic that ``DS_NeRF/run_nerf.py``'s
``render()`` / ``render_rays()`` / ``network_query_fn`` surface performs in the
reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package (``spin-nerf_amd``) never
does, and fails loudly when its HIP library is missing.

Parity status: PINNED.  The reference publishes no golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself:
``tests/golden/make_golden.py`` imports the reference read-only in the build
container and dumps input/output fixtures (``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks every function below against them.

Each function cites the reference lines it follows (paths relative to
``/root/reference``).  All random draws are explicit arguments (``t_rand``,
``u``, ``noise``) so a GPU kernel can be fed the identical numbers; the
reference's ``pytest=True`` hook (numpy seed-0 draws) is reproduced by
``pytest_randoms``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# loss lambdas  (DS_NeRF/run_nerf_helpers.py:15-18)
# --------------------------------------------------------------------------

def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / math.log(10.0)


# --------------------------------------------------------------------------
# positional encoding  (DS_NeRF/run_nerf_helpers.py:22-70)
# --------------------------------------------------------------------------

def embed(x: torch.Tensor, multires: int, i_embed: int = 0) -> torch.Tensor:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)].

    Channel order: the raw input first, then per frequency (sin xyz, cos xyz)
    (helpers:31-46).  Frequencies are exact powers of two (helpers:39), no pi.
    ``i_embed == -1`` is the identity (helpers:56-57).
    """
    if i_embed == -1:
        return x
    outs = [x]
    for k in range(multires):
        f = float(2.0 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def embed_dim(multires: int, i_embed: int = 0) -> int:
    return 3 if i_embed == -1 else 3 + 6 * multires


def embed_kernel_arithmetic(x: torch.Tensor, multires: int) -> torch.Tensor:
    """The same encoding with the ARGUMENT REDUCTION of the HIP bf16-mode kernels (spin-nerf_amd/csrc/mlp_device.h: encode /
    encode_static): the angle goes to the hardware sine as a number of revolutions, rev = fl32(x * fl32(2^k / 2 pi)), reduced
    to its fractional part (exact in fp32), then sin / cos of 2 pi frac.  The product's rounding is the only inexact step that
    matters: 2^-24 of |rev| revolutions, 6e-5 rad at x = 2, k = 9 — below the bf16 rounding of the result (2e-3) by far, but
    a quarter of an fp16 ulp (round 6: the encodings are fp16), so the emulation reproduces it; sin / cos themselves are
    evaluated in double (the hardware instruction's own error is not modelled).  Test infrastructure for the bf16-mode
    comparisons only: the reference's encoding is embed() above."""
    x = x.to(torch.float32)
    outs = [x]
    c = torch.tensor(0.15915494309189535, dtype=torch.float32)
    for k in range(multires):
        sc = c * float(2.0 ** k)                       # exact scaling of the fp32 constant
        rev = x * sc                                   # one fp32 rounding
        fr = (rev - torch.floor(rev)).to(torch.float64)
        outs.append(torch.sin(2.0 * math.pi * fr).to(torch.float32))
        outs.append(torch.cos(2.0 * math.pi * fr).to(torch.float32))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------
# NeRF MLP  (DS_NeRF/run_nerf_helpers.py:74-127)
# --------------------------------------------------------------------------

def init_nerf_params(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                     use_viewdirs=True, seed=0, gain=1.0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Parameters under the reference's state-dict key names with nn.Linear's
    default init distribution (U(-1/sqrt(in), 1/sqrt(in)) for weight and bias).  Drawn from
    numpy's frozen legacy MT19937 stream so fixtures only need to record (seed, gain), not the
    1.19 M weights.  ``gain`` > 1 widens the weights so raw outputs are not the near-constant
    ones default init gives (default init is too weak a test)."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, fin, fout, g=gain):
        bound = g / math.sqrt(fin)
        sd[name + ".weight"] = torch.from_numpy(rs.uniform(-bound, bound, size=(fout, fin))).to(dtype)
        sd[name + ".bias"] = torch.from_numpy(rs.uniform(-bound, bound, size=(fout,))).to(dtype)

    lin("pts_linears.0", input_ch, W)
    for i in range(D - 1):
        lin(f"pts_linears.{i + 1}", W + input_ch if i in skips else W, W)
    lin("views_linears.0", input_ch_views + W, W // 2)
    if use_viewdirs:
        lin("feature_linear", W, W)
        lin("alpha_linear", W, 1)
        lin("rgb_linear", W // 2, 3)
    else:
        lin("output_linear", W, output_ch)
    return sd


def make_wild_params(seed=1, use_viewdirs=True, output_ch=4, input_ch=63, input_ch_views=27):
    """A 'trained-like' net for fixtures: gain-2.5 init (raw rgb spread ~±3) with the density
    head boosted (x4 weight, +0.5 bias) so rays see a mix of empty space and opaque samples."""
    sd = init_nerf_params(seed=seed, gain=2.5, use_viewdirs=use_viewdirs, output_ch=output_ch,
                          input_ch=input_ch, input_ch_views=input_ch_views)
    if use_viewdirs:
        sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * 4.0
        sd["alpha_linear.bias"] = sd["alpha_linear.bias"] + 0.5
    else:
        sd["output_linear.weight"] = sd["output_linear.weight"].clone()
        sd["output_linear.weight"][3] *= 4.0
        sd["output_linear.bias"] = sd["output_linear.bias"].clone()
        sd["output_linear.bias"][3] += 0.5
    return sd


def nerf_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, input_ch=63, input_ch_views=27,
                 skips=(4,), use_viewdirs=True) -> torch.Tensor:
    """NeRF.forward (helpers:104-127).  ``x`` = cat(embedded pts, embedded dirs).

    8x(Linear+ReLU); after layer index i in ``skips`` the *input* is concatenated in
    front (helpers:110-111); alpha head has no activation (helpers:114); feature head
    has no ReLU (helpers:115); cat(feature, views) -> Linear(283,128)+ReLU -> rgb
    (helpers:116-122); output = cat(rgb, alpha) (helpers:123).
    """
    input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    D = len([k for k in sd if k.startswith("pts_linears.") and k.endswith(".weight")])
    h = input_pts
    for i in range(D):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([input_pts, h], -1)
    if use_viewdirs:
        alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
        feature = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
        h = torch.cat([feature, input_views], -1)
        h = F.relu(F.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
        rgb = F.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
        return torch.cat([rgb, alpha], -1)
    return F.linear(h, sd["output_linear.weight"], sd["output_linear.bias"])


def nerf_rgb_forward(sd: Dict[str, torch.Tensor], sd_alpha: Dict[str, torch.Tensor], x: torch.Tensor, input_ch=63,
                     input_ch_views=27, skips=(4,)) -> torch.Tensor:
    """NeRF_RGB.forward with view directions (helpers:191-216): the trunk, feature/views/rgb heads of ``sd`` (which
    has no alpha_linear) and the density of the frozen ``sd_alpha`` network evaluated under no_grad (:202-203)."""
    input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    h = input_pts
    for i in range(8):
        h = torch.relu(torch.nn.functional.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([input_pts, h], -1)
    with torch.no_grad():
        alpha = nerf_forward(sd_alpha, x, input_ch, input_ch_views, skips, True)[..., 3][..., None]
    feature = torch.nn.functional.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    h = torch.cat([feature, input_views], -1)
    h = torch.relu(torch.nn.functional.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
    rgb = torch.nn.functional.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    return torch.cat([rgb, alpha], -1)


def _bf16(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 and back (what v_cvt_pk_bf16_f32 does)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _f16(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to fp16 and back (v_cvt_pk_f16_f32)."""
    return t.to(torch.float16).to(torch.float32)


class _EncLinear(torch.autograd.Function):
    """y = enc16 W^T: the forward multiplies the fp16 encoding, the weight gradient is taken against its bf16 re-rounding — what
    the HIP path does (the saved encodings are bf16: the weight-gradient pass pairs them with d z, which needs bf16's range).
    No gradient to the encoding."""
    @staticmethod
    def forward(ctx, enc16, W):
        ctx.save_for_backward(_bf16(enc16))
        return enc16 @ W.t()

    @staticmethod
    def backward(ctx, g):
        (encbf,) = ctx.saved_tensors
        return None, g.t() @ encbf


def nerf_forward_bf16emu(sd, x, input_ch=63, input_ch_views=27, skips=(4,), use_viewdirs=True, enc_f16=True):
    """Same network with the HIP bf16 kernel's rounding points emulated: weights and
    every MFMA *input* activation rounded to bf16, accumulation / bias / ReLU in fp32.
    Used to hold the bf16 kernel to a tight tolerance instead of a loose fp32 one.

    Round 6 (spin-nerf_amd/csrc/mlp_layout.h: EncF16): the positional / directional encodings, and the weight COLUMNS that
    multiply them (all of pts_linears.0, the first input_ch columns of the skip layer, the last input_ch_views columns of
    views_linears.0), are fp16 instead of bf16 — the same MFMA rate with 11 mantissa bits where the inputs live in [-1, 1].
    ``enc_f16=False`` is the round-5 arithmetic (everything bf16)."""
    q = _bf16
    qe = _f16 if enc_f16 else _bf16
    if enc_f16:
        # weights: rounded in the forward, the gradient passes through unrounded (a cast's autograd casts the GRADIENT too, and
        # fp16 has no range for weight gradients of 1e-6; the kernels accumulate them in fp32)
        qw = lambda W: W + (_f16(W) - W).detach()
    else:
        qw = _bf16
    input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    if enc_f16:
        # the kernels re-derive the encodings from the raw coordinates (the first three columns of an embedding,
        # helpers:31-33) with their own argument reduction: reproduce it (embed_kernel_arithmetic)
        if input_ch > 3 and (input_ch - 3) % 6 == 0:
            input_pts = embed_kernel_arithmetic(input_pts[..., :3], (input_ch - 3) // 6)
        if input_ch_views > 3 and (input_ch_views - 3) % 6 == 0:
            input_views = embed_kernel_arithmetic(input_views[..., :3], (input_ch_views - 3) // 6)
    input_pts, input_views = qe(input_pts), qe(input_views)
    D = len([k for k in sd if k.startswith("pts_linears.") and k.endswith(".weight")])
    h = input_pts
    def enc_lin(enc, W):          # the encoding segment of a layer
        return _EncLinear.apply(enc, qw(W)) if enc_f16 else F.linear(enc, q(W))
    for i in range(D):
        W = sd[f"pts_linears.{i}.weight"]
        b = sd[f"pts_linears.{i}.bias"]
        if i == 0:
            z = enc_lin(input_pts, W) + b
        elif (i - 1) in skips:                      # input = cat([input_pts, h]) (helpers:110-111)
            z = enc_lin(input_pts, W[:, :input_ch]) + F.linear(h, q(W[:, input_ch:]), b)
        else:
            z = F.linear(h, q(W), b)
        h = q(F.relu(z))
    if use_viewdirs:
        alpha = F.linear(h, q(sd["alpha_linear.weight"]), sd["alpha_linear.bias"])
        feature = q(F.linear(h, q(sd["feature_linear.weight"]), sd["feature_linear.bias"]))
        Wv = sd["views_linears.0.weight"]
        nv = Wv.shape[1] - input_ch_views           # input = cat([feature, input_views]) (helpers:119)
        z = F.linear(feature, q(Wv[:, :nv]), sd["views_linears.0.bias"])
        if input_ch_views > 0:
            z = z + enc_lin(input_views, Wv[:, nv:])
        h = q(F.relu(z))
        rgb = F.linear(h, q(sd["rgb_linear.weight"]), sd["rgb_linear.bias"])
        return torch.cat([rgb, alpha], -1)
    return F.linear(h, q(sd["output_linear.weight"]), sd["output_linear.bias"])


# --------------------------------------------------------------------------
# run_network == network_query_fn  (DS_NeRF/run_nerf.py:56-71, 427-430)
# --------------------------------------------------------------------------

def run_network(sd, inputs, viewdirs, multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                netchunk=None, mlp=nerf_forward):
    """Flatten, embed, expand viewdirs per sample (run_nerf.py:63), cat, apply the net,
    reshape back.  ``netchunk`` is a memory knob only (run_nerf.py:44-53)."""
    flat = inputs.reshape(-1, inputs.shape[-1])
    emb = embed(flat, multires, i_embed)
    n_views = 0
    if viewdirs is not None:
        d = viewdirs[:, None].expand(inputs.shape).reshape(-1, inputs.shape[-1])
        demb = embed(d, multires_views, i_embed)
        n_views = demb.shape[-1]
        emb = torch.cat([emb, demb], -1)
    out = mlp(sd, emb, input_ch=embed_dim(multires, i_embed), input_ch_views=n_views,
              use_viewdirs=use_viewdirs and viewdirs is not None)
    return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])


# --------------------------------------------------------------------------
# rays  (DS_NeRF/run_nerf_helpers.py:249-260, 283-300)
# --------------------------------------------------------------------------

def get_rays(H, W, focal, c2w):
    """Pinhole rays (helpers:249-260): dirs = [(i-W/2)/f, -(j-H/2)/f, -1]; d = R dirs; o = t."""
    dt = c2w.dtype
    i = torch.arange(W, dtype=dt)[None, :].expand(H, W)
    j = torch.arange(H, dtype=dt)[:, None].expand(H, W)
    dirs = torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NDC warp (helpers:283-300)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# stratified sampling  (DS_NeRF/run_nerf.py:646-668)
# --------------------------------------------------------------------------

def sample_z(near, far, N_samples, lindisp=False, t_rand: Optional[torch.Tensor] = None):
    """z = near(1-t)+far t, or 1/(1/near(1-t)+1/far t) if lindisp (run_nerf.py:646-650);
    with ``t_rand`` (perturb>0): z = lower+(upper-lower)*t_rand over mid-point bins
    (run_nerf.py:654-668).  near/far: [N,1]."""
    t_vals = torch.linspace(0., 1., steps=N_samples, dtype=near.dtype)
    if not lindisp:
        z = near * (1. - t_vals) + far * t_vals
    else:
        z = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
    z = z.expand([near.shape[0], N_samples])
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


# --------------------------------------------------------------------------
# hierarchical sampling  (DS_NeRF/run_nerf_helpers.py:304-347)
# --------------------------------------------------------------------------

def sample_pdf(bins, weights, N_samples, det=False, u: Optional[torch.Tensor] = None):
    """Inverse-CDF sampling.  ``u`` overrides the uniform draws; det -> linspace(0,1,N)."""
    weights = weights + 1e-5                                  # helpers:306
    pdf = weights / torch.sum(weights, -1, keepdim=True)      # helpers:307
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)  # helpers:308-309
    if u is None:
        if det:
            u = torch.linspace(0., 1., steps=N_samples, dtype=bins.dtype)
            u = u.expand(list(cdf.shape[:-1]) + [N_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [N_samples], dtype=bins.dtype)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)             # helpers:331
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b = torch.gather(cdf, -1, below)
    cdf_a = torch.gather(cdf, -1, above)
    bins_b = torch.gather(bins, -1, below)
    bins_a = torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)  # helpers:343
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b)


# --------------------------------------------------------------------------
# alpha compositing  (DS_NeRF/run_nerf_helpers.py:350-401)
# --------------------------------------------------------------------------

def raw2outputs(raw, z_vals, rays_d, noise=None, white_bkgd=False, need_alpha=False,
                detach_weights=False):
    """``noise`` is the already-scaled additive density noise ([N,S]) or None.
    Returns (rgb_map, disp_map, acc_map, weights, depth_map, alpha|None) like the reference."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)   # helpers:366-367
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)                # helpers:369
    rgb = torch.sigmoid(raw[..., :3])
    sigma_in = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1. - torch.exp(-F.relu(sigma_in) * dists)                       # helpers:364,382
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * T                                                     # helpers:384
    w_rgb = weights.detach() if detach_weights else weights                 # helpers:385-388
    rgb_map = torch.sum(w_rgb[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)  # helpers:391
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map, (alpha if need_alpha else None)


# --------------------------------------------------------------------------
# the reference's pytest=True random hook
# --------------------------------------------------------------------------

def raw2outputs_mvseg(raw, z_vals, rays_d, noise=None, white_bkgd=False, only_object=False, threshold=None,
                      harsh_bg_remove=False):
    """MVSeg variant (MVSeg/DS_NeRF/run_nerf_helpers.py:350-413): raw carries a 5th channel of per-sample logits,
    composited with DETACHED weights: prob_map = sum(w.detach() * logit) (:405).  `only_object` (:383-397): alpha is
    multiplied by 1 - sigmoid(logit), with `threshold` zeroed above it and box-smoothed five times along the ray;
    `harsh_bg_remove` (:410-411) subtracts 10 (1 - acc) from prob_map.
    Returns (rgb_map, disp_map, acc_map, weights, depth_map, prob_map, logits)."""
    logits = raw[..., 4]
    if not only_object:
        rgb_map, disp_map, acc_map, weights, depth_map, _ = raw2outputs(raw, z_vals, rays_d, noise, white_bkgd)
    else:
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
        dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
        sig = raw[..., 3] if noise is None else raw[..., 3] + noise
        alpha = (1. - torch.exp(-F.relu(sig) * dists)) * (1 - torch.sigmoid(logits))
        if threshold is not None:
            alpha = torch.where(alpha > threshold, torch.zeros_like(alpha), alpha)     # alpha[alpha > threshold] = 0
            for _ in range(5):
                z0 = torch.zeros((alpha.shape[0], 1), dtype=alpha.dtype)
                alpha = (torch.hstack([z0, alpha[:, :-1]]) + alpha + torch.hstack([alpha[:, 1:], z0])) / 3
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
        weights = alpha * T
        rgb_map = torch.sum(weights[..., None] * torch.sigmoid(raw[..., :3]), -2)
        depth_map = torch.sum(weights * z_vals, -1)
        acc_map = torch.sum(weights, -1)
        disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
        if white_bkgd:
            rgb_map = rgb_map + (1. - acc_map[..., None])
    prob_map = torch.sum(weights.detach() * logits, -1)
    if only_object and harsh_bg_remove:
        prob_map = prob_map - 10 * (1. - acc_map)
    return rgb_map, disp_map, acc_map, weights, depth_map, prob_map, logits


def sigma_loss(sd, rays_o, rays_d, viewdirs, near, depths, N_samples, perturb=0., t_rand=None, noise=None):
    """SigmaLoss.calculate_loss (DS_NeRF/loss.py:15-44): samples between near and the known depth, stratified
    perturbation, sigma = relu(raw[...,3] + noise), loss = -exp(sigma_last) / (sum exp(sigma) + 1) per ray."""
    N_rays = rays_o.shape[0]
    t_vals = torch.linspace(0., 1., steps=N_samples).expand([N_rays, N_samples])
    z_vals = near * (1. - t_vals) + depths[:, None] * t_vals
    if perturb > 0.:
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = run_network(sd, pts, viewdirs)
    sigma = F.relu(raw[..., 3] + (noise if noise is not None else 0.))
    return -torch.exp(sigma[:, -1]) / (torch.sum(torch.exp(sigma), dim=1) + 1)


def pytest_randoms(N_rays, N_samples, N_importance, perturb, raw_noise_std, dtype=torch.float32):
    """The numbers the reference draws under ``pytest=True``: each site re-seeds numpy
    with 0 and draws ``np.random.rand`` (run_nerf.py:663-666, helpers:319-327, 377-380 —
    density noise is *uniform* there).  Every site therefore sees the same stream prefix."""
    out = {"t_rand": None, "u": None, "noise_c": None, "noise_f": None}

    def draw(*shape):
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*shape)).to(dtype)

    if perturb > 0.:
        out["t_rand"] = draw(N_rays, N_samples)
        if N_importance > 0:
            out["u"] = draw(N_rays, N_importance)
    elif N_importance > 0:
        # det path under pytest: float64 np.linspace cast to fp32 (helpers:322-327), which is not
        # bit-identical to the torch.linspace of the non-pytest det path (helpers:313)
        out["u"] = torch.Tensor(np.broadcast_to(np.linspace(0., 1., N_importance),
                                                (N_rays, N_importance)).copy()).to(dtype)
    if raw_noise_std > 0.:
        out["noise_c"] = draw(N_rays, N_samples) * raw_noise_std
        if N_importance > 0:
            out["noise_f"] = draw(N_rays, N_samples + N_importance) * raw_noise_std
    return out


# --------------------------------------------------------------------------
# render_rays / render  (DS_NeRF/run_nerf.py:593-737, 90-165)
# --------------------------------------------------------------------------

def render_rays(ray_batch, sd_coarse, sd_fine, N_samples, N_importance=0, retraw=False, lindisp=False,
                perturb=0., white_bkgd=False, need_alpha=False, detach_weights=False,
                t_rand=None, u=None, noise_c=None, noise_f=None,
                multires=10, multires_views=4, i_embed=0, mlp=nerf_forward):
    """One chunk of rays through the whole pipeline.  Randoms are explicit:
    ``t_rand`` [N,Nc] (needed iff perturb>0), ``u`` [N,Nf] (optional; when None, perturb==0 uses
    the deterministic linspace and perturb>0 draws torch.rand, run_nerf.py:699),
    ``noise_c``/``noise_f`` pre-scaled."""
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 9 else None       # run_nerf.py:642
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z_vals = sample_z(near, far, N_samples, lindisp, t_rand if perturb > 0. else None)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]

    def query(p, sd):
        return run_network(sd, p, viewdirs, multires, multires_views, i_embed,
                           use_viewdirs=viewdirs is not None, mlp=mlp)

    raw = query(pts, sd_coarse)
    rgb_map, disp_map, acc_map, weights, depth_map, alpha = raw2outputs(
        raw, z_vals, rays_d, noise_c, white_bkgd, need_alpha, detach_weights)
    z_samples = None
    if N_importance > 0:
        rgb0, disp0, acc0, alpha0 = rgb_map, disp_map, acc_map, alpha
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.),
                               u=u).detach()                                 # run_nerf.py:697-700
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)       # run_nerf.py:702
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        raw = query(pts, sd_fine if sd_fine is not None else sd_coarse)
        rgb_map, disp_map, acc_map, weights, depth_map, alpha = raw2outputs(
            raw, z_vals, rays_d, noise_f, white_bkgd, need_alpha, detach_weights)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map,
           'weights': weights, 'z_vals': z_vals}
    if retraw:
        ret['raw'] = raw
    if need_alpha:
        ret['alpha'] = alpha
        ret['alpha0'] = alpha0   # NameError when N_importance == 0, as in the reference (run_nerf.py:721)
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'] = rgb0, disp0, acc0
        ret['z_std'] = torch.std(z_samples, dim=-1, unbiased=False)
    return ret


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1.,
           use_viewdirs=False, c2w_staticcam=None, depths=None, patch=None, randoms=None, **kwargs):
    """render() (run_nerf.py:90-165).  ``randoms`` = dict of full-batch t_rand/u/noise_c/noise_f,
    sliced per chunk.  Returns [rgb_map, disp_map, acc_map, depth_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w)
        if patch is not None:
            i, j, l1, l2 = patch
            rays_o, rays_d = rays_o[i:i + l1, j:j + l2, :], rays_d[i:i + l1, j:j + l2, :]
    else:
        rays_o, rays_d = rays
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d                                       # before NDC (run_nerf.py:128-135)
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = viewdirs.reshape(-1, 3)
    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])]
    if depths is not None:
        cols.append(depths.reshape(-1, 1))
    if use_viewdirs:
        cols.append(viewdirs)
    rays_flat = torch.cat(cols, -1)
    randoms = randoms or {}
    parts: Dict[str, list] = {}
    for s in range(0, rays_flat.shape[0], chunk):
        rnd = {k: (v[s:s + chunk] if v is not None else None) for k, v in randoms.items()}
        ret = render_rays(rays_flat[s:s + chunk], **rnd, **kwargs)
        for k, v in ret.items():
            parts.setdefault(k, []).append(v)
    all_ret = {k: torch.cat(v, 0) for k, v in parts.items()}
    for k in all_ret:
        all_ret[k] = all_ret[k].reshape(list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map', 'depth_map']
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


# --------------------------------------------------------------------------
# one training step (the rays/s metric) — CPU baseline body
# --------------------------------------------------------------------------

class AdamState:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) restated (run_nerf.py:433-434)."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


def train_step(sd_c, sd_f, opt: AdamState, rays, target, render_kwargs, randoms=None):
    """One ``render()`` of the batch + mse(rgb)+mse(rgb0) + backward + Adam — the step the
    rays/s metric counts (run_nerf.py:1465-1490, 1611-1612)."""
    for p in opt.params:
        p.grad = None
    rgb, disp, acc, depth, extras = render(rays=rays, sd_coarse=sd_c, sd_fine=sd_f, randoms=randoms,
                                           retraw=True, **render_kwargs)
    loss = img2mse(rgb, target)
    if 'rgb0' in extras:
        loss = loss + img2mse(extras['rgb0'], target)
    loss.backward()
    opt.step()
    return loss.detach(), rgb.detach()
