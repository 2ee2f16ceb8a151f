"""torch.autograd bindings of the HIP kernels (one Function per C-ABI forward/backward pair).

Every function here enqueues work on torch's current HIP stream through ``_lib`` and allocates its
outputs/workspaces with torch's caching allocator; there is no eager/PyTorch fallback.
"""
import torch

from . import _lib
from ._lib import check, f32c, ptr, stream


def _req(t, numel, dev, what):
    """The kernels take raw device pointers: a tensor on another device, or with fewer elements than the kernel will read, is a
    fault or a silent out-of-bounds read there — so it is an error HERE (the reference's torch ops raise for the same inputs).
    ``t`` None passes (optional operands)."""
    if t is None:
        return
    if t.device != dev:
        raise ValueError(f"{what}: expected a tensor on {dev}, got {t.device}")
    if t.numel() != numel:
        raise ValueError(f"{what}: expected {numel} elements, got {tuple(t.shape)}")


# ----------------------------------------------------------------------------------------------
# stratified sampling (run_nerf.py:646-668)
# ----------------------------------------------------------------------------------------------
def sample_coarse(ray_batch, N_samples, lindisp=False, t_rand=None):
    """z_vals [N_rays, N_samples] from near/far = ray_batch[:, 6:8]; ``t_rand`` = perturb draws."""
    lib = _lib.load()
    rays = f32c(ray_batch)
    n = rays.shape[0]
    z = torch.empty(n, N_samples, device=rays.device, dtype=torch.float32)
    tr = f32c(t_rand) if t_rand is not None else None
    _req(tr, n * N_samples, rays.device, "sample_coarse: t_rand")
    check(lib.snr_sample_coarse(ptr(rays), rays.shape[1], n, N_samples, int(bool(lindisp)), ptr(tr), ptr(z),
                                stream()), "snr_sample_coarse")
    return z


# ----------------------------------------------------------------------------------------------
# hierarchical sampling (helpers:304-347 + run_nerf.py:697-702, 726)
# ----------------------------------------------------------------------------------------------
def sample_fine(z_coarse, weights, N_importance, u=None):
    """-> (z_vals sorted union [N, Nc+Nf], z_samples [N, Nf], z_std [N]).  ``u`` None = det linspace.
    Outputs are detached by construction (the reference detaches z_samples, run_nerf.py:700)."""
    lib = _lib.load()
    zc, w = f32c(z_coarse.detach()), f32c(weights.detach())
    n, nc = zc.shape
    z_out = torch.empty(n, nc + N_importance, device=zc.device, dtype=torch.float32)
    z_s = torch.empty(n, N_importance, device=zc.device, dtype=torch.float32)
    z_std = torch.empty(n, device=zc.device, dtype=torch.float32)
    uu = f32c(u) if u is not None else None
    _req(w, n * nc, zc.device, "sample_fine: weights")
    _req(uu, n * N_importance, zc.device, "sample_fine: u")
    check(lib.snr_sample_fine(ptr(zc), ptr(w), ptr(uu), n, nc, N_importance, ptr(z_out), ptr(z_s), ptr(z_std),
                              stream()), "snr_sample_fine")
    return z_out, z_s, z_std


# ----------------------------------------------------------------------------------------------
# alpha compositing (helpers:350-401)
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """The reference's sample_pdf (helpers:304-347) with its signature: bins [N, nb], weights [N, nb-1] ->
    samples [N, N_samples] (no gradient, like the reference's `.detach()`ed use).  ``pytest`` reproduces its numpy
    seed-0 draws; ``u`` injects them."""
    import numpy as np
    lib = _lib.load()
    b, w = f32c(bins.detach()), f32c(weights.detach())
    n = b.shape[0]
    if u is None:
        if pytest:
            np.random.seed(0)
            if det:
                u = torch.Tensor(np.broadcast_to(np.linspace(0., 1., N_samples), (n, N_samples)).copy())
            else:
                u = torch.Tensor(np.random.rand(n, N_samples))
            u = u.to(b.device)
        elif not det:
            u = torch.rand(n, N_samples, device=b.device)
    uc = f32c(u) if u is not None else None
    out = torch.empty(n, N_samples, device=b.device, dtype=torch.float32)
    _req(w, n * (b.shape[1] - 1), b.device, "sample_pdf: weights")
    _req(uc, n * N_samples, b.device, "sample_pdf: u")
    check(lib.snr_sample_pdf(ptr(b), ptr(w), ptr(uc), n, b.shape[1], N_samples, ptr(out), stream()), "snr_sample_pdf")
    return out


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, white_bkgd, detach_weights, need_alpha):
        lib = _lib.load()
        raw_c, z, r = f32c(raw), f32c(z_vals), f32c(rays)
        n, S, C = raw_c.shape
        dev = raw_c.device
        rgb = torch.empty(n, 3, device=dev)
        disp = torch.empty(n, device=dev)
        acc = torch.empty(n, device=dev)
        depth = torch.empty(n, device=dev)
        w = torch.empty(n, S, device=dev)
        alpha = torch.empty(n, S, device=dev) if need_alpha else None
        nz = f32c(noise) if noise is not None else None
        _req(z, n * S, dev, "raw2outputs: z_vals")
        _req(nz, n * S, dev, "raw2outputs: noise")
        if r.device != dev or r.dim() != 2 or r.shape[0] != n or r.shape[1] < 6:
            raise ValueError(f"raw2outputs: rays must be [{n}, >= 6] on {dev}, got {tuple(r.shape)} on {r.device}")
        check(lib.snr_composite_forward(ptr(raw_c), C, ptr(z), ptr(r), r.shape[1], ptr(nz), n, S,
                                        int(bool(white_bkgd)), ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(w),
                                        ptr(alpha), stream()), "snr_composite_forward")
        ctx.save_for_backward(raw_c, z, r, nz)
        ctx.flags = (int(bool(white_bkgd)), int(bool(detach_weights)), need_alpha)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None (= the kernel's NULL), not as zero fills
        return rgb, disp, acc, depth, w, alpha

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_w, g_alpha):
        lib = _lib.load()
        if all(g is None for g in (g_rgb, g_disp, g_acc, g_depth, g_w, g_alpha)):
            return None, None, None, None, None, None, None
        raw, z, r, nz = ctx.saved_tensors
        white, detach, need_alpha = ctx.flags
        n, S, C = raw.shape
        d_raw = torch.empty_like(raw)
        gs = [f32c(g) if g is not None else None for g in (g_rgb, g_disp, g_acc, g_depth, g_w, g_alpha)]
        check(lib.snr_composite_backward(ptr(raw), C, ptr(z), ptr(r), r.shape[1], ptr(nz), n, S, white, detach,
                                         ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]),
                                         ptr(gs[5]) if need_alpha else None, ptr(d_raw), stream()),
              "snr_composite_backward")
        return d_raw, None, None, None, None, None, None


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, need_alpha=False,
                detach_weights=False, noise=None, rays=None):
    """Same signature and 6-tuple as the reference's raw2outputs (helpers:350-401):
    (rgb_map, disp_map, acc_map, weights, depth_map, alpha|None).

    ``noise`` (pre-scaled, [N,S]) overrides the random draw; ``pytest`` reproduces the reference's
    numpy seed-0 *uniform* draw (helpers:377-380).  ``rays`` may carry the packed ray rows so that
    rays_d is read in place."""
    import numpy as np
    if noise is None and raw_noise_std > 0.:
        if pytest:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std).to(raw.device)
        else:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
    if rays is None:
        rays = torch.cat([torch.zeros_like(rays_d), rays_d], -1)
    rgb, disp, acc, depth, w, alpha = _Composite.apply(raw, z_vals, rays, noise, white_bkgd, detach_weights,
                                                       need_alpha)
    return rgb, disp, acc, w, depth, alpha


class _CompositeAlpha(torch.autograd.Function):
    """compositing from caller-computed opacities (snr_composite_alpha_forward / _backward)"""

    @staticmethod
    def forward(ctx, raw, alpha, z_vals, rays, white_bkgd):
        lib = _lib.load()
        raw_c, a, z, r = f32c(raw), f32c(alpha), f32c(z_vals), f32c(rays)
        n, S, C = raw_c.shape
        dev = raw_c.device
        rgb = torch.empty(n, 3, device=dev); disp = torch.empty(n, device=dev); acc = torch.empty(n, device=dev)
        depth = torch.empty(n, device=dev); w = torch.empty(n, S, device=dev)
        _req(a, n * S, dev, "composite from alpha: alpha")
        _req(z, n * S, dev, "composite from alpha: z_vals")
        if r.device != dev or r.dim() != 2 or r.shape[0] != n or r.shape[1] < 6:
            raise ValueError(f"composite from alpha: rays must be [{n}, >= 6] on {dev}, got {tuple(r.shape)} on {r.device}")
        check(lib.snr_composite_alpha_forward(ptr(raw_c), C, ptr(z), ptr(r), r.shape[1], ptr(a), n, S, int(bool(white_bkgd)),
                                              ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(w), stream()),
              "snr_composite_alpha_forward")
        ctx.save_for_backward(raw_c, a, z, r)
        ctx.white = int(bool(white_bkgd))
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, depth, w

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_w):
        lib = _lib.load()
        raw, a, z, r = ctx.saved_tensors
        n, S, C = raw.shape
        d_raw, d_alpha = torch.empty_like(raw), torch.empty_like(a)
        gs = [f32c(g) if g is not None else None for g in (g_rgb, g_disp, g_acc, g_depth, g_w)]
        check(lib.snr_composite_alpha_backward(ptr(raw), C, ptr(z), ptr(r), r.shape[1], ptr(a), n, S, ctx.white, 0, ptr(gs[0]),
                                               ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]), ptr(d_raw), ptr(d_alpha),
                                               stream()), "snr_composite_alpha_backward")
        return d_raw, d_alpha, None, None, None


def raw2outputs_mvseg(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, only_object=False,
                      threshold=None, harsh_bg_remove=False, noise=None, rays=None):
    """MVSeg's 5-channel raw2outputs with its signature (MVSeg/DS_NeRF/run_nerf_helpers.py:350-413): the logit channel
    is composited with the DETACHED weights (`prob_map = sum(w.detach() * logit)`, :405).

    Default path: the compositing kernel handles channels 0..3 and zeroes the gradient of channel 4.
    ``only_object`` (:383-397, 410-411): the opacity is multiplied by 1 - sigmoid(logit), optionally zeroed above
    ``threshold`` and box-smoothed 5 times along the ray, BEFORE the transmittance product — those few element-wise steps
    on [N_rays, S] are torch ops feeding the kernel's composite-from-alpha entry point (MVSeg's evaluation path, not the
    training hot path); ``harsh_bg_remove`` subtracts 10 (1 - acc) from prob_map.
    Returns (rgb_map, disp_map, acc_map, weights, depth_map, prob_map, logits)."""
    import numpy as np
    if raw.shape[-1] < 5:
        raise ValueError("raw2outputs_mvseg needs 5 raw channels (rgb, sigma, logit)")
    logits = raw[..., 4]
    if not only_object:
        rgb, disp, acc, w, depth, _ = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest, noise=noise,
                                                  rays=rays)
    else:
        if noise is None and raw_noise_std > 0.:
            if pytest:
                np.random.seed(0)
                noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std).to(raw.device)
            else:
                noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(rays_d[..., None, :], dim=-1)
        sig = raw[..., 3] if noise is None else raw[..., 3] + noise
        alpha = (1. - torch.exp(-torch.relu(sig) * dists)) * (1 - torch.sigmoid(logits))
        if threshold is not None:
            alpha = torch.where(alpha > threshold, torch.zeros_like(alpha), alpha)
            zero = torch.zeros_like(alpha[:, :1])
            for _ in range(5):
                alpha = (torch.cat([zero, alpha[:, :-1]], 1) + alpha + torch.cat([alpha[:, 1:], zero], 1)) / 3
        if rays is None:
            rays = torch.cat([torch.zeros_like(rays_d), rays_d], -1)
        rgb, disp, acc, depth, w = _CompositeAlpha.apply(raw, alpha, z_vals, rays, white_bkgd)
    prob = torch.sum(w.detach() * logits, -1)
    if only_object and harsh_bg_remove:
        prob = prob - 10 * (1. - acc)
    return rgb, disp, acc, w, depth, prob, logits


# ----------------------------------------------------------------------------------------------
# rays (helpers:249-300)
# ----------------------------------------------------------------------------------------------
def make_rays(H, W, focal, c2w, patch=None, ndc=True, near=0., far=1., use_viewdirs=False, device=None):
    """Packed ray rows [h*w, 8 or 11] = o d near far (viewdirs) for a full frame or a patch
    (run_nerf.py:117-153 fused with get_rays / ndc_rays)."""
    import ctypes
    lib = _lib.load()
    c = torch.as_tensor(c2w, dtype=torch.float32).detach().cpu()[:3, :4].contiguous()
    arr = (ctypes.c_float * 12)(*c.reshape(-1).tolist())
    i0, j0, h, w = (0, 0, H, W) if patch is None else patch
    ld = 11 if use_viewdirs else 8
    rays = torch.empty(h * w, ld, device=device or torch.device("cuda"), dtype=torch.float32)
    check(lib.snr_make_rays(int(H), int(W), float(focal), arr, int(i0), int(j0), int(h), int(w), int(bool(ndc)),
                            float(near), float(far), int(bool(use_viewdirs)), ptr(rays), ld, stream()),
          "snr_make_rays")
    return rays


def pack_rays(rays_o, rays_d, H, W, focal, ndc=True, near=0., far=1., use_viewdirs=False, view_src=None, depths=None,
              ndc_near=1.):
    """Packed rows [o d near far (depth) (viewdirs)] for rays the caller already holds (run_nerf.py:117-153): one
    kernel instead of norm / div / ones / cat.  ``near`` / ``far`` may be per-ray tensors (:106-107), ``view_src`` =
    the directions viewdirs are taken from (c2w_staticcam, :131-133), ``depths`` = the COLMAP depth column (:148-149)."""
    lib = _lib.load()
    o, d = f32c(rays_o.detach().reshape(-1, 3)), f32c(rays_d.detach().reshape(-1, 3))
    n = o.shape[0]

    def rows(t):   # per-ray bound: anything that broadcasts against [n, 1] like the reference's near * ones_like(...)
        return f32c(t.detach().to(o.device).reshape(-1, 1).expand(n, 1).reshape(-1))
    near_t = rows(near) if isinstance(near, torch.Tensor) else None
    far_t = rows(far) if isinstance(far, torch.Tensor) else None
    vs = f32c(view_src.detach().reshape(-1, 3)) if view_src is not None else None
    dp = f32c(depths.detach().reshape(-1)) if depths is not None else None
    ld = 8 + (1 if dp is not None else 0) + (3 if use_viewdirs else 0)
    rays = torch.empty(n, ld, device=o.device, dtype=torch.float32)
    check(lib.snr_pack_rays(ptr(o), ptr(d), ptr(vs), n, int(H), int(W), float(focal), int(bool(ndc)), float(ndc_near),
                            0. if near_t is not None else float(near), 0. if far_t is not None else float(far),
                            ptr(near_t), ptr(far_t), ptr(dp), int(bool(use_viewdirs)), ptr(rays), ld, stream()),
          "snr_pack_rays")
    return rays


def embed(x, multires):
    """Embedder.embed (helpers:22-52) as a standalone op: [..., C] -> [..., C * (1 + 2 * multires)]."""
    lib = _lib.load()
    xc = f32c(x.detach().reshape(-1, x.shape[-1]))
    out = torch.empty(xc.shape[0], xc.shape[1] * (1 + 2 * multires), device=xc.device, dtype=torch.float32)
    if xc.shape[0]:
        check(lib.snr_embed(ptr(xc), xc.shape[0], xc.shape[1], int(multires), ptr(out), stream()), "snr_embed")
    return out.reshape(list(x.shape[:-1]) + [out.shape[-1]])


def mse_pair(a, b, target):
    """(loss, fine-term, d loss/d a, d loss/d b) for loss = mean((a-t)^2) [+ mean((b-t)^2)] in one launch."""
    lib = _lib.load()
    a_c, t_c = f32c(a.detach()), f32c(target.detach())
    b_c = f32c(b.detach()) if b is not None else None
    loss = torch.empty(2, device=a_c.device, dtype=torch.float32)
    ga = torch.empty_like(a_c)
    gb = torch.empty_like(b_c) if b_c is not None else None
    _req(t_c, a_c.numel(), a_c.device, "mse_pair: target")
    _req(b_c, a_c.numel(), a_c.device, "mse_pair: second map")
    check(lib.snr_mse_pair(ptr(a_c), ptr(b_c), ptr(t_c), a_c.numel(), ptr(loss), ptr(ga), ptr(gb), stream()),
          "snr_mse_pair")
    return loss[0], loss[1], ga, gb


# ----------------------------------------------------------------------------------------------
# NeRF MLP (run_nerf.py:56-71; helpers:104-127)
# ----------------------------------------------------------------------------------------------
class _Mlp(torch.autograd.Function):
    """raw = MLP(flat_params; sample positions, view directions).  Gradient flows to the flat
    parameter buffer only (nothing upstream of the encodings needs one)."""

    @staticmethod
    def forward(ctx, flat, net, pts, rays, z_vals, viewdirs, n_samples, S):
        lib = _lib.load()
        cfg = net.cfg
        packed = net.packed_weights()
        raw = torch.empty(n_samples, cfg.out_ch, device=flat.device, dtype=torch.float32)
        need_grad = ctx.needs_input_grad[0]
        act = None
        if need_grad:
            nbytes = lib.snr_mlp_act_bytes(cfg, n_samples)
            if nbytes <= 0:
                check(int(nbytes), "snr_mlp_act_bytes")
            act = torch.empty(nbytes, device=flat.device, dtype=torch.uint8)
        vd_ld = viewdirs.stride(0) if viewdirs is not None else 0
        check(lib.snr_mlp_forward(cfg, ptr(packed), ptr(pts), ptr(rays), rays.shape[1] if rays is not None else 0,
                                  ptr(z_vals), ptr(viewdirs), vd_ld, n_samples, S, ptr(raw), ptr(act), stream()),
              "snr_mlp_forward")
        ctx.net, ctx.n_samples = net, n_samples
        ctx.act, ctx.packed = act, packed
        # the packed blob is re-packed IN PLACE when the parameters or the precision change: remember which pack this
        # graph was recorded against
        ctx.pack_gen, ctx.cfg_key = net.pack_generation, cfg.key()
        ctx.flat_key = (flat._version, net.weights_generation)
        ctx.set_materialize_grads(False)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        if d_raw is None:
            ctx.act = None
            return None, None, None, None, None, None, None, None
        lib = _lib.load()
        net, n = ctx.net, ctx.n_samples
        cfg = net.cfg
        if ctx.act is None:
            raise RuntimeError("the saved activations of this MLP evaluation were released by its first backward "
                               "(retain_graph / a second backward through the same forward is not supported)")
        if (net.pack_generation != ctx.pack_gen or cfg.key() != ctx.cfg_key
                or (net.flat._version, net.weights_generation) != ctx.flat_key):
            raise RuntimeError("the network's packed weights changed between forward and backward (optimizer step, "
                               "load_state_dict or set_precision in between): re-run the forward")
        g = torch.empty_like(net.flat)
        ws_bytes = lib.snr_mlp_bwd_ws_bytes(cfg, n)
        if ws_bytes <= 0:
            check(int(ws_bytes), "snr_mlp_bwd_ws_bytes")
        ws = torch.empty(ws_bytes, device=g.device, dtype=torch.uint8)
        check(lib.snr_mlp_backward(cfg, ptr(ctx.packed), ptr(net.flat.detach()), ptr(f32c(d_raw)), n, ptr(ctx.act),
                                   ptr(ws), ptr(g), 0, stream()), "snr_mlp_backward")
        ctx.act = None
        return g, None, None, None, None, None, None, None


def mlp_query(net, pts=None, rays=None, z_vals=None, viewdirs=None, samples_per_ray=1):
    """Low-level entry: either ``pts`` [M,3] or (``rays`` rows + ``z_vals`` [N,S]).  ``viewdirs`` is a
    [N,>=3] (possibly strided) view whose first three columns are the unit directions."""
    if pts is not None:
        pts = f32c(pts.detach())
        M = pts.shape[0]
    else:
        rays, z_vals = f32c(rays.detach()), f32c(z_vals.detach())
        M = z_vals.numel()
        samples_per_ray = z_vals.shape[1]
    if net.cfg.use_viewdirs:
        if viewdirs is None:
            raise ValueError("this network was built with use_viewdirs=True; viewdirs is required")
        viewdirs = viewdirs.detach()
        if viewdirs.dtype != torch.float32 or viewdirs.stride(-1) != 1:
            viewdirs = f32c(viewdirs)
    else:
        viewdirs = None
    # raw pointers from here on: shapes and devices the kernel will assume
    dev = net.flat.device
    n_rays = -(-M // max(int(samples_per_ray), 1))
    if pts is not None:
        if pts.device != dev or pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError(f"mlp_query: pts must be [M, 3] on {dev}, got {tuple(pts.shape)} on {pts.device}")
    else:
        if rays.device != dev or z_vals.device != dev or rays.dim() != 2 or rays.shape[0] != z_vals.shape[0] or rays.shape[1] < 6:
            raise ValueError(f"mlp_query: rays [N, >= 6] and z_vals [N, S] on {dev} expected, got {tuple(rays.shape)} / {tuple(z_vals.shape)}")
    if viewdirs is not None and (viewdirs.device != dev or viewdirs.dim() != 2 or viewdirs.shape[0] < n_rays or viewdirs.shape[1] < 3):
        raise ValueError(f"mlp_query: viewdirs must be [>= {n_rays}, >= 3] on {dev}, got {tuple(viewdirs.shape)} on {viewdirs.device}")
    # Function.forward always runs with grad mode off, and needs_input_grad stays True under torch.no_grad(): a detached
    # buffer is what tells it that this evaluation saves nothing (inference kernel, no activation workspace)
    flat = net.flat if torch.is_grad_enabled() else net.flat.detach()
    return _Mlp.apply(flat, net, pts, rays, z_vals, viewdirs, M, samples_per_ray)


# ----------------------------------------------------------------------------------------------
# the training step's production forms: in-kernel random draws, loss folded into the compositing kernel
# ----------------------------------------------------------------------------------------------
def sample_coarse_rng(ray_batch, N_samples, lindisp, seed, offset):
    lib = _lib.load()
    n = ray_batch.shape[0]
    z = torch.empty(n, N_samples, device=ray_batch.device, dtype=torch.float32)
    check(lib.snr_sample_coarse_rng(ptr(ray_batch), ray_batch.shape[1], n, N_samples, int(bool(lindisp)), int(seed),
                                    int(offset), ptr(z), stream()), "snr_sample_coarse_rng")
    return z


def sample_fine_rng(z_coarse, weights, N_importance, seed, offset):
    lib = _lib.load()
    n, nc = z_coarse.shape
    z_out = torch.empty(n, nc + N_importance, device=z_coarse.device, dtype=torch.float32)
    check(lib.snr_sample_fine_rng(ptr(z_coarse), ptr(weights), n, nc, N_importance, int(seed), int(offset), ptr(z_out),
                                  None, None, stream()), "snr_sample_fine_rng")
    return z_out


def composite_train(raw, z_vals, rays, target, loss, loss_also=None, noise=None, noise_std=0., seed=0, offset=0,
                    white_bkgd=False, detach_weights=False):
    """raw2outputs + mean((rgb_map - target)^2) + the backward of both in one launch -> (rgb, disp, acc, depth, weights,
    d loss / d raw); the loss term is added to loss[0] (and loss_also[0])."""
    lib = _lib.load()
    n, S, C = raw.shape
    dev = raw.device
    rgb = torch.empty(n, 3, device=dev); disp = torch.empty(n, device=dev); acc = torch.empty(n, device=dev)
    depth = torch.empty(n, device=dev); w = torch.empty(n, S, device=dev); d_raw = torch.empty_like(raw)
    _req(z_vals, n * S, dev, "composite_train: z_vals")
    _req(target, 3 * n, dev, "composite_train: target")
    _req(noise, n * S, dev, "composite_train: noise")
    _req(loss, loss.numel(), dev, "composite_train: loss")
    _req(loss_also, loss_also.numel() if loss_also is not None else 0, dev, "composite_train: loss_also")
    if rays.device != dev or rays.shape[0] != n:
        raise ValueError(f"composite_train: rays must have {n} rows on {dev}")
    check(lib.snr_composite_train(ptr(raw), C, ptr(z_vals), ptr(rays), rays.shape[1], ptr(noise), float(noise_std),
                                  int(seed), int(offset), n, S, int(bool(white_bkgd)), int(bool(detach_weights)),
                                  ptr(target), n, ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(w), ptr(d_raw), ptr(loss),
                                  ptr(loss_also), stream()), "snr_composite_train")
    return rgb, disp, acc, depth, w, d_raw


def mlp_train_forward(net, rays, z_vals, viewdirs):
    """NeRF forward for the autograd-free training step: (raw [N,S,C], saved context for mlp_train_backward)."""
    if hasattr(net, "train_forward"):          # NeRF_TCNN (hashgrid.py) brings its own kernels
        return net.train_forward(rays, z_vals, viewdirs)
    lib = _lib.load()
    cfg = net.cfg
    packed = net.packed_weights()
    n = z_vals.numel()
    raw = torch.empty(z_vals.shape[0], z_vals.shape[1], cfg.out_ch, device=z_vals.device, dtype=torch.float32)
    nbytes = lib.snr_mlp_act_bytes(cfg, n)
    if nbytes <= 0:
        check(int(nbytes), "snr_mlp_act_bytes")
    act = torch.empty(nbytes, device=z_vals.device, dtype=torch.uint8)
    vd = viewdirs if cfg.use_viewdirs else None
    check(lib.snr_mlp_forward(cfg, ptr(packed), None, ptr(rays), rays.shape[1], ptr(z_vals), ptr(vd),
                              vd.stride(0) if vd is not None else 0, n, z_vals.shape[1], ptr(raw), ptr(act), stream()),
          "snr_mlp_forward")
    return raw, (packed, act, n)


def mlp_train_backward(net, saved, d_raw):
    if hasattr(net, "train_backward"):
        return net.train_backward(saved, d_raw)
    lib = _lib.load()
    packed, act, n = saved
    cfg = net.cfg
    g = torch.empty_like(net.flat.data)
    ws_bytes = lib.snr_mlp_bwd_ws_bytes(cfg, n)
    if ws_bytes <= 0:
        check(int(ws_bytes), "snr_mlp_bwd_ws_bytes")
    ws = torch.empty(ws_bytes, device=g.device, dtype=torch.uint8)
    check(lib.snr_mlp_backward(cfg, ptr(packed), ptr(net.flat.detach()), ptr(d_raw), n, ptr(act), ptr(ws), ptr(g), 0,
                               stream()), "snr_mlp_backward")
    return g


def mlp_train_backward_multi(nets, saveds, d_raws):
    """The backward passes of several NeRF MLPs as ONE launch sequence (snr_mlp_backward_multi): one chain launch, one
    weight-gradient launch, one reduce for all of them.  Returns the flat gradients in the order of ``nets``."""
    import ctypes
    lib = _lib.load()
    items = (_lib.MlpBwdItem * len(nets))()
    keep, grads = [], []
    for i, (net, (packed, act, n), d_raw) in enumerate(zip(nets, saveds, d_raws)):
        cfg = net.cfg
        g = torch.empty_like(net.flat.data)
        ws_bytes = lib.snr_mlp_bwd_ws_bytes(cfg, n)
        if ws_bytes <= 0:
            check(int(ws_bytes), "snr_mlp_bwd_ws_bytes")
        ws = torch.empty(ws_bytes, device=g.device, dtype=torch.uint8)
        items[i] = _lib.MlpBwdItem(ctypes.pointer(cfg), ptr(packed), ptr(net.flat.detach()), ptr(d_raw), n, ptr(act), ptr(ws),
                                   ptr(g), 0)
        keep.append((ws, d_raw, packed, act))
        grads.append(g)
    check(lib.snr_mlp_backward_multi(items, len(nets), stream()), "snr_mlp_backward_multi")
    return grads


# ----------------------------------------------------------------------------------------------
# render_rays as one library call (snr_render_rays_fused_*): the training step's launch sequence is enqueued by the
# library, every intermediate lives in one workspace
# ----------------------------------------------------------------------------------------------
class FusedRender:
    """handle of one fused training forward: the workspace the backward consumes and views of what it holds"""

    def __init__(self, rc, nets, keep, rays, n, ws, layout, maps, loss):
        self.rc, self.nets, self.keep, self.rays, self.n, self.ws, self.layout = rc, nets, keep, rays, n, ws, layout
        self.rgb, self.disp, self.acc, self.depth, self.rgb0, self.disp0, self.acc0, self.z_std = maps
        self.loss = loss

    def view(self, name, *shape):
        """tensor view of a workspace section ('z_vals', 'weights', 'raw', 'z_coarse', 'weights0', 'raw0', 'z_samples')"""
        off = getattr(self.layout, name)
        if off < 0:
            raise KeyError(name)
        count = 1
        for d in shape:
            count *= d
        return self.ws[off:off + 4 * count].view(torch.float32).view(*shape)


def _net_struct(net):
    packed = net.packed_weights()
    if hasattr(net, "cfg"):
        s = _lib.Net(_lib.NET_MLP, net.cfg, ptr(packed), ptr(net.flat.detach()))
    else:
        s = _lib.Net(_lib.NET_HASHGRID, _lib.MlpConfig(), ptr(packed), ptr(net.flat.detach()))
    return s, packed


def fused_forward(net_c, net_f, rays, N_samples, N_importance, lindisp, white_bkgd, perturb, raw_noise_std, seed, offset,
                  target, loss, n_rays_global=None, randoms=None, offset_base=None, prepare=None, loss_terms=None, guard_term=-1):
    """render_rays + the loss terms + the compositing backward of one training step in one library call.  ``net_f`` None
    with N_importance > 0 = the coarse network evaluated twice.  Consumes the Philox offsets offset+1 .. offset+4.
    Returns a FusedRender; fused_backward(handle) gives the parameter gradients.  ``target`` None (and ``loss`` None) =
    inference: forward only, nothing saved for a backward.  ``offset_base`` = device tensor holding a snr_step_state (its
    first field is added to the draw offsets at run time: graph replays).
    ``prepare`` = dict(rays_o, rays_d, H, W, focal, ndc, near, far, use_viewdirs) instead of packed ``rays``: the packed rows,
    the stratified z_vals and the zero fill of ``loss`` are made by ONE launch (snr_render_step_prepare) in front of the
    forward; the rows are in the returned handle (``.rays``).
    ``loss_terms`` (instead of ``target``) = list of dicts(first, n, kind, target, count=None, slot, slot_final=-1): the loss as
    terms over ray ranges (snr_render_rays_fused_forward_terms; kinds _lib.LOSS_*), ``guard_term`` the index of the NaN-guarded
    one; ``loss`` then has 4 slots (zeroed by the prepare launch, or by the caller)."""
    import ctypes
    lib = _lib.load()
    rnd = randoms or {}
    if prepare is not None:
        ro, rd = f32c(prepare["rays_o"]), f32c(prepare["rays_d"])
        n = ro.shape[0]
        rays = torch.empty(n, 11 if prepare["use_viewdirs"] else 8, device=ro.device, dtype=torch.float32)
    n = rays.shape[0]
    flags = (_lib.RENDER_Z_COARSE_READY if prepare is not None else 0) | (_lib.RENDER_LOSS4 if loss_terms is not None else 0)
    rc = _lib.RenderConfig(int(N_samples), int(N_importance), int(bool(lindisp)), int(bool(white_bkgd)),
                           int(perturb > 0.), float(raw_noise_std), flags)
    train = target is not None or loss_terms is not None
    sc, pc = _net_struct(net_c)
    two = N_importance > 0 and net_f is not None and net_f is not net_c
    sf, pf = _net_struct(net_f) if two else (None, None)
    L = _lib.RenderWsLayout()
    fptr = ctypes.byref(sf) if two else None
    check(lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(sc), fptr, n, int(train), ctypes.byref(L)),
          "snr_render_rays_fused_layout")
    dev = rays.device
    ws = torch.empty(L.total, device=dev, dtype=torch.uint8)
    maps8 = torch.empty(8, n, 3, device=dev, dtype=torch.float32)      # one allocation for the eight output maps
    rgb, rgb0 = maps8[0], maps8[1]
    flat = maps8.view(8, -1)
    disp, acc, depth, disp0, acc0, z_std = (flat[2 + k, :n] for k in range(6))   # the first n floats of a slab each
    arr = {k: (f32c(rnd[k]) if rnd.get(k) is not None else None) for k in ("t_rand", "u", "noise_c", "noise_f")}
    if prepare is not None:
        z_coarse = ws[L.z_coarse:]
        check(lib.snr_render_step_prepare(ctypes.byref(rc), ptr(ro), ptr(rd), n, int(prepare["H"]), int(prepare["W"]),
                                          float(prepare["focal"]), int(bool(prepare["ndc"])), float(prepare["near"]),
                                          float(prepare["far"]), int(bool(prepare["use_viewdirs"])), ptr(rays), rays.shape[1],
                                          ptr(arr["t_rand"]), int(seed), int(offset), ptr(offset_base), ptr(z_coarse), ptr(loss),
                                          stream()), "snr_render_step_prepare")
    if loss_terms is not None:
        lt = _lib.LossTerms()
        lt.n_terms, lt.guard_term = len(loss_terms), int(guard_term)
        keep = []
        for k, t in enumerate(loss_terms):
            tg = f32c(t["target"])
            per_ray = 1 if int(t["kind"]) == _lib.LOSS_DISP else 3
            if not tg.is_cuda or tg.device != dev or tg.numel() < per_ray * int(t["n"]):
                raise _lib.HipLibraryError(f"loss term {k}: the target must be a tensor of {per_ray} x {int(t['n'])} elements on {dev} "
                                           f"(got {tuple(tg.shape)} on {tg.device})")
            keep.append(tg)
            lt.term[k] = _lib.LossTerm(int(t["first"]), int(t["n"]), int(t["kind"]), tg.data_ptr(),
                                       int(t.get("count") or t["n"]), int(t["slot"]), int(t.get("slot_final", -1)))
        arr["_targets"] = keep
        check(lib.snr_render_rays_fused_forward_terms(
            ctypes.byref(rc), ctypes.byref(sc), fptr, ptr(rays), rays.shape[1], n, ptr(arr["t_rand"]), ptr(arr["u"]),
            ptr(arr["noise_c"]), ptr(arr["noise_f"]), int(seed), int(offset), ptr(offset_base), ctypes.byref(lt), ptr(ws),
            ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(rgb0), ptr(disp0), ptr(acc0), ptr(z_std), ptr(loss), stream()),
            "snr_render_rays_fused_forward_terms")
        return FusedRender(rc, (sc, sf), (pc, pf, arr), rays, n, ws, L, (rgb, disp, acc, depth, rgb0, disp0, acc0, z_std), loss)
    if target is not None and (not target.is_cuda or target.device != dev or target.numel() < 3 * n):
        raise _lib.HipLibraryError(f"the target must be a tensor of 3 x {n} elements on {dev} (got {tuple(target.shape)} on {target.device})")
    check(lib.snr_render_rays_fused_forward(
        ctypes.byref(rc), ctypes.byref(sc), fptr, ptr(rays), rays.shape[1], n, ptr(arr["t_rand"]), ptr(arr["u"]),
        ptr(arr["noise_c"]), ptr(arr["noise_f"]), int(seed), int(offset), ptr(offset_base), ptr(target),
        int(n_rays_global or n), ptr(ws),
        ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(rgb0), ptr(disp0), ptr(acc0), ptr(z_std), ptr(loss), stream()),
        "snr_render_rays_fused_forward")
    return FusedRender(rc, (sc, sf), (pc, pf, arr), rays, n, ws, L, (rgb, disp, acc, depth, rgb0, disp0, acc0, z_std), loss)


PASS_COARSE, PASS_FINE = 1, 2


def fused_backward(h, grad_c, grad_f=None, accumulate=False, passes=PASS_COARSE | PASS_FINE):
    """parameter gradients of a fused training forward into the flat buffers grad_c / grad_f"""
    import ctypes
    lib = _lib.load()
    sc, sf = h.nets
    check(lib.snr_render_rays_fused_backward(ctypes.byref(h.rc), ctypes.byref(sc), ctypes.byref(sf) if sf is not None else None,
                                             ptr(h.rays), h.rays.shape[1], h.n, ptr(h.ws), ptr(grad_c), ptr(grad_f),
                                             int(bool(accumulate)), int(passes), stream()), "snr_render_rays_fused_backward")


# ----------------------------------------------------------------------------------------------
# Adam on a flat buffer (run_nerf.py:433-434)
# ----------------------------------------------------------------------------------------------
def adam_step_(params, grads, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    lib = _lib.load()
    check(lib.snr_adam_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), float(lr),
                            float(beta1), float(beta2), float(eps), int(step), float(grad_scale), stream()),
          "snr_adam_step")


def adam_pack_step_(nets, grads, exp_avgs, exp_avg_sqs, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """Adam on the flat parameter buffers of 1..2 NeRF MLPs AND the re-pack of their weights, one launch
    (snr_adam_pack_multi).  The networks' packed blobs are rewritten in place: the next forward needs no snr_mlp_pack."""
    import ctypes
    lib = _lib.load()
    items = (_lib.AdamPackItem * len(nets))()
    keep = []
    for i, (net, g, m, v) in enumerate(zip(nets, grads, exp_avgs, exp_avg_sqs)):
        packed = net.packed_weights()      # (packs once if the blob is missing or stale: its padding must exist)
        items[i] = _lib.AdamPackItem(ctypes.pointer(net.cfg), ptr(net.flat.data), ptr(g), ptr(m), ptr(v), ptr(packed))
        keep.append((packed, g))
    st = lib.snr_adam_pack_multi(items, len(nets), float(lr), float(beta1), float(beta2), float(eps), int(step),
                                 float(grad_scale), None, stream())
    if st == _lib.ERR_UNSUPPORTED:
        return False      # a shape the fused kernel does not cover: nothing was launched (the caller groups equal configurations only)
    check(st, "snr_adam_pack_multi")
    for net in nets:
        net.note_packed_in_place()
    return True
