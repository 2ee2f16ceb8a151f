"""Frame loop over camera poses on top of the HIP render path — the caller SURVEY.md §8 (f-2) ranks next.

Mirrors `render_path` of the reference (DS_NeRF/run_nerf.py:168-307): same arguments, same return value
`(rgbs, disps, (Xs, Ys))`, same on-disk dump (`intrinsics.txt`, `rgb/%06d.png`, `images/%06d.png`,
`depth|disp|weight|z[|alpha]/%06d.npy`, `pose/%06d.txt`), the random-patch mode used for the
perceptual-loss feed (`:196-215`) included.  Every frame is one `render(c2w=...)`: rays, viewdirs, NDC
and packing are produced on the GPU (`snr_make_rays`), so a 378x504 frame is ~6 chunked launches of the
fused kernels and nothing else.

PNG output does not depend on imageio / cv2 (absent on the GPU image): `write_png` is a 30-line
zlib encoder for 8-bit RGB / grey images.
"""
import os
import random
import struct
import zlib

import numpy as np
import torch

from .render import render


def to8b(x):
    """run_nerf_helpers.py:17"""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def write_png(filename, img):
    """8-bit grey [H,W] or RGB [H,W,3] array -> PNG (filter 0, one IDAT)."""
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
        raise ValueError("write_png expects uint8 [H,W] or [H,W,3]")
    h, w = img.shape[:2]
    color = 2 if img.ndim == 3 else 0
    rows = img.reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    with open(filename, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def render_path(render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                disp_require_grad=False, need_alpha=False, rgb_require_grad=False, detach_weights=False,
                patch_len=None, masks=None):
    H, W, focal = hwf
    if render_factor != 0:
        # render downsampled for speed (:172-176)
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor

    if savedir is not None:
        K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
        np.savetxt(os.path.join(savedir, 'intrinsics.txt'), K)
        names = ['rgb', 'depth', 'images', 'weight', 'z', 'pose', 'disp'] + (['alpha'] if need_alpha else [])
        dirs = {n: os.path.join(savedir, n) for n in names}
        for d in dirs.values():
            os.makedirs(d, exist_ok=True)

    rgbs, disps, Xs, Ys = [], [], [], []
    for i, c2w in enumerate(render_poses):
        if disp_require_grad or rgb_require_grad:
            patch = None
            if patch_len is not None:
                # top-left corner drawn inside the bounding box of the frame's mask (:197-209)
                m = np.where(_np(masks[i]) != 0)
                m = (m[0] // render_factor, m[1] // render_factor)
                Xs.append(random.randint(int(m[0].min()), int(max(m[0].max() - patch_len[0], m[0].min()))))
                Ys.append(random.randint(int(m[1].min()), int(max(m[1].max() - patch_len[1], m[1].min()))))
                patch = (Xs[-1], Ys[-1], patch_len[0], patch_len[1])
            rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True,
                                                   need_alpha=need_alpha, detach_weights=detach_weights,
                                                   patch=patch, **render_kwargs)
        else:
            with torch.no_grad():
                rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True,
                                                       need_alpha=need_alpha, **render_kwargs)

        disps.append(disp if disp_require_grad else _np(disp))
        rgbs.append(rgb if rgb_require_grad else _np(rgb))

        if savedir is not None:
            name = '{:06d}'.format(i)
            rgb8 = to8b(_np(rgbs[-1]))   # (the reference's NaN scrub on the uint8 image is a no-op)
            write_png(os.path.join(dirs['rgb'], name + '.png'), rgb8)
            if gt_imgs is not None:
                write_png(os.path.join(dirs['images'], name + '.png'), to8b(_np(gt_imgs[i])))
            np.save(os.path.join(dirs['depth'], name + '.npy'), _np(depth))
            np.save(os.path.join(dirs['disp'], name + '.npy'), _np(disp))
            np.save(os.path.join(dirs['weight'], name + '.npy'), _np(extras['weights']))
            np.save(os.path.join(dirs['z'], name + '.npy'), _np(extras['z_vals']))
            if need_alpha:
                np.save(os.path.join(dirs['alpha'], name + '.npy'), _np(extras['alpha']))
            pose = np.concatenate([_np(render_poses[i])[:3, :4], np.array([[0, 0, 0, 1]])], axis=0)
            np.savetxt(os.path.join(dirs['pose'], name + '.txt'), pose)

    disps = torch.stack(disps, 0) if disp_require_grad else np.stack(disps, 0)
    rgbs = torch.stack(rgbs, 0) if rgb_require_grad else np.stack(rgbs, 0)
    return rgbs, disps, (Xs, Ys)


def render_sharded(H, W, focal, c2w, chunk, render_kwargs, group=None, render_fn=None):
    """Full frame on all ranks of a process group: rank r renders the row band
    [r*ceil(H/world), ...) through render()'s patch mode, then one all_gather of [rows, W, 6] =
    rgb(3) | disp | acc | depth per rank assembles the frame everywhere.  Rays are independent, so this
    is the only collective of the inference path; without an initialised process group it is render().

    Returns [rgb_map[H,W,3], disp_map[H,W], acc_map[H,W], depth_map[H,W]] (no extras: the per-sample
    tensors stay on the rank that produced them)."""
    import torch.distributed as dist
    render_fn = render_fn or render
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    rows = -(-H // world)                       # rows per rank (the last band may be shorter or empty)
    i0 = min(rank * rows, H)
    h = max(0, min(rows, H - i0))
    dev = c2w.device if isinstance(c2w, torch.Tensor) else torch.device("cpu")
    band = torch.zeros(rows, W, 6, device=dev, dtype=torch.float32)
    if h > 0:
        with torch.no_grad():
            rgb, disp, acc, depth, _ = render_fn(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], patch=(i0, 0, h, W),
                                                 **render_kwargs)
        band[:h, :, 0:3] = rgb
        band[:h, :, 3], band[:h, :, 4], band[:h, :, 5] = disp, acc, depth
    if world > 1:
        parts = [torch.empty_like(band) for _ in range(world)]
        dist.all_gather(parts, band, group=group)
        frame = torch.cat(parts, 0)[:H]
    else:
        frame = band[:H]
    return [frame[..., 0:3], frame[..., 3], frame[..., 4], frame[..., 5]]


def convert_pose(C2W):
    """run_nerf.py:342-347: flip the y and z axes of a 4x4 camera-to-world matrix (OpenGL <-> OpenCV)."""
    flip_yz = np.eye(4)
    flip_yz[1, 1] = -1
    flip_yz[2, 2] = -1
    return np.matmul(C2W, flip_yz)


def render_path_projection(render_poses, hwf, chunk, render_kwargs, render_factor=0):
    """run_nerf.py:310-339: per pose the fine pass's sample depths and weights (numpy) plus the converted pose and
    the intrinsics — the inputs of the mask-projection tooling."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    z_vals, weights, c2ws = [], [], []
    for i, c2w in enumerate(render_poses):
        with torch.no_grad():
            _, _, _, _, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True, **render_kwargs)
        z_vals.append(_np(extras['z_vals']))
        weights.append(_np(extras['weights']))
        c2ws.append(convert_pose(np.concatenate([_np(render_poses[i])[:3, :4], np.array([[0, 0, 0, 1]])], axis=0)))
    return z_vals, weights, c2ws, K


def sample_sigma(rays_o, rays_d, viewdirs, network, z_vals, network_query):
    """run_nerf_helpers.py:404-417: colour and density along given depths plus the composited depth.  (The
    reference unpacks five values from its six-valued raw2outputs there and cannot run as written; this returns
    what it means to: rgb [N,S,3], sigma [N,S], depth_map [N].)"""
    from .ops import raw2outputs
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = network_query(pts, viewdirs, network)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = torch.relu(raw[..., 3])
    depth_map = raw2outputs(raw, z_vals, rays_d)[4]
    return rgb, sigma, depth_map


def render_test_ray(rays_o, rays_d, hwf, ndc, near, far, use_viewdirs, N_samples, network, network_query_fn, **kwargs):
    """run_nerf.py:350-377: evenly spaced depths between near and far for the given rays, one network query."""
    from .render import ndc_rays
    H, W, focal = hwf
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    near_t, far_t = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
    t_vals = torch.linspace(0., 1., steps=N_samples, device=rays_d.device)
    z_vals = (near_t * (1. - t_vals) + far_t * t_vals).reshape([rays_o.shape[0], N_samples])
    rgb, sigma, depth_maps = sample_sigma(rays_o, rays_d, viewdirs, network, z_vals, network_query_fn)
    return rgb, sigma, z_vals, depth_maps
