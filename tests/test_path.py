"""render_path (DS_NeRF/run_nerf.py:168-307): PNG writer on CPU; frame loop, dump formats and
patch mode on the GPU against render() itself (whose parity with the reference fixture
`render_c2w_fine_vd` is test_gpu_render.py's job)."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from helpers import load, T


def _read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xffffffff
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    w, h, depth, color = hdr[:4]
    assert depth == 8 and color in (0, 2)
    ch = 3 if color == 2 else 1
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    assert (raw[:, 0] == 0).all()
    img = raw[:, 1:].reshape(h, w, ch)
    return img if ch == 3 else img[..., 0]


def test_write_png_round_trip(tmp_path):
    import importlib
    P = importlib.import_module("spin-nerf_amd.path")
    rs = np.random.RandomState(0)
    for shape in ((7, 5, 3), (4, 9)):
        img = rs.randint(0, 256, size=shape).astype(np.uint8)
        f = str(tmp_path / "a.png")
        P.write_png(f, img)
        assert np.array_equal(_read_png(f), img)
    with pytest.raises(ValueError):
        P.write_png(str(tmp_path / "b.png"), np.zeros((3, 3), np.float32))
    assert P.to8b(np.array([-1.0, 0.5, 2.0])).tolist() == [0, 127, 255]


@pytest.mark.gpu
def test_render_path_frames_dumps_and_patches(tmp_path):
    import spin_nerf_amd as S
    from test_gpu_render import build
    g = load("render_c2w_fine_vd")
    _, _, kw = build(S, g)
    H, W, f, chunk = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["chunk"])
    c2w = T(g["c2w"]).cuda()
    c2w_b = c2w.clone(); c2w_b[:3, 3] += 0.05
    poses = torch.stack([c2w, c2w_b], 0)
    gt = np.random.RandomState(1).rand(2, H, W, 3).astype(np.float32)
    out = str(tmp_path)
    rgbs, disps, (Xs, Ys) = S.render_path(poses, (H, W, f), chunk, kw, gt_imgs=gt, savedir=out, need_alpha=True)
    assert isinstance(rgbs, np.ndarray) and rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W) and Xs == [] == Ys
    for i in range(2):
        with torch.no_grad():
            rgb, disp, acc, depth, ex = S.render(H, W, f, chunk=chunk, c2w=poses[i, :3, :4], retraw=True,
                                                 need_alpha=True, **kw)
        assert np.array_equal(rgbs[i], rgb.cpu().numpy()) and np.array_equal(disps[i], disp.cpu().numpy())
        n = "%06d" % i
        assert np.array_equal(_read_png(os.path.join(out, "rgb", n + ".png")), S.to8b(rgbs[i]))
        assert np.array_equal(_read_png(os.path.join(out, "images", n + ".png")), S.to8b(gt[i]))
        assert np.array_equal(np.load(os.path.join(out, "depth", n + ".npy")), depth.cpu().numpy())
        assert np.array_equal(np.load(os.path.join(out, "disp", n + ".npy")), disp.cpu().numpy())
        assert np.load(os.path.join(out, "weight", n + ".npy")).shape == (H, W, 64 + int(g["Nf"]))
        assert np.array_equal(np.load(os.path.join(out, "z", n + ".npy")), ex["z_vals"].cpu().numpy())
        assert np.array_equal(np.load(os.path.join(out, "alpha", n + ".npy")), ex["alpha"].cpu().numpy())
        pose = np.loadtxt(os.path.join(out, "pose", n + ".txt"))
        assert pose.shape == (4, 4) and np.allclose(pose[:3], poses[i, :3, :4].cpu().numpy()) and pose[3].tolist() == [0, 0, 0, 1]
    K = np.loadtxt(os.path.join(out, "intrinsics.txt"))
    assert np.allclose(K, [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]])
    # frame 0 against the reference-generated fixture (free-running gate of test_gpu_render.py: a few %
    # of the pixels may sit on an ill-conditioned resampling bin)
    ref = g["rgb"].reshape(H, W, 3)
    # measured (tests/golden/fp32_render_measured.json: render_path_frame0): 11 of 360 values past 1e-4, 24 past 1e-5, the
    # largest 5.0e-3 — gates = 1.5 x the counts, 2 x the maximum
    import json
    m = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp32_render_measured.json")))["render_path_frame0"]
    d = np.abs(rgbs[0] - ref)
    assert int((d > 1e-4).sum()) <= int(np.ceil(1.5 * m["over_1e4"])) and int((d > 1e-5).sum()) <= int(np.ceil(1.5 * m["over_1e5"]))
    assert float(d.max()) <= 2.0 * m["max"]

    # patch mode with gradients: a len1 x len2 window whose corner lies in the mask's bounding box; perturb=0 and
    # raw_noise_std=0 in this case, so the patch equals the same window of the full frame
    masks = np.zeros((2, H, W), np.uint8); masks[:, 2:9, 3:11] = 1
    prgb, pdisp, (Xs, Ys) = S.render_path(poses, (H, W, f), chunk, kw, render_factor=1, rgb_require_grad=True,
                                          disp_require_grad=True, patch_len=(4, 5), masks=masks)
    assert isinstance(prgb, torch.Tensor) and prgb.shape == (2, 4, 5, 3) and pdisp.shape == (2, 4, 5)
    assert prgb.requires_grad and len(Xs) == len(Ys) == 2
    for i in range(2):
        assert 2 <= Xs[i] <= 8 - 4 + 0 or Xs[i] == 2
        win = rgbs[i][Xs[i]:Xs[i] + 4, Ys[i]:Ys[i] + 5]
        assert np.allclose(prgb[i].detach().cpu().numpy(), win, atol=1e-6)
    prgb.sum().backward()
    # rgb of the fine pass reaches the fine network only: the resampled depths are detached (run_nerf.py:700)
    assert float(kw["network_fine"].flat.grad.abs().max()) > 0 and kw["network_fn"].flat.grad is None


@pytest.mark.gpu
def test_render_sharded_single_rank_is_render():
    import spin_nerf_amd as S
    from test_gpu_render import build
    g = load("render_c2w_fine_vd")
    _, _, kw = build(S, g)
    H, W, f, chunk = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["chunk"])
    c2w = T(g["c2w"]).cuda()
    with torch.no_grad():
        ref = S.render(H, W, f, chunk=chunk, c2w=c2w[:3, :4], **kw)
    got = S.render_sharded(H, W, f, c2w, chunk, kw)
    for a, b in zip(got, ref[:4]):
        assert torch.equal(a, b)


def test_convert_pose_flips_y_and_z():
    import spin_nerf_amd as S
    c = np.arange(16, dtype=np.float64).reshape(4, 4)
    out = S.convert_pose(c)
    assert np.array_equal(out[:, 0], c[:, 0]) and np.array_equal(out[:, 1], -c[:, 1])
    assert np.array_equal(out[:, 2], -c[:, 2]) and np.array_equal(out[:, 3], c[:, 3])


@pytest.mark.gpu
def test_projection_and_test_ray_helpers():
    """render_path_projection (run_nerf.py:310-339) returns render()'s own z_vals / weights; render_test_ray
    (:350-377) agrees with the CPU oracle's network + compositing on evenly spaced depths."""
    import spin_nerf_amd as S
    from oracle import nerf_oracle as O
    from test_gpu_render import build
    from helpers import render_case_nets
    g = load("render_c2w_fine_vd")
    _, _, kw = build(S, g)
    H, W, f, chunk = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["chunk"])
    c2w = T(g["c2w"]).cuda()
    poses = torch.stack([c2w, c2w], 0)
    z, w, c2ws, K = S.render_path_projection(poses, (H, W, f), chunk, kw)
    with torch.no_grad():
        ex = S.render(H, W, f, chunk=chunk, c2w=c2w[:3, :4], retraw=True, **kw)[4]
    assert np.array_equal(z[1], ex["z_vals"].cpu().numpy()) and np.array_equal(w[0], ex["weights"].cpu().numpy())
    full = np.concatenate([c2w[:3, :4].cpu().numpy(), [[0, 0, 0, 1]]], 0)
    assert np.allclose(c2ws[0], S.convert_pose(full)) and K[0, 2] == W / 2

    ro, rd = S.get_rays(H, W, f, c2w[:3, :4])
    ro, rd = ro[::3, ::4].reshape(-1, 3), rd[::3, ::4].reshape(-1, 3)
    near, far = float(g["near"]), float(g["far"])
    with torch.no_grad():
        rgb, sigma, zv, depth = S.render_test_ray(ro, rd, (H, W, f), False, near, far, True, 48, kw["network_fn"],
                                                  kw["network_query_fn"])
    sd_c, _ = render_case_nets(g)
    ro_c, rd_c = ro.cpu(), rd.cpu()
    vd = rd_c / rd_c.norm(dim=-1, keepdim=True)
    zr = near * (1 - torch.linspace(0, 1, 48)) + far * torch.linspace(0, 1, 48)
    zr = zr.expand(ro_c.shape[0], 48)
    raw = O.run_network(sd_c, ro_c[:, None] + rd_c[:, None] * zr[..., None], vd)
    np.testing.assert_allclose(zv.cpu().numpy(), zr.numpy(), atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), torch.sigmoid(raw[..., :3]).numpy(), atol=2e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), torch.relu(raw[..., 3]).numpy(), atol=2e-4, rtol=2e-5)
    np.testing.assert_allclose(depth.cpu().numpy(), O.raw2outputs(raw, zr, rd_c)[4].numpy(), rtol=2e-4, atol=1e-5)
