"""CPU: pose half of the LLFF ingestion against the reference's own load_llff_data run on
synthetic arrays (tests/golden/make_golden_poses.py), and the ray table of train() (run_nerf.py:1228-1247)."""
import importlib

import numpy as np
import pytest

from helpers import load

P = importlib.import_module("spin-nerf_amd.poses")


@pytest.mark.parametrize("name", ["poses_default", "poses_spherify", "poses_norecenter"])
def test_llff_poses_match_reference(name):
    g = load(name)
    bd = None if float(g["bd_factor"]) < 0 else float(g["bd_factor"])
    poses, bds, render_poses, i_test = P.llff_poses(g["poses_in"].copy(), g["bds_in"].copy(), recenter=bool(g["recenter"]),
                                                    bd_factor=bd, spherify=bool(g["spherify"]),
                                                    spherify_hack=bool(g["hack"]))
    assert poses.dtype == np.float32 and poses.shape == g["poses"].shape
    np.testing.assert_allclose(poses, g["poses"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(bds, g["bds"], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(render_poses, g["render_poses"], atol=2e-5, rtol=1e-5)
    assert i_test == int(g["i_test"])


def test_pose_helpers_are_consistent():
    rs = np.random.RandomState(0)
    m = P.viewmatrix(rs.randn(3), np.array([0, 1.0, 0]), rs.randn(3))
    assert np.allclose(m[:, :3].T @ m[:, :3], np.eye(3), atol=1e-12)          # orthonormal frame
    g = load("poses_default")
    poses = g["poses"]
    avg = P.poses_avg(P.recenter_poses(poses))
    assert np.allclose(avg[:3, :3], np.eye(3), atol=1e-4) and np.allclose(avg[:3, 3], 0, atol=1e-5)   # idempotent
    assert np.allclose(P.ptstocam(poses[0, :3, 3][None], poses[0]), 0, atol=1e-6)


def test_ray_table_layout():
    """[n_train*H*W, 3, 4]: (ro | rd | rgb) x (xyz, label), training views only, get_rays_np's pixel order."""
    from oracle import nerf_oracle as O
    import torch
    rs = np.random.RandomState(1)
    g = load("poses_default")
    poses = g["poses"]
    N, H, W, focal = poses.shape[0], 6, 8, 9.0
    images = rs.rand(N, H, W, 3).astype(np.float32)
    labels = rs.rand(N, H, W).astype(np.float32)
    i_train = [0, 2, 5]
    t = P.build_ray_table(poses, images, labels, H, W, focal, i_train)
    assert t.shape == (3 * H * W, 3, 4) and t.dtype == np.float32
    for n, i in enumerate(i_train):
        ro, rd = O.get_rays(H, W, focal, torch.from_numpy(poses[i, :3, :4]))
        blk = t[n * H * W:(n + 1) * H * W].reshape(H, W, 3, 4)
        np.testing.assert_allclose(blk[..., 0, :3], ro.numpy(), atol=1e-6)
        np.testing.assert_allclose(blk[..., 1, :3], rd.numpy(), atol=1e-6)
        assert np.array_equal(blk[..., 2, :3], images[i])
        for r in range(3):
            assert np.array_equal(blk[..., r, 3], labels[i])


def test_dilate_and_resize_against_independent_implementations():
    from scipy import ndimage
    rs = np.random.RandomState(2)
    m = (rs.rand(23, 31) > 0.9).astype(np.float64) * rs.rand(23, 31)
    ref = m
    for _ in range(5):
        ref = ndimage.grey_dilation(ref, size=(5, 5), mode="constant", cval=-np.inf)
    assert np.array_equal(P.dilate(m, 5, 5), ref)
    a = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(P.resize_nearest(a, 6, 8), np.repeat(np.repeat(a, 2, 0), 2, 1))
    assert np.array_equal(P.resize_nearest(a, 3, 2), a[:, [0, 2]])


def test_load_llff_folder_round_trip(tmp_path):
    """A synthetic LLFF folder written with the package's own PNG encoder: images, masks (one missing), depths."""
    S = importlib.import_module("spin-nerf_amd")
    g = load("poses_default")
    N, H, W = 7, 6, 8
    rs = np.random.RandomState(4)
    base = tmp_path / "scene"
    for d in ("images_2/lama_images", "images_2/label", "images_2/depth"):
        (base / d).mkdir(parents=True)
    arr = np.concatenate([np.moveaxis(g["poses_in"], -1, 0).reshape(N, 15), np.moveaxis(g["bds_in"], -1, 0)], 1)
    np.save(base / "poses_bounds.npy", arr)
    imgs = rs.randint(0, 256, size=(N, H, W, 3)).astype(np.uint8)
    msk = (rs.rand(N, H, W) > 0.8).astype(np.uint8) * 255
    dep = rs.randint(0, 256, size=(N, H, W)).astype(np.uint8)
    for i in range(N):
        S.write_png(str(base / "images_2/lama_images" / f"{i:03d}.png"), imgs[i])
        if i != 3:
            S.write_png(str(base / "images_2/label" / f"{i:03d}.png"), msk[i])
        S.write_png(str(base / "images_2/depth" / f"{i:03d}.png"), dep[i])
    images, poses, bds, render_poses, i_test, masks, depths, idx = P.load_llff_data(str(base), factor=2)
    assert images.shape == (N, H, W, 3) and images.dtype == np.float32
    np.testing.assert_allclose(images, imgs / 255.0, atol=1e-7)
    np.testing.assert_allclose(depths, dep / 255.0, atol=1e-7)
    assert idx == [0, 1, 2, 4, 5, 6] and np.all(masks[3] == -1)
    for i in idx:
        np.testing.assert_allclose(masks[i], P.dilate(msk[i] / 255.0, 5, 5), atol=1e-7)
    # hwf column: image size, focal / factor; the rest equals the pose-only path with that column set
    pin = g["poses_in"].copy()
    pin[:2, 4, :] = np.array([H, W]).reshape(2, 1)
    pin[2, 4, :] = pin[2, 4, :] / 2
    p2, b2, r2, t2 = P.llff_poses(pin, g["bds_in"].copy())
    assert np.array_equal(poses, p2) and np.array_equal(bds, b2) and np.array_equal(render_poses, r2) and i_test == t2
    with pytest.raises(FileNotFoundError):
        P.load_llff_data(str(base), factor=8)


def test_rays_by_coordinate_pick_the_same_rays_as_the_full_grid():
    g = load("poses_default")
    c2w = g["poses"][2, :3, :4]
    H, W, focal = 6, 8, 9.0
    ro, rd = P.get_rays_np(H, W, focal, c2w)
    coords = np.array([[0, 0], [3, 2], [7, 5]], dtype=np.float32)      # (x, y)
    ro_c, rd_c = P.get_rays_by_coord_np(H, W, focal, c2w, coords)
    for n, (x, y) in enumerate(coords.astype(int)):
        np.testing.assert_allclose(rd_c[n], rd[y, x], atol=1e-6)
        np.testing.assert_allclose(ro_c[n], ro[y, x], atol=0)
