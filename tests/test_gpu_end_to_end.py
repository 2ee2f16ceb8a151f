"""GPU: the pieces around the hot path used together the way the reference's train() uses them — LLFF folder ->
poses / ray table -> three-render iterations on the fused kernels -> render_path dump -> checkpoint round trip."""
import argparse
import contextlib
import importlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(tmp_path, N=7, H=24, W=32):
    """A folder in LLFF layout: cameras on an arc looking at a shaded sphere; lama_images, label, depth."""
    import spin_nerf_amd as S
    P = importlib.import_module("spin-nerf_amd.poses")
    from test_gpu_train import sphere_scene
    base = tmp_path / "scene"
    for d in ("images_4/lama_images", "images_4/label", "images_4/depth"):
        (base / d).mkdir(parents=True)
    rows, focal = [], 30.0 * 4
    for k in range(N):
        a = -0.5 + k / (N - 1.0)
        pos = np.array([4 * np.sin(a), 0.3, 4 * np.cos(a)])
        z = pos / np.linalg.norm(pos)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        m = np.stack([-y, x, z, pos, np.array([H * 4, W * 4, focal])], 1)    # LLFF storage: [-u, r, -t | pos | hwf]
        rows.append(np.concatenate([m.reshape(-1), [2.0, 6.0]]))
        c2w = torch.tensor(np.stack([x, y, z, pos], 1), dtype=torch.float32)
        ro, rd = S.get_rays(H, W, focal / 4, c2w.cuda())
        img = sphere_scene(ro, rd, white=False).cpu().numpy()
        S.write_png(str(base / "images_4/lama_images" / f"{k:03d}.png"), S.to8b(img))
        msk = np.zeros((H, W), np.uint8); msk[8:14, 10:18] = 255
        S.write_png(str(base / "images_4/label" / f"{k:03d}.png"), msk)
        S.write_png(str(base / "images_4/depth" / f"{k:03d}.png"), np.full((H, W), 64, np.uint8))
    np.save(base / "poses_bounds.npy", np.stack(rows, 0))
    return str(base), H, W


def test_folder_to_checkpoint(tmp_path):
    import spin_nerf_amd as S
    P = importlib.import_module("spin-nerf_amd.poses")
    train = importlib.import_module("spin-nerf_amd.train")
    base, H, W = _scene(tmp_path)
    images, poses, bds, render_poses, i_test, masks, depths, idx = P.load_llff_data(base, factor=4, recenter=True,
                                                                                    bd_factor=.75, spherify=False)
    assert images.shape == (7, H, W, 3) and poses.shape == (7, 3, 5) and render_poses.shape == (120, 3, 5)
    hwf = poses[0, :3, -1]
    assert (int(hwf[0]), int(hwf[1])) == (H, W)
    focal = float(hwf[2])
    i_train = [i for i in range(7) if i != i_test]
    table = P.build_ray_table(poses, images, masks, H, W, focal, i_train)            # [n*H*W, 3, 4]
    table_inp = P.build_ray_table(poses, images, depths, H, W, focal, i_train)
    clf = table[table[:, 0, 3] == 0]                                                   # unmasked pixels only

    (tmp_path / "run").mkdir()
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=32, N_samples=32,
                              alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              netchunk=65536, lrate=5e-4, basedir=str(tmp_path), expname="run", ft_path=None, no_reload=True,
                              perturb=1.0, white_bkgd=False, raw_noise_std=1.0, dataset_type="llff", no_ndc=True,
                              lindisp=False, sigma_loss=False, no_coarse=False, precision="bf16")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, *_ = S.create_nerf(args, device=torch.device("cuda"))
    near, far = float(bds.min() * .9), float(bds.max() * 1.)
    kw_train.update(near=near, far=far); kw_test.update(near=near, far=far)
    tr = train.RenderTrainer(kw_train, lrate=5e-4)
    g = torch.Generator().manual_seed(0)

    def batch(tab, n=256):
        sel = torch.randint(0, tab.shape[0], (n,), generator=g)
        t = torch.from_numpy(tab[sel.numpy()]).cuda()
        return torch.stack([t[:, 0, :3], t[:, 1, :3]], 0), t[:, 2, :3], t[:, 0, 3]

    psnrs = []
    for it in range(60):
        r_clf, t_clf, _ = batch(clf)
        r_all, t_all, _ = batch(table)
        r_inp, _, d_inp = batch(table_inp)
        loss, psnr = tr.spin_iteration(H, W, focal, r_clf, t_clf, r_all, t_all, r_inp, d_inp, batched=(it % 2 == 0))
        assert np.isfinite(float(loss))
        psnrs.append(float(psnr))
    assert np.mean(psnrs[-10:]) > np.mean(psnrs[:10]) + 1.0, (psnrs[:10], psnrs[-10:])

    out = tmp_path / "renders"; out.mkdir()
    rp = torch.from_numpy(render_poses[:2]).cuda()
    rgbs, disps, _ = S.render_path(rp, (H, W, focal), 4096, kw_test, savedir=str(out))
    assert rgbs.shape == (2, H, W, 3) and np.isfinite(rgbs).all() and os.path.exists(out / "rgb" / "000001.png")

    ck = str(tmp_path / "run" / "000060.tar")
    tr.save_checkpoint(ck)
    args.no_reload = False
    with contextlib.redirect_stdout(io.StringIO()):
        kw2, _, start, _, _ = S.create_nerf(args, device=torch.device("cuda"))
    assert start == 60
    assert torch.equal(kw2["network_fine"].flat.detach(), kw_train["network_fine"].flat.detach())
