"""Camera-pose side of the LLFF ingestion: what `load_llff_data` (DS_NeRF/load_llff.py:315-433) does
to `poses_bounds.npy` once the files are read — axis reorder, bound rescale, recentering, the "spherify hack", the
spiral render path, the hold-out view — and the ray table `train()` builds from poses + images
(run_nerf.py:1225-1262).  Pure numpy, host side; image / mask / depth file reading (cv2, imageio) is not built.

Pinned by fixtures produced by the reference's own `load_llff_data` with only its file reader replaced by synthetic
arrays (tests/golden/make_golden_poses.py)."""
import numpy as np


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """load_llff.py:197-203: camera frame with z = viewing axis, columns [x y z pos]."""
    z = normalize(z)
    x = normalize(np.cross(up, z))
    y = normalize(np.cross(z, x))
    return np.stack([x, y, z, pos], 1)


def ptstocam(pts, c2w):
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """load_llff.py:211-219: mean position, summed viewing / up axes, hwf of the first pose -> [3,5]."""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    z = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(z, up, center), hwf], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """load_llff.py:222-232."""
    out = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return out


def _to44(p):
    return np.concatenate([p, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])], 1)


def recenter_poses(poses):
    """load_llff.py:235-247: express all poses in the average pose's frame."""
    out = poses + 0
    c2w = np.concatenate([poses_avg(poses)[:3, :4], np.reshape([0, 0, 0, 1.], [1, 4])], -2)
    out[:, :3, :4] = (np.linalg.inv(c2w) @ _to44(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """load_llff.py:252-312.  Like the reference this scales `bds` IN PLACE.  Returns
    (poses_reset, new_poses, bds, scale, inverse of the reset transform)."""
    rays_d, rays_o = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - rays_d * np.transpose(rays_d, [0, 2, 1])
    b = -A @ rays_o
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    up = (poses[:, :3, 3] - center).mean(0)
    v0 = normalize(up)
    v1 = normalize(np.cross([.1, .2, .3], v0))
    v2 = normalize(np.cross(v0, v1))
    c2w = np.stack([v1, v2, v0, center], 1)
    inv = np.linalg.inv(_to44(c2w[None]))
    reset = inv @ _to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    ring = []
    for th in np.linspace(0., 2. * np.pi, 120):
        o = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        z = normalize(o)
        x = normalize(np.cross(z, np.array([0, 0, -1.])))
        y = normalize(np.cross(z, x))
        ring.append(np.stack([x, y, z, o], 1))
    ring = np.stack(ring, 0)
    hwf = poses[0, :3, -1:]
    ring = np.concatenate([ring, np.broadcast_to(hwf, ring[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, ring, bds, sc, inv


def llff_poses(poses, bds, recenter=True, bd_factor=.75, spherify=False, path_zflat=False, spherify_hack=True):
    """The pose half of load_llff_data (load_llff.py:326-420).  `poses` [3,5,N] and `bds` [2,N] as _load_data returns
    them (hwf column already set).  Returns (poses [N,3,5] float32, bds [N,2] float32, render_poses [M,3,5] float32,
    i_test).  As in the reference the spiral path is computed unconditionally and is what `render_poses` ends up
    being, whatever the spherify flags produced before it; `path_zflat` halves the view count (the reference's float
    count does not survive current numpy, an int is used here)."""
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)   # [-u, r, -t] -> [r, u, -t]
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, _, bds, _, _ = spherify_poses(poses, bds)
    elif spherify_hack:
        _, _, bds_s, sc_s, _ = spherify_poses(poses, bds)     # only its rounding of bds survives (:354-357)
        bds = bds_s / sc_s
    c2w = poses_avg(poses)
    up = normalize(poses[:, :3, 1].sum(0))
    close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
    dt = .75
    focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)
    zdelta = close_depth * .2
    rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    n_views, n_rots = 120, 2
    if path_zflat:
        zloc = -close_depth * .1
        c2w[:3, 3] = c2w[:3, 3] + zloc * c2w[:3, 2]
        rads[2] = 0.
        n_rots, n_views = 1, 60
    render_poses = np.array(render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    return poses.astype(np.float32), bds, render_poses, i_test


def get_rays_np(H, W, focal, c2w):
    """run_nerf_helpers.py:263-272."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def get_rays_by_coord_np(H, W, focal, c2w, coords):
    """run_nerf_helpers.py:275-280: rays through the given (x, y) pixel coordinates [n, 2] (COLMAP key points)."""
    x, y = (coords[:, 0] - W * .5) / focal, -(coords[:, 1] - H * .5) / focal
    dirs = np.stack([x, y, -np.ones_like(x)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    return np.broadcast_to(c2w[:3, -1], np.shape(rays_d)), rays_d


def build_ray_table(poses, images, labels, H, W, focal, i_train):
    """run_nerf.py:1228-1247: rows [ro | rd | rgb] x (xyz, label) for every pixel of the training views,
    shape [n_train*H*W, 3, 4] float32 (label = mask value or inpainted depth of the pixel, repeated)."""
    rays = np.stack([np.stack(get_rays_np(H, W, focal, p), 0) for p in poses[:, :3, :4]], 0)   # [N, 2, H, W, 3]
    lab = np.repeat(np.expand_dims(labels, -1)[:, None], 3, axis=1)                            # [N, 3, H, W, 1]
    t = np.concatenate([np.concatenate([rays, images[:, None]], 1), lab], -1)                  # [N, 3, H, W, 4]
    t = np.transpose(t, [0, 2, 3, 1, 4])
    t = np.stack([t[i] for i in i_train], 0)
    return np.reshape(t, [-1, 3, 4]).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# file-reading half (load_llff.py:68-190).  Parity with the reference is NOT pinned here: it reads through imageio
# and dilates / resizes with cv2, neither of which exists in the build container; PNG decoding is lossless (PIL
# gives the same bytes), the 5x5 dilation and the nearest-neighbour resize are checked against scipy / by hand.
# ----------------------------------------------------------------------------------------------------------------
_IMG_EXT = ('JPG', 'jpg', 'jpeg', 'png')


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def dilate(msk, ksize=5, iterations=5):
    """cv2.dilate(msk, ones((k,k)), iterations=n): n passes of a k x k maximum filter; pixels outside the image do
    not take part (cv2's default border for dilation)."""
    r = ksize // 2
    out = np.asarray(msk, dtype=np.float64)
    H, W = out.shape
    for _ in range(iterations):
        pad = np.full((H + 2 * r, W + 2 * r), -np.inf)
        pad[r:r + H, r:r + W] = out
        out = np.max(np.stack([pad[i:i + H, j:j + W] for i in range(ksize) for j in range(ksize)], 0), 0)
    return out


def resize_nearest(a, H, W):
    """cv2.resize(..., interpolation=INTER_NEAREST): source index = floor(dst * src / dst_size)."""
    ys = np.minimum((np.arange(H) * (a.shape[0] / H)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum((np.arange(W) * (a.shape[1] / W)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def load_llff_folder(basedir, factor=None, prepare=False, lpips=False, load_imgs=True):
    """_load_data (load_llff.py:68-190) for a folder whose down-scaled image directories already exist
    (`images_<factor>`; the reference shells out to ImageMagick to create them): poses_bounds.npy -> poses [3,5,N]
    with the hwf column set from the image size and the factor, bds [2,N]; images from `images<sfx>/lama_images`
    (or `images<sfx>` with prepare=True) / 255; masks from `images<sfx>/label/<stem>.png`, scaled to [0,1],
    dilated 5x5 five times (-1 everywhere when the file is missing), sign-flipped for all but the fifth-last view
    when `lpips` (load_llff.py:162-163), divided by their global maximum; depths from `images<sfx>/depth` / 255."""
    import os
    arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])
    sfx = '' if factor is None else '_{}'.format(factor)
    factor = 1 if factor is None else factor
    imgdir = os.path.join(basedir, 'images' + sfx) if prepare else os.path.join(basedir, 'images' + sfx, 'lama_images')
    mskdir = os.path.join(basedir, 'images' + sfx, 'label')
    depthdir = os.path.join(basedir, 'images' + sfx, 'depth')
    if not os.path.exists(imgdir):
        raise FileNotFoundError(f"{imgdir} does not exist (down-scaled image folders are not generated here)")
    names = [f for f in sorted(os.listdir(imgdir)) if f.endswith(_IMG_EXT)]
    imgfiles = [os.path.join(imgdir, f) for f in names]
    mskfiles = [os.path.join(mskdir, f.split('.')[0] + '.png') for f in names if 'cutout' not in f and 'pseudo' not in f]
    try:
        depthfiles = [os.path.join(depthdir, f.split('.')[0] + '.png') for f in sorted(os.listdir(depthdir))
                      if f.endswith(_IMG_EXT)]
    except OSError:
        depthfiles = mskfiles
    if poses.shape[-1] > len(imgfiles):
        poses = poses[:, :, :len(imgfiles)]
    if poses.shape[-1] != len(imgfiles):
        raise ValueError('Mismatch between imgs {} and poses {}'.format(len(imgfiles), poses.shape[-1]))
    sh = _imread(imgfiles[0]).shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([_imread(f)[..., :3] / 255. for f in imgfiles], -1)
    H, W = imgs.shape[0], imgs.shape[1]

    def plane(f, scale_by_max):
        m = _imread(f)
        m = m / (m.max() if scale_by_max else 255.)
        if m.ndim > 2:
            m = m[:, :, 0]
        if m.shape != (H, W):
            m = resize_nearest(m, H, W)
        return m

    masks, mask_indices = [], []
    for i, f in enumerate(mskfiles):
        try:
            m = dilate(plane(f, True), 5, 5)
            mask_indices.append(i)
            if (i != len(mskfiles) - 5) and (not prepare) and lpips:
                m = m * (-1)
            masks.append(m)
        except (OSError, ValueError):
            masks.append(-np.ones((H, W)))
    depths = []
    for f in depthfiles:
        try:
            depths.append(plane(f, False))
        except (OSError, ValueError):
            depths.append(-np.ones((H, W)))
    masks = np.stack(masks, -1)
    masks = masks / np.max(masks)
    return poses, bds, imgs, masks, np.stack(depths, -1), mask_indices


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False,
                   spherify_hack=True, prepare=False, lpips=False):
    """load_llff_data (load_llff.py:315-433): (images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses, i_test,
    masks [N,H,W], inpainted_depths [N,H,W], mask_indices), all float32."""
    poses, bds, imgs, masks, depths, mask_indices = load_llff_folder(basedir, factor=factor, prepare=prepare, lpips=lpips)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    masks = np.moveaxis(masks, -1, 0).astype(np.float32)
    depths = np.moveaxis(depths, -1, 0).astype(np.float32)
    poses, bds, render_poses, i_test = llff_poses(poses, bds, recenter, bd_factor, spherify, path_zflat, spherify_hack)
    return images, poses, bds, render_poses, i_test, masks, depths, mask_indices
