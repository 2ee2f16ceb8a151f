"""Camera-pose side of the LLFF ingestion: what `load_llff_data` (DS_NeRF/load_llff.py:315-433) does
to `poses_bounds.npy` once the files are read — axis reorder, bound rescale, recentering, the "spherify hack", the
spiral render path, the hold-out view — and the ray table `train()` builds from poses + images
(run_nerf.py:1225-1262) — plus, further down, the file-reading half (`load_llff_folder`, `minify`), the COLMAP depth
loader (`load_colmap_depth`, load_llff.py:448-501) and the four shuffled ray feeds of train() (run_nerf.py:1264-1417).
Pure numpy / PIL, host side.

Pinned by fixtures produced by the reference's own `load_llff_data` with only its file reader replaced by synthetic
arrays (tests/golden/make_golden_poses.py) and by its `load_colmap_depth` on a synthetic COLMAP model
(tests/golden/make_golden_colmap.py).  The image / mask / depth file reader itself is parity-UNPINNED: the reference reads
through imageio and cv2, neither of which is on this image (see the note above `load_llff_folder`)."""
import os

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


# ----------------------------------------------------------------------------------------------
# COLMAP sparse depth (load_llff.py:436-501) and pre-scaled image folders (load_llff.py:14-66)
# ----------------------------------------------------------------------------------------------
def read_images_binary(path):
    """COLMAP images.bin -> {image_id: dict(qvec, tvec, camera_id, name, xys [n,2], point3D_ids [n])}
    (format of colmapUtils/read_write_model.py:225-257: uint64 count; per image int32 id, 4+3 doubles, int32 camera, a
    zero-terminated name, uint64 n, n x (double x, double y, int64 point3D id))"""
    import struct
    out = {}
    with open(path, "rb") as f:
        n_img = struct.unpack("<Q", f.read(8))[0]
        for _ in range(n_img):
            vals = struct.unpack("<idddddddi", f.read(64))
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00":
                    break
                name += c
            n2d = struct.unpack("<Q", f.read(8))[0]
            rec = np.frombuffer(f.read(24 * n2d), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            out[vals[0]] = dict(qvec=np.array(vals[1:5]), tvec=np.array(vals[5:8]), camera_id=vals[8],
                                name=name.decode("utf-8"), xys=np.column_stack([rec["x"], rec["y"]]).reshape(-1, 2),
                                point3D_ids=rec["id"].astype(np.int64))
    return out


def read_points3d_binary(path):
    """COLMAP points3D.bin -> {point3D_id: dict(xyz, rgb, error)} (read_write_model.py:336-363; tracks are skipped)"""
    import struct
    out = {}
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        for _ in range(n):
            vals = struct.unpack("<QdddBBBd", f.read(43))
            track = struct.unpack("<Q", f.read(8))[0]
            f.seek(8 * track, 1)
            out[vals[0]] = dict(xyz=np.array(vals[1:4]), rgb=np.array(vals[4:7]), error=float(vals[7]))
    return out


def qvec2rotmat(q):
    """read_write_model.py: Image.qvec2rotmat (w, x, y, z)"""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def colmap_poses(images):
    """camera-to-world matrices in image-id order of the dict (load_llff.py:436-445)"""
    poses = []
    for i in images:
        w2c = np.concatenate([np.concatenate([qvec2rotmat(images[i]["qvec"]), images[i]["tvec"].reshape(3, 1)], 1),
                              np.array([[0, 0, 0, 1.]])], 0)
        poses.append(np.linalg.inv(w2c))
    return np.array(poses)


def load_colmap_depth(basedir, factor=8, bd_factor=.75, prepare=False, bds_raw=None, save=True):
    """Per-image sparse depths from the COLMAP reconstruction (load_llff.py:448-501): for every registered 2D point
    with a 3D point, depth = camera z-axis . (X - camera position) * sc, kept when inside the image's [near, far] bounds,
    with weight 2 exp(-(err / mean err)^2) and pixel coordinates / factor.  Returns (and saves to colmap_depth.npy like the
    reference) a list of dicts {"depth", "coord", "weight"}; ``bds_raw`` [N,2] overrides the bounds read from
    poses_bounds.npy (the reference takes them from _load_data)."""
    images = read_images_binary(os.path.join(basedir, "sparse", "0", "images.bin"))
    points = read_points3d_binary(os.path.join(basedir, "sparse", "0", "points3D.bin"))
    err_mean = np.mean(np.array([p["error"] for p in points.values()]))
    print("Mean Projection Error:", err_mean)
    poses = colmap_poses(images)
    if bds_raw is None:
        arr = np.load(os.path.join(basedir, "poses_bounds.npy"))
        bds_raw = arr[:, -2:]
    bds_raw = np.asarray(bds_raw).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds_raw.min() * bd_factor)
    print('near/far:', np.ndarray.min(bds_raw) * .9 * sc, np.ndarray.max(bds_raw) * 1. * sc)
    data_list = []
    for id_im in range(1, len(images) + 1):
        depth_list, coord_list, weight_list = [], [], []
        im = images[id_im]
        for i in range(len(im["xys"])):
            id_3d = im["point3D_ids"][i]
            if id_3d == -1:
                continue
            p3 = points[id_3d]
            depth = (poses[id_im - 1, :3, 2].T @ (p3["xyz"] - poses[id_im - 1, :3, 3])) * sc
            if depth < bds_raw[id_im - 1, 0] * sc or depth > bds_raw[id_im - 1, 1] * sc:
                continue
            depth_list.append(depth)
            coord_list.append(im["xys"][i] / factor)
            weight_list.append(2 * np.exp(-(p3["error"] / err_mean) ** 2))
        if len(depth_list) > 0:
            data_list.append({"depth": np.array(depth_list), "coord": np.array(coord_list), "weight": np.array(weight_list)})
    if save:
        np.save(os.path.join(basedir, "colmap_depth.npy"), np.array(data_list, dtype=object), allow_pickle=True)
    return data_list


def minify(basedir, factors=(), resolutions=()):
    """Pre-scaled copies of <basedir>/images as images_<f> / images_<W>x<H> PNG folders (load_llff.py:14-66).  The reference
    shells out to ImageMagick's `mogrify -resize`; neither it nor cv2 is on this image, so the resampling is PIL's (box
    reduction for integer factors, LANCZOS otherwise) — parity unpinned, the folder / naming contract is the reference's."""
    from PIL import Image
    imgdir = os.path.join(basedir, "images")
    names = [f for f in sorted(os.listdir(imgdir)) if f.split(".")[-1] in ("JPG", "jpg", "png", "jpeg", "PNG")]
    for r in list(factors) + list(resolutions):
        name = "images_{}".format(r) if isinstance(r, int) else "images_{}x{}".format(r[1], r[0])
        out = os.path.join(basedir, name)
        if os.path.exists(out):
            continue
        print("Minifying", r, basedir)
        os.makedirs(out)
        for f in names:
            im = Image.open(os.path.join(imgdir, f))
            if isinstance(r, int):
                size = (int(round(im.size[0] / r)), int(round(im.size[1] / r)))
                small = im.reduce(r) if (im.size[0] % r == 0 and im.size[1] % r == 0) else im.resize(size, Image.LANCZOS)
            else:
                small = im.resize((r[1], r[0]), Image.LANCZOS)
            small.save(os.path.join(out, f.rsplit(".", 1)[0] + ".png"))


# ----------------------------------------------------------------------------------------------
# train()'s ray tables and its four shuffled feeds (run_nerf.py:1228-1348, 1362-1417)
# ----------------------------------------------------------------------------------------------
def build_depth_rays(depth_gts, masks, poses, H, W, focal, i_train, prepare=False):
    """rays_depth [n, 4, 3] = (ray origin, ray direction, depth x3, weight x3) of the COLMAP points of the training views
    that fall outside the object mask (run_nerf.py:1264-1300), and max_depth = max weight-column value as the
    reference computes it (:1302-1303 takes column 3)."""
    out = []
    for i in i_train:
        g = depth_gts[i]
        coord, weight, depth = g["coord"], g["weight"], g["depth"]
        if not prepare:
            keep = [k for k in range(len(coord))
                    if masks[i][min(int(coord[k][1]), masks[i].shape[0] - 1)][min(int(coord[k][0]), masks[i].shape[1] - 1)] == 0]
            coord, weight, depth = coord[keep], weight[keep], depth[keep]
        rays = np.stack(get_rays_by_coord_np(H, W, focal, poses[i, :3, :4], coord), axis=0)      # 2 x n x 3
        rays = np.transpose(rays, [1, 0, 2])
        dv = np.repeat(depth[:, None, None], 3, axis=2)
        wv = np.repeat(weight[:, None, None], 3, axis=2)
        out.append(np.concatenate([rays, dv, wv], axis=1))
    rays_depth = np.concatenate(out, axis=0).astype(np.float32)
    return rays_depth, float(np.max(rays_depth[:, 3, 0]))


def split_ray_tables(rays_rgb, rays_inp, prepare=False, train_gt=False):
    """The three tables train() samples from (run_nerf.py:1306-1322): rows are [3, 4] = (o | d | rgb) x (xyz, label).
    -> (rays_rgb: pixels of label 1 unless --prepare, rays_rgb_clf: label 0 unless --train_gt / --prepare,
        rays_inp: the inpainted-depth table at the pixels whose label is not 0)"""
    lab = rays_rgb[:, :, 3]
    clf = rays_rgb.reshape(-1, 3, 4) if (train_gt or prepare) else rays_rgb[lab == 0].reshape(-1, 3, 4)
    inp = rays_inp[lab != 0].reshape(-1, 3, 4)
    rgb = rays_rgb if prepare else rays_rgb[lab == 1].reshape(-1, 3, 4)
    return rgb, clf, inp


class RayFeeds:
    """The shuffled mini-batch feeds of train() (run_nerf.py:1336-1348, 1362-1417): one per table, each a pass over a
    fresh random permutation of its rows in batches of N_rand (the last batch of a pass is short, like DataLoader's),
    restarted when exhausted.  Tables live on the device; a batch is one index_select."""

    def __init__(self, rays_rgb, rays_inp, rays_rgb_clf, rays_depth=None, N_rand=1024, device="cuda", seed=None):
        import torch
        self.torch = torch
        self.N = N_rand
        self.gen = torch.Generator(device="cpu")
        if seed is not None:
            self.gen.manual_seed(seed)
        self.tab = {k: torch.as_tensor(v, dtype=torch.float32).to(device) for k, v in
                    (("rgb", rays_rgb), ("inp", rays_inp), ("clf", rays_rgb_clf), ("depth", rays_depth)) if v is not None}
        self.perm, self.pos = {}, {}

    def _next(self, key):
        t = self.tab[key]
        if key not in self.perm or self.pos[key] >= t.shape[0]:
            self.perm[key] = self.torch.randperm(t.shape[0], generator=self.gen).to(t.device)
            self.pos[key] = 0
        idx = self.perm[key][self.pos[key]:self.pos[key] + self.N]
        self.pos[key] += self.N
        return self.torch.transpose(t.index_select(0, idx), 0, 1)

    def next_batch(self):
        """dict with the reference's names: batch_rays [2,B,3], target_s [B,3], label_s; batch_inp, target_inp, depth_inp;
        batch_rays_clf, target_clf, label_s_clf; and with COLMAP depth batch_rays_depth, target_depth, ray_weights"""
        out = {}
        b = self._next("rgb")
        out["batch_rays"], out["target_s"], out["label_s"] = b[:2, :, :-1].contiguous(), b[2, :, :3], b[2, :, 3]
        b = self._next("inp")
        out["batch_inp"], out["target_inp"], out["depth_inp"] = b[:2, :, :-1].contiguous(), b[2, :, :3], b[2, :, 3]
        b = self._next("clf")
        out["batch_rays_clf"], out["target_clf"], out["label_s_clf"] = b[:2, :, :-1].contiguous(), b[2, :, :3], b[2, :, 3]
        if "depth" in self.tab:
            b = self._next("depth")
            out["batch_rays_depth"], out["target_depth"], out["ray_weights"] = b[:2].contiguous(), b[2, :, 0], b[3, :, 0]
        return out
