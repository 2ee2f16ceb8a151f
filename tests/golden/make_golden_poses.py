"""Golden vectors for the pose half of the LLFF ingestion (DS_NeRF/load_llff.py:193-433).  Build container only:

    python tests/golden/make_golden_poses.py

The reference's `load_llff_data` runs unmodified; only its file reader `_load_data` is replaced by a function that
returns synthetic arrays of the shapes it documents (there is no dataset, cv2 or imageio in the container)."""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
from make_golden import _stub, npz

for n in ["cv2", "imageio", "colmapUtils", "colmapUtils.read_write_model", "colmapUtils.read_write_dense"]:
    _stub(n)
sys.path.insert(0, "/root/reference/DS_NeRF")
import load_llff as L

rs = np.random.RandomState(3)
N, H, W = 7, 6, 8
# cameras on an arc around the origin, LLFF storage convention [-u, r, -t | pos | hwf]
cols = []
for k in range(N):
    a = -0.5 + k / (N - 1.0)
    pos = np.array([2.5 * np.sin(a), 0.2 * rs.randn(), 2.5 * np.cos(a)]) + 0.05 * rs.randn(3)
    z = pos / np.linalg.norm(pos)
    x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    r, u, t = x, y, -z                                  # right, up, -viewing
    cols.append(np.stack([-u, r, -t, pos, np.array([H, W, 9.0])], 1))   # [3,5]
poses = np.stack(cols, -1)                              # [3,5,N]
bds = np.stack([1.2 + 0.3 * rs.rand(N), 5.0 + rs.rand(N)], 0)   # [2,N]
imgs = rs.rand(H, W, 3, N).astype(np.float32)
masks = (rs.rand(H, W, 1, N) > 0.5).astype(np.float32)
depths = rs.rand(H, W, 1, N).astype(np.float32)

for name, kw in (("poses_default", {}), ("poses_spherify", dict(spherify=True)), ("poses_norecenter", dict(recenter=False, spherify_hack=False, bd_factor=None))):
    L._load_data = lambda *a, **k: (poses.copy(), bds.copy(), imgs.copy(), masks.copy(), depths.copy(), None)
    images, p, b, rp, i_test, m, d, _ = L.load_llff_data("synthetic", factor=1, **kw)
    npz(name, poses_in=poses, bds_in=bds, poses=p, bds=b, render_poses=rp, i_test=i_test,
        recenter=int(kw.get("recenter", True)), spherify=int(kw.get("spherify", False)),
        hack=int(kw.get("spherify_hack", True)), bd_factor=-1.0 if kw.get("bd_factor", .75) is None else .75)
