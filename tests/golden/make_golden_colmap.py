"""Golden vectors for the COLMAP sparse-depth loader (DS_NeRF/load_llff.py:436-501).  Build container only:

    python tests/golden/make_golden_colmap.py

A small synthetic COLMAP model is written with the reference's own `write_images_binary` / `write_points3D_binary`
(colmapUtils/read_write_model.py), then the reference's `load_colmap_depth` runs on it unmodified; only its image
reader `_load_data` (imageio / cv2 are not in the container) is replaced by a function that returns the bounds.
The fixture stores the two binary files (inputs) and the per-image depth / coord / weight lists (outputs)."""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
from make_golden import _stub, npz

for n in ["cv2", "imageio"]:
    _stub(n)
sys.path.insert(0, "/root/reference/DS_NeRF")
import load_llff as L
from colmapUtils import read_write_model as RW

rs = np.random.RandomState(5)
N_IMG, N_PTS = 4, 60
pts = {}
for pid in range(1, N_PTS + 1):
    pts[pid] = RW.Point3D(id=pid, xyz=rs.uniform(-1, 1, 3) + np.array([0, 0, 4.0]), rgb=rs.randint(0, 255, 3),
                          error=np.array(rs.uniform(0.2, 2.0)), image_ids=np.array([1, 2]), point2D_idxs=np.array([0, 1]))
images = {}
for iid in range(1, N_IMG + 1):
    ang = 0.1 * iid
    q = np.array([np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0])      # rotation about y
    t = np.array([0.1 * iid, -0.05 * iid, 0.2])
    n2d = 25
    ids = rs.choice(np.arange(1, N_PTS + 1), n2d, replace=False).astype(np.int64)
    ids[rs.rand(n2d) < 0.2] = -1                                     # unmatched keypoints
    images[iid] = RW.Image(id=iid, qvec=q, tvec=t, camera_id=1, name=f"img{iid:03d}.png",
                           xys=rs.uniform(0, 400, size=(n2d, 2)), point3D_ids=ids)
bds = np.stack([3.2 + 0.2 * rs.rand(N_IMG), 4.6 + 0.3 * rs.rand(N_IMG)], 0)      # [2, N]: some points fall outside

d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, "sparse", "0"))
RW.write_images_binary(images, os.path.join(d, "sparse", "0", "images.bin"))
RW.write_points3d_binary(pts, os.path.join(d, "sparse", "0", "points3D.bin"))
L._load_data = lambda *a, **k: (None, bds.copy(), None, None, None, None)
out = L.load_colmap_depth(d, factor=8, bd_factor=.75)
assert 0 < len(out) <= N_IMG
fix = dict(images_bin=np.frombuffer(open(os.path.join(d, "sparse", "0", "images.bin"), "rb").read(), dtype=np.uint8),
           points_bin=np.frombuffer(open(os.path.join(d, "sparse", "0", "points3D.bin"), "rb").read(), dtype=np.uint8),
           bds=bds, n=len(out))
for i, e in enumerate(out):
    fix[f"depth{i}"], fix[f"coord{i}"], fix[f"weight{i}"] = e["depth"], e["coord"], e["weight"]
npz("colmap_depth", **fix)
print("images with depths:", len(out), [len(e["depth"]) for e in out])
