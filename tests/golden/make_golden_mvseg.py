"""Golden vectors for the MVSeg variant of raw2outputs (MVSeg/DS_NeRF/run_nerf_helpers.py:350-413):
5 raw channels, the composited logit `prob_map = sum(w.detach() * logit)` and the gradients of a loss on it.

Run ONLY in the build container:   python tests/golden/make_golden_mvseg.py
Imports the reference read-only with stubs for cv2 / clip / torchvision (module-scope imports it does not need
for this function).  Stores inputs and outputs, nothing of the source."""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch
from make_golden import _stub, npz

for n in ["cv2", "torchvision"]:
    _stub(n)
_stub("clip", load=lambda *a, **k: (None, None))   # module scope calls clip.load(...) (helpers:469)
spec = importlib.util.spec_from_file_location("mvseg_helpers", "/root/reference/MVSeg/DS_NeRF/run_nerf_helpers.py")
H = importlib.util.module_from_spec(spec)
spec.loader.exec_module(H)
torch.autograd.set_detect_anomaly(False)

rs = np.random.RandomState(7)
for name, S, white in (("mvseg_r2o_s64", 64, False), ("mvseg_r2o_s192_white", 192, True)):
    n = 24
    raw = torch.from_numpy(rs.normal(size=(n, S, 5)).astype(np.float32) * 2).requires_grad_(True)
    z = torch.sort(torch.from_numpy(rs.uniform(2, 6, size=(n, S)).astype(np.float32)), -1)[0]
    d = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32))
    out = H.raw2outputs(raw, z, d, 0, white, pytest=False)
    rgb, disp, acc, w, depth, prob, logits = out
    tgt = torch.from_numpy(rs.uniform(size=(n,)).astype(np.float32))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(prob, tgt) + H.img2mse(rgb, torch.zeros_like(rgb))
    loss.backward()
    npz(name, raw=raw, z_vals=z, rays_d=d, white=int(white), rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth,
        prob=prob, logits=logits, target=tgt, loss=loss, d_raw=raw.grad)

# only_object (helpers:383-397, 410-411): alpha gated by the logit, with and without the threshold + smoothing branch
for name, S, thr, harsh in (("mvseg_r2o_only_object", 64, None, True), ("mvseg_r2o_only_object_thr", 64, 0.6, False)):
    n = 24
    raw = torch.from_numpy(rs.normal(size=(n, S, 5)).astype(np.float32) * 2).requires_grad_(True)
    z = torch.sort(torch.from_numpy(rs.uniform(2, 6, size=(n, S)).astype(np.float32)), -1)[0]
    d = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32))
    out = H.raw2outputs(raw, z, d, 0, True, pytest=False, only_object=True, threshold=thr, harsh_bg_remove=harsh)
    rgb, disp, acc, w, depth, prob, logits = out
    tgt = torch.from_numpy(rs.uniform(size=(n,)).astype(np.float32))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(prob, tgt) + H.img2mse(rgb, torch.zeros_like(rgb))
    loss.backward()
    npz(name, raw=raw, z_vals=z, rays_d=d, white=1, rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth,
        prob=prob, logits=logits, target=tgt, loss=loss, d_raw=raw.grad, threshold=-1.0 if thr is None else thr,
        harsh=int(harsh), only_object=1)
