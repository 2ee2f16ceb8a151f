"""MVSeg's 5-channel raw2outputs (MVSeg/DS_NeRF/run_nerf_helpers.py:350-413) against fixtures
generated from the reference (tests/golden/make_golden_mvseg.py): oracle on CPU, HIP path on the GPU, forward
and the gradient of BCE(prob_map) + mse(rgb)."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from helpers import load, T

CASES = ["mvseg_r2o_s64", "mvseg_r2o_s192_white", "mvseg_r2o_only_object", "mvseg_r2o_only_object_thr"]


def _kw(g):
    """the only_object arguments of a fixture (MVSeg/DS_NeRF/run_nerf_helpers.py:383-397, 410-411)"""
    if "only_object" not in g:
        return {}
    return dict(only_object=True, threshold=None if float(g["threshold"]) < 0 else float(g["threshold"]),
                harsh_bg_remove=bool(g["harsh"]))


def _check(fn, g, dev, atol):
    raw = T(g["raw"]).to(dev).requires_grad_(True)
    out = fn(raw, T(g["z_vals"]).to(dev), T(g["rays_d"]).to(dev), white_bkgd=bool(g["white"]), **_kw(g))
    names = ["rgb", "disp", "acc", "weights", "depth", "prob", "logits"]
    for n, o in zip(names, out):
        np.testing.assert_allclose(o.detach().cpu().numpy(), g[n], atol=atol, rtol=2e-4 if n in ("disp", "depth") else 1e-5,
                                   err_msg=n)
    tgt = T(g["target"]).to(dev)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out[5], tgt) + ((out[0]) ** 2).mean()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(raw.grad.cpu().numpy(), g["d_raw"], atol=2e-7, rtol=1e-3)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_mvseg_reference(name):
    _check(O.raw2outputs_mvseg, load(name), "cpu", 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_mvseg_reference(name):
    import spin_nerf_amd as S
    _check(lambda raw, z, d, white_bkgd, **kw: S.raw2outputs_mvseg(raw, z, d, 0, white_bkgd, **kw), load(name), "cuda", 2e-6)
