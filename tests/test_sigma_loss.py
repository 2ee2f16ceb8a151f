"""SigmaLoss (`--sigma_loss`, DS_NeRF/loss.py:8-44; a caller of network_query_fn named in SURVEY.md §8b) against
fixtures generated from the reference (tests/golden/make_golden_sigma.py) with the random draws injected."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from helpers import load, T

CASES = ["sigma_loss_det", "sigma_loss_rand"]


def _grad_check(g, w0_grad, alpha_grad, tol):
    sub = w0_grad.reshape(-1).cpu()[::7].numpy()
    rel = np.linalg.norm(sub - g["g_pts0"]) / np.linalg.norm(g["g_pts0"])
    assert rel < tol, rel
    assert abs(float(w0_grad.double().norm()) / float(g["g_pts0_norm"]) - 1) < tol
    np.testing.assert_allclose(alpha_grad.reshape(-1).cpu().numpy(), g["g_alpha"], rtol=tol, atol=1e-6 * np.abs(g["g_alpha"]).max())


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = load(name)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.make_wild_params(seed=41).items()}
    out = O.sigma_loss(sd, T(g["rays_o"]), T(g["rays_d"]), T(g["viewdirs"]), T(g["near"]), T(g["depths"]), 64,
                       perturb=float(g["perturb"]), t_rand=T(g["t_rand"]) if "t_rand" in g else None,
                       noise=T(g["noise"]) if "noise" in g else None)
    np.testing.assert_allclose(out.detach().numpy(), g["loss"], rtol=2e-5, atol=1e-7)
    out.sum().backward()
    _grad_check(g, sd["pts_linears.0.weight"].grad, sd["alpha_linear.weight"].grad, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference(name):
    import spin_nerf_amd as S
    g = load(name)
    net = S.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True, precision="fp32").cuda()
    net.load_state_dict(O.make_wild_params(seed=41))
    sl = S.SigmaLoss(64, float(g["perturb"]), float(g["std"]))
    rnd = {"t_rand": T(g["t_rand"]).cuda(), "noise": T(g["noise"]).cuda()} if "t_rand" in g else None
    cu = lambda k: T(g[k]).cuda()
    out = sl.calculate_loss(cu("rays_o"), cu("rays_d"), cu("viewdirs"), cu("near"), cu("far"), cu("depths"),
                            lambda p, v, n: S.run_network(p, v, n), net, randoms=rnd)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["loss"], rtol=1e-4, atol=1e-6)
    out.sum().backward()
    v = net.named_views(net.flat.grad)
    _grad_check(g, v["pts_linears.0.weight"], v["alpha_linear.weight"], 2e-3)


def test_create_nerf_wires_sigma_loss(tmp_path):
    import argparse
    import spin_nerf_amd as S
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=0, N_samples=32,
                              alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              netchunk=65536, lrate=5e-4, basedir=str(tmp_path), expname="", ft_path=None, no_reload=True,
                              perturb=1.0, white_bkgd=False, raw_noise_std=0.5, dataset_type="llff", no_ndc=False,
                              lindisp=False, sigma_loss=True, no_coarse=False)
    kw_train, kw_test, *_ = S.create_nerf(args, device=torch.device("cpu"))
    sl = kw_train["sigma_loss"]
    assert isinstance(sl, S.SigmaLoss) and (sl.N_samples, sl.perturb, sl.raw_noise_std) == (32, 1.0, 0.5)
    assert "sigma_loss" not in kw_test      # attached after the test kwargs are copied (run_nerf.py:486-492)
