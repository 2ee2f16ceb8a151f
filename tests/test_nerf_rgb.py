"""NeRF_RGB (DS_NeRF/run_nerf_helpers.py:159-216): colour network whose density comes from a frozen
network.  Fixture generated from the reference module (tests/golden/make_golden_rgb.py)."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from helpers import load, T


def _params():
    sd_a = O.make_wild_params(seed=31)
    sd_rgb = {k: v for k, v in O.make_wild_params(seed=32).items() if not k.startswith("alpha_linear")}
    return sd_a, sd_rgb


def _check_grads(g, named):
    for k, gr in named.items():
        gr = gr.reshape(-1).detach().cpu()
        sub = (gr[::61] if gr.numel() > 4096 else gr).numpy()
        ref = g["g_" + k]
        rel = np.linalg.norm(sub - ref) / max(np.linalg.norm(ref), 1e-20)
        assert rel < 2e-3, f"{k}: {rel:.2e}"
        assert abs(float(gr.double().norm()) / float(g["g_" + k + ".norm"]) - 1) < 2e-3, k


def test_oracle_matches_reference_module():
    g = load("nerf_rgb_vd")
    sd_a, sd_rgb = _params()
    p = {k: v.clone().requires_grad_(True) for k, v in sd_rgb.items()}
    out = O.nerf_rgb_forward(p, sd_a, T(g["x"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=2e-5, rtol=2e-5)
    (out * T(g["d_out"])).sum().backward()
    _check_grads(g, {k: v.grad for k, v in p.items()})


def test_state_dict_layout_matches_reference():
    import spin_nerf_amd as S
    g = load("nerf_rgb_vd")
    alpha = S.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True)
    net = S.NeRF_RGB(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True, alpha_model=alpha)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    sd_a, sd_rgb = _params()
    net.load_state_dict({**sd_rgb, **{"alpha_model." + k: v for k, v in sd_a.items()}})   # strict
    v = net.named_views()
    assert float(v["alpha_linear.weight"].abs().max()) == 0.0 and float(v["alpha_linear.bias"].abs().max()) == 0.0
    assert torch.equal(net.state_dict()["rgb_linear.weight"], sd_rgb["rgb_linear.weight"])
    assert torch.equal(alpha.state_dict()["alpha_linear.weight"], sd_a["alpha_linear.weight"])


@pytest.mark.gpu
def test_hip_matches_reference_module():
    import spin_nerf_amd as S
    g = load("nerf_rgb_vd")
    sd_a, sd_rgb = _params()
    alpha = S.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True, precision="fp32").cuda()
    alpha.load_state_dict(sd_a)
    net = S.NeRF_RGB(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True, alpha_model=alpha,
                     precision="fp32").cuda()
    net.load_state_dict({**sd_rgb, **{"alpha_model." + k: v for k, v in sd_a.items()}})
    d_out = T(g["d_out"]).cuda()
    out = net(T(g["x"]).cuda())                                   # reference calling convention
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], atol=3e-5, rtol=3e-5)
    (out * d_out).sum().backward()
    views = net.named_views(net.flat.grad)
    assert float(views["alpha_linear.weight"].abs().max()) == 0.0     # its output is replaced: no gradient
    assert alpha.flat.grad is None                                     # the density network is frozen (no_grad)
    _check_grads(g, {k: v for k, v in views.items() if not k.startswith("alpha_linear")})
    out2 = net.query(T(g["pts"]).cuda()[:, None, :], T(g["dirs"]).cuda())[:, 0]
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["out"], atol=3e-5, rtol=3e-5)


@pytest.mark.gpu
def test_render_with_frozen_density_and_no_coarse_network(tmp_path):
    """render() with NeRF_RGB: the coarse pass falls back to network_fine.alpha_model when network_fn is None
    (run_nerf.py:680-692); gradients reach the colour network only."""
    import argparse, contextlib, io
    import spin_nerf_amd as S
    dev = torch.device("cuda")
    torch.manual_seed(0)
    donor = S.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True)
    torch.save({"network_fine_state_dict": donor.state_dict()}, tmp_path / "alpha.tar")
    (tmp_path / "run").mkdir()
    for no_coarse in (False, True):
        args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=32,
                                  N_samples=64, alpha_model_path=str(tmp_path / "alpha.tar"), netdepth=8, netwidth=256,
                                  netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, basedir=str(tmp_path),
                                  expname="run", ft_path=None, no_reload=True, perturb=1.0, white_bkgd=False,
                                  raw_noise_std=0.0, dataset_type="llff", no_ndc=True, lindisp=False, sigma_loss=False,
                                  no_coarse=no_coarse, precision="fp32")
        with contextlib.redirect_stdout(io.StringIO()):
            kw, *_ = S.create_nerf(args, device=dev)
        kw.update(near=2.0, far=6.0)
        c2w = torch.eye(4, device=dev)[:3, :4].clone(); c2w[2, 3] = 4.0
        ro, rd = S.get_rays(8, 10, 12.0, c2w)
        rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
        rgb, disp, acc, depth, ex = S.render(8, 10, 12.0, chunk=64, rays=rays, retraw=True, **kw)
        assert rgb.shape == (80, 3) and torch.isfinite(rgb).all() and ex["raw"].shape == (80, 96, 4)
        rgb.sum().backward()
        fine = kw["network_fine"]
        assert float(fine.flat.grad.abs().max()) > 0 and fine.alpha_model.flat.grad is None
        # the density the fine pass composites is the frozen network's
        with torch.no_grad():
            a = fine.alpha_model.query_rays(S.ops.pack_rays(rays[0], rays[1], 8, 10, 12.0, ndc=False, near=2.0, far=6.0,
                                                            use_viewdirs=True), ex["z_vals"],
                                            torch.nn.functional.normalize(rays[1], dim=-1))
        assert torch.allclose(ex["raw"][..., 3], a[..., 3], atol=1e-5)
