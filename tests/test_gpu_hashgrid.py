"""GPU: the hash-grid network (BASELINE config 5; NeRF_TCNN, DS_NeRF/run_nerf_helpers_tcnn.py:13-113) against
oracle/hashgrid_oracle.py.

PARITY UNPINNED: the reference delegates this arithmetic to tiny-cuda-nn, which is not in the reference tree (unpinned
git dependency, requirements.txt:13) and has no fixture in it; the oracle restates the published definition (its header
lists the sources).  What is established here is HIP kernels == that restatement, plus that the path trains."""
import argparse
import importlib
import math
import tempfile

import numpy as np
import pytest
import torch

from oracle import hashgrid_oracle as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import spin_nerf_amd as S
    assert torch.cuda.is_available()
    S._lib.load()
    return S


def make(S, seed, grid_gain=5e3, net_gain=1.5):
    """'trained-like' parameters: the U(-1e-4, 1e-4) initial table gives features of 1e-4 — too weak a test"""
    sd = H.init_params(seed)
    sd["encoder.params"] = sd["encoder.params"] * grid_gain
    sd["sigma_net.params"] = sd["sigma_net.params"] * net_gain
    sd["color_net.params"] = sd["color_net.params"] * net_gain
    net = S.NeRF_TCNN().cuda()
    net.load_state_dict(sd)
    return sd, net


def samples(seed, n_rays, sps, spread=40.0):
    rs = np.random.RandomState(seed)
    pts = torch.from_numpy(rs.uniform(-spread, spread, size=(n_rays, sps, 3)).astype(np.float32))
    dirs = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(n_rays, 3)).astype(np.float32)), dim=-1)
    return pts, dirs


def test_state_dict_keys_and_sizes(S):
    net = S.NeRF_TCNN()
    sd = net.state_dict()
    assert list(sd) == ["encoder.params", "sigma_net.params", "encoder_dir.params", "color_net.params"]
    _, total = H.level_table()
    assert sd["encoder.params"].numel() == 2 * total and sd["sigma_net.params"].numel() == 3072
    assert sd["encoder_dir.params"].numel() == 0 and sd["color_net.params"].numel() == 7168
    assert float(sd["encoder.params"].abs().max()) <= 1e-4


@pytest.mark.parametrize("spread", [40.0, 2.0])
def test_forward_matches_oracle(S, spread):
    sd, net = make(S, 0)
    pts, dirs = samples(1, 37, 23, spread)   # 851 samples: not a multiple of the 32-sample tile
    with torch.no_grad():
        out = net.query(pts.cuda(), dirs.cuda()).cpu()
    emu = H.run_network(sd, pts, dirs, bf16emu=True)
    ref = H.run_network(sd, pts, dirs)
    scale = float(ref.abs().max())
    assert scale > 0.05
    # same rounding points as the emulation (measured: agreement to 7e-7, i.e. fp32 summation order; an occasional flipped
    # bf16 rounding would show as ~1e-3): the encoding, both MLPs and the SH basis are the restated definition
    d = np.abs(out.numpy() - emu.numpy())
    assert float((d < 2e-6 * scale).mean()) > 0.995, float((d < 2e-6 * scale).mean())
    assert float(d.max()) < 1e-2 * scale, float(d.max())   # (measured: 1 element of 3404 off by 2e-3 of the range)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=3e-2 * scale, rtol=0)   # bf16 vs fp32 (measured 0.7 %)
    # the reference calling convention: forward(cat(position, direction))
    with torch.no_grad():
        out2 = net(torch.cat([pts.reshape(-1, 3), dirs[:, None].expand(pts.shape).reshape(-1, 3)], -1).cuda()).cpu()
    assert torch.equal(out2, out.reshape(-1, 4))


def test_ray_form_equals_point_form(S):
    sd, net = make(S, 2)
    rs = np.random.RandomState(3)
    n_rays, sps = 19, 64
    rays = torch.from_numpy(rs.normal(size=(n_rays, 11)).astype(np.float32))
    rays[:, 8:11] = torch.nn.functional.normalize(rays[:, 8:11], dim=-1)
    z = torch.sort(torch.from_numpy(rs.uniform(0.5, 6, size=(n_rays, sps)).astype(np.float32)), -1)[0]
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    with torch.no_grad():
        a = net.query_rays(rays.cuda(), z.cuda(), rays.cuda()[:, -3:])
        b = net.query(pts.cuda(), rays.cuda()[:, 8:11])
    assert torch.equal(a, b)


def test_encoding_matches_oracle_bit_for_bit(S):
    """the 32 features the training forward saves == bf16(oracle hash_encode): dense and hashed levels, both features"""
    L = S._lib
    lib = L.load()
    sd, net = make(S, 4)
    for spread in (40.0, 2.0):
        pts, dirs = samples(5, 41, 16, spread)
        n = 41 * 16
        p = pts.reshape(-1, 3).cuda().contiguous()
        vd = dirs.cuda().contiguous()
        raw = torch.empty(n, 4, device="cuda")
        act = torch.zeros(lib.snr_hashgrid_act_bytes(n), dtype=torch.uint8, device="cuda")
        L.check(lib.snr_hashgrid_forward(L.ptr(net.flat.detach()), L.ptr(net.packed_weights()), L.ptr(p), None, 0, None,
                                         L.ptr(vd), 3, n, 16, L.ptr(raw), L.ptr(act), L.stream()), "fwd")
        a = act.cpu().view(torch.bfloat16).reshape(-1, 2, 2, 32, 8).float()    # [tile][k-step q][half g][sample][e]
        enc = a.permute(0, 3, 1, 2, 4).reshape(-1, 32)[:n]                      # feature 16q + 8g + e
        ref = H.hash_encode(H.to_unit_cube(pts.reshape(-1, 3)), sd["encoder.params"]).to(torch.bfloat16).float()
        d = (enc - ref).abs()
        # (fp32 summation order may flip a bf16 rounding once in 10^4 values)
        assert float((d == 0).float().mean()) > 0.999 and float(d.max()) <= 2.0 ** -8 * float(ref.abs().max())


@pytest.mark.parametrize("layout", ["scattered", "along_rays"])
def test_backward_matches_autograd_through_the_oracle(S, layout):
    """'scattered': every sample in its own cells (per-lane atomics); 'along_rays': 32 consecutive samples of a ray share
    the coarse cells — the merged path (half-wave reduction, one leader's atomics) carries most of the table gradient."""
    sd, net = make(S, 4)
    if layout == "scattered":
        pts, dirs = samples(5, 41, 16)
    else:
        rs = np.random.RandomState(8)
        n_rays, sps = 12, 64
        o = torch.from_numpy(rs.normal(scale=0.3, size=(n_rays, 3)).astype(np.float32)) + torch.tensor([0., 0., 4.])
        dirs = torch.nn.functional.normalize(torch.from_numpy((rs.normal(size=(n_rays, 3)) * [0.3, 0.3, 0.1] + [0, 0, -1]).astype(np.float32)), dim=-1)
        z = torch.linspace(2.0, 6.0, sps)[None, :] + torch.from_numpy(rs.uniform(0, 0.05, size=(n_rays, sps)).astype(np.float32))
        pts = o[:, None, :] + dirs[:, None, :] * z[:, :, None]
    rs = np.random.RandomState(6)
    d_raw = torch.from_numpy(rs.normal(size=tuple(pts.shape[:2]) + (4,)).astype(np.float32))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.numel()}
    full = dict(sd); full.update(p)
    (H.run_network(full, pts, dirs, bf16emu=True) * d_raw).sum().backward()
    out = net.query(pts.cuda(), dirs.cuda())
    (out * d_raw.cuda()).sum().backward()
    got = net.named_views(net.flat.grad)
    for k, v in p.items():
        a, b = got[k].cpu().double(), v.grad.double()
        rel = float((a - b).norm() / b.norm())
        cos = float((a @ b) / (a.norm() * b.norm()))
        # the emulation rounds the back-propagated gradients at its casts, the kernel at its MFMA inputs (measured 0.2-0.5 %)
        assert rel < 2e-2 and cos > 0.9995, f"{k}: relative L2 error {rel:.2e}, cosine {cos:.5f}"
    # per level as well: a level whose merged sums went to the wrong entries would hide in the total
    levels, total = H.level_table()
    a = got["encoder.params"].cpu().double().reshape(total, 2)
    b = p["encoder.params"].grad.double().reshape(total, 2)
    for l, (scale, res, n, off, hashed) in enumerate(levels):
        rel = float((a[off:off + n] - b[off:off + n]).norm() / b[off:off + n].norm())
        assert rel < 3e-2, f"level {l}: relative L2 error {rel:.2e}"
    # untouched table entries have an exactly zero gradient, and the gradient is rebuilt (not accumulated) per call
    gt = got["encoder.params"]
    assert float((gt == 0).float().mean()) > 0.9
    net.flat.grad = None
    (net.query(pts.cuda(), dirs.cuda()) * d_raw.cuda()).sum().backward()
    again = net.named_views(net.flat.grad)["encoder.params"]
    assert float((again - gt).abs().max()) <= 1e-4 * float(gt.abs().max())   # atomics: order-dependent rounding only


def test_network_alone_learns_a_regression_target(S):
    """Adam on the kernels' gradients fits a smooth colour field and a density step (no rendering involved): the loss
    drops by two orders of magnitude in 300 steps (measured 0.44 -> 0.004)."""
    torch.manual_seed(0)
    net = S.NeRF_TCNN().cuda()
    m, v = torch.zeros_like(net.flat.data), torch.zeros_like(net.flat.data)
    g = torch.Generator(device="cuda").manual_seed(1)
    hist = []
    for it in range(300):
        pts = torch.rand(256, 16, 3, device="cuda", generator=g) * 4 - 2
        dirs = torch.nn.functional.normalize(torch.randn(256, 3, device="cuda", generator=g), dim=-1)
        tgt = torch.cat([torch.sin(pts * 2.0), (pts.norm(dim=-1, keepdim=True) < 1.0).float() * 3.0], -1)
        net.flat.grad = None
        loss = ((net.query(pts, dirs) - tgt) ** 2).mean()
        loss.backward()
        S.adam_step_(net.flat.data, net.flat.grad, m, v, 1e-2, it + 1)
        net.mark_weights_changed()
        hist.append(float(loss.detach()))
    assert np.mean(hist[-10:]) < 0.03 * np.mean(hist[:5]), (np.mean(hist[:5]), np.mean(hist[-10:]))


def test_render_with_hash_networks_matches_oracle_pipeline(S):
    """render() over two NeRF_TCNN networks (create_nerf_tcnn's kwargs) against the oracle's render() with the hash-grid
    restatement plugged in as its network: maps, loss and the gradients of both networks."""
    from oracle import nerf_oracle as O
    sd_c, net_c = make(S, 11, grid_gain=3e4)
    sd_f, net_f = make(S, 12, grid_gain=3e4)
    H_, W_, f = 12, 16, 20.0
    rs = np.random.RandomState(2)
    n, Nc, Nf = 24, 64, 64
    ro = torch.from_numpy(rs.normal(scale=0.2, size=(n, 3)).astype(np.float32)) + torch.tensor([0., 0., 4.])
    rd = torch.from_numpy((rs.normal(size=(n, 3)) * [0.3, 0.3, 0.1] + [0, 0, -1]).astype(np.float32))
    rays = torch.stack([ro, rd], 0)
    rnd = dict(t_rand=torch.from_numpy(rs.uniform(size=(n, Nc)).astype(np.float32)),
               u=torch.from_numpy(rs.uniform(size=(n, Nf)).astype(np.float32)), noise_c=None, noise_f=None)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False, near=2.0, far=6.0)
    rgb, disp, acc, depth, ex = S.render(H_, W_, f, rays=rays.cuda(), retraw=True,
                                         randoms={k: (v.cuda() if v is not None else None) for k, v in rnd.items()}, **kw)
    target = torch.from_numpy(rs.uniform(size=(n, 3)).astype(np.float32))
    loss = S.img2mse(rgb, target.cuda()) + S.img2mse(ex["rgb0"], target.cuda())
    loss.backward()
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items() if v.numel()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items() if v.numel()}
    fc, ff = dict(sd_c), dict(sd_f)
    fc.update(pc); ff.update(pf)
    mlp = lambda sd, x, **_: H.nerf_tcnn_forward(sd, x, bf16emu=True)
    r = O.render(H_, W_, f, rays=rays, sd_coarse=fc, sd_fine=ff, randoms=rnd, retraw=True, N_samples=Nc, N_importance=Nf,
                 perturb=1.0, white_bkgd=True, lindisp=False, use_viewdirs=True, ndc=False, near=2.0, far=6.0, i_embed=-1,
                 mlp=mlp)
    r_loss = O.img2mse(r[0], target) + O.img2mse(r[4]["rgb0"], target)
    r_loss.backward()
    assert float(r[2].mean()) > 0.2, "the test scene must not be empty"
    np.testing.assert_allclose(ex["rgb0"].detach().cpu().numpy(), r[4]["rgb0"].detach().numpy(), atol=2e-3)
    d = (rgb.detach().cpu() - r[0].detach()).abs()
    assert float((d < 2e-3).float().mean()) > 0.8 and float(d.max()) < 0.1     # free-running fine stage (test_gpu_render.py)
    assert abs(float(loss) - float(r_loss)) < 2e-2 * abs(float(r_loss))
    for net, p, tol in ((net_c, pc, 3e-2), (net_f, pf, 1.5e-1)):
        got = net.named_views(net.flat.grad)
        for k, v in p.items():
            a, b = got[k].cpu().double(), v.grad.double()
            rel = float((a - b).norm() / b.norm())
            assert rel < tol, f"{k}: relative L2 error {rel:.2e}"


def _args(**over):
    a = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=64, N_samples=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=1e-2, basedir=tempfile.mkdtemp(),
        expname="", ft_path=None, no_reload=True, perturb=1.0, white_bkgd=False, raw_noise_std=0.0, dataset_type="llff",
        no_ndc=True, lindisp=False, sigma_loss=False, no_coarse=False, masked_NeRF=False, object_removal=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_create_nerf_tcnn_contract_and_a_training_step(S):
    """create_nerf_tcnn (run_nerf.py:499-590): kwargs, identity embedders, fresh networks (checkpoints are never reloaded
    on this path, :548), then RenderTrainer steps and a full-frame render run on them."""
    import contextlib, io
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, start, grad_vars, opt = S.create_nerf_tcnn(_args(), device=torch.device("cuda"))
    assert isinstance(kw_train["network_fn"], S.NeRF_TCNN) and isinstance(kw_train["network_fine"], S.NeRF_TCNN)
    assert set(kw_train) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                             "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"} and start == 0
    assert len(grad_vars) == 2 and isinstance(opt, torch.optim.Adam) and kw_test["perturb"] is False
    kw_train.update(near=2.0, far=6.0); kw_test.update(near=2.0, far=6.0)
    tr = RenderTrainer(kw_train, lrate=5e-4, lrate_decay=250)
    dev = torch.device("cuda")
    H_, W_, f = 24, 32, 40.0
    c2w = torch.eye(4)[:3, :4].clone(); c2w[2, 3] = 4.0
    ro, rd = S.get_rays(H_, W_, f, c2w.to(dev))
    rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    before = [n.flat.detach().clone() for n in tr.nets]
    for it in range(3):
        loss, rgb = tr.step(H_, W_, f, rays, torch.rand(H_ * W_, 3, device=dev))
        assert np.isfinite(float(loss))
    for n, b in zip(tr.nets, before):
        assert float((n.flat.detach() - b).abs().max()) > 0 and bool(torch.isfinite(n.flat).all())
    with torch.no_grad():
        rgb, disp, acc, depth, ex = S.render(H_, W_, f, chunk=32768, c2w=c2w.to(dev), **kw_test)
    assert tuple(rgb.shape) == (H_, W_, 3) and bool(torch.isfinite(rgb).all())


@pytest.mark.timeout(600)
@pytest.mark.parametrize("noise", [1.0, 0.0])
def test_hash_grid_training_learns_the_analytic_sphere(S, noise):
    """create_nerf_tcnn's networks under the reference's recipe (ReLU density through raw2outputs with raw_noise_std = 1,
    the value of the reference's configs; Adam; lrate 1e-2) on the analytic sphere of tests/test_gpu_train.py: 600
    iterations of 512 rays.  Measured on MI355X over six initialisation seeds (tests/probes/hashgrid_seed_sweep.py,
    profiles/r02_hashgrid_train.txt): 34.9-36.7 dB over the last 50 iterations, 34.6-36.8 dB for a full-frame render of a
    training view; noise-free runs of the same six seeds reach 37-38 dB (they died into an all-empty state until the
    compositing backward's 0 * NaN on rays that hit nothing was fixed in round 2).  The CPU oracle
    trained from the same initial parameters follows the HIP path's curve (17.7 / 25.8 / 27.7 / 29.2 / 30.3 dB vs
    17.6 / 25.9 / 27.6 / 28.9 / 29.7 dB at iterations 50...250, tests/probes/hashgrid_oracle_train.py)."""
    import contextlib, io
    from test_gpu_train import sphere_scene, H as HH, W as WW, FOCAL, NEAR, FAR
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, kw_test, *_ = S.create_nerf_tcnn(_args(lrate=1e-2, raw_noise_std=noise), device=dev)
    kw_train.update(near=NEAR, far=FAR); kw_test.update(near=NEAR, far=FAR)
    tr = RenderTrainer(kw_train, lrate=1e-2, lrate_decay=250)

    def camera(a):
        eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
        z = eye / eye.norm()
        x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
        return torch.cat([torch.stack([x, torch.linalg.cross(z, x), z], 1), eye[:, None]], 1).to(dev)
    rays_all, tgt_all = [], []
    for k in range(6):
        ro, rd = S.get_rays(HH, WW, FOCAL, camera(2 * math.pi * k / 6))
        rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
        tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), False))
    rays_all, tgt_all = torch.cat(rays_all, 1), torch.cat(tgt_all, 0)
    g = torch.Generator().manual_seed(1)
    ps = []
    for it in range(600):
        sel = torch.randint(0, rays_all.shape[1], (512,), generator=g).to(dev)
        loss, rgb = tr.step(HH, WW, FOCAL, rays_all[:, sel].contiguous(), tgt_all[sel])
        ps.append(float(-10.0 * torch.log10(torch.mean((rgb - tgt_all[sel]) ** 2))))
    def view_psnr(c2w):
        with torch.no_grad():
            rgb, disp, acc, depth, ex = S.render(HH, WW, FOCAL, chunk=32768, c2w=c2w, **kw_test)
        ro, rd = S.get_rays(HH, WW, FOCAL, c2w)
        return float(-10.0 * torch.log10(torch.mean((rgb - sphere_scene(ro, rd, False)) ** 2))), float(acc.mean())
    seen, acc = view_psnr(camera(0.0))                      # a training camera, rendered without perturbation
    held_out, _ = view_psnr(camera(2 * math.pi * 0.5 / 6))  # between two training cameras: printed, not gated — six
    # views do not constrain a 2^19-entry table with 1000 cells per unit length (the oracle recipe overfits the same way)
    print(f"hash-grid training PSNR: start {np.mean(ps[:5]):.2f} dB, last 50 of 600: {np.mean(ps[-50:]):.2f} dB; full-frame "
          f"render of a training view {seen:.2f} dB (mean opacity {acc:.3f}), of a held-out view {held_out:.2f} dB")
    assert np.mean(ps[-50:]) > 30.0, np.mean(ps[-50:])
    assert seen > 30.0, seen


def test_autograd_free_step_equals_the_autograd_step_for_hash_networks(S, monkeypatch):
    """RenderTrainer.step's direct route (train.py: _step_direct) also drives NeRF_TCNN networks; with the same injected
    draws it must give the loss, render and gradients of the render() + autograd route.  The table gradient is summed by
    atomics in both routes, so the comparison is to rounding, not bit-for-bit."""
    import contextlib, io
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    n, Nc, Nf = 200, 64, 64
    rs = np.random.RandomState(3)
    ro = torch.from_numpy(rs.normal(scale=0.2, size=(n, 3)).astype(np.float32)) + torch.tensor([0., 0., 4.])
    rd = torch.from_numpy((rs.normal(size=(n, 3)) * [0.3, 0.3, 0.1] + [0, 0, -1]).astype(np.float32))
    rays = torch.stack([ro, rd], 0).to(dev)
    target = torch.from_numpy(rs.uniform(size=(n, 3)).astype(np.float32)).to(dev)
    rnd = dict(t_rand=torch.from_numpy(rs.uniform(size=(n, Nc)).astype(np.float32)).to(dev),
               u=torch.from_numpy(rs.uniform(size=(n, Nf)).astype(np.float32)).to(dev),
               noise_c=torch.from_numpy(rs.normal(size=(n, Nc)).astype(np.float32)).to(dev),
               noise_f=torch.from_numpy(rs.normal(size=(n, Nc + Nf)).astype(np.float32)).to(dev))
    trainers = []
    for _ in range(2):
        with contextlib.redirect_stdout(io.StringIO()):
            kw, *_ = S.create_nerf_tcnn(_args(raw_noise_std=1.0), device=dev)
        for net, seed in ((kw["network_fn"], 11), (kw["network_fine"], 12)):
            net.load_state_dict(make(S, seed, grid_gain=3e4)[0])
        kw.update(near=2.0, far=6.0)
        trainers.append((RenderTrainer(kw, lrate=1e-2, lrate_decay=250), [kw["network_fn"], kw["network_fine"]]))
    (ta, na), (tb, nb) = trainers
    assert ta._direct_ok(rays, 32768, {"randoms": rnd})
    la, rgb_a = ta.step(24, 32, 40.0, rays, target, randoms=rnd)
    monkeypatch.setenv("SNR_NO_DIRECT_STEP", "1")
    assert not tb._direct_ok(rays, 32768, {"randoms": rnd})
    lb, rgb_b = tb.step(24, 32, 40.0, rays, target, randoms=rnd)
    assert abs(float(la) - float(lb)) < 1e-6 * abs(float(lb))
    assert float((rgb_a - rgb_b).abs().max()) < 1e-6
    for a, b in zip(na, nb):
        rel = float((a.flat.grad - b.flat.grad).norm() / b.flat.grad.norm())
        assert rel < 1e-5, rel
        assert float(b.flat.grad.abs().max()) > 0


def test_graph_replayed_step_with_hash_networks(S):
    """the captured-graph route of RenderTrainer.step on NeRF_TCNN networks: counters, learning-rate schedule and the loss
    trajectory follow the eager trainer (the table gradient is summed by atomics, so parameters agree only statistically)."""
    import contextlib, io
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    rs = np.random.RandomState(5)
    n = 256
    ro = torch.from_numpy(rs.normal(scale=0.2, size=(n, 3)).astype(np.float32)) + torch.tensor([0., 0., 4.])
    rd = torch.from_numpy((rs.normal(size=(n, 3)) * [0.3, 0.3, 0.1] + [0, 0, -1]).astype(np.float32))
    rays = torch.stack([ro, rd], 0).to(dev)
    target = torch.from_numpy(rs.uniform(size=(n, 3)).astype(np.float32)).to(dev)
    out = []
    for graph in (False, True):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            kw, *_ = S.create_nerf_tcnn(_args(raw_noise_std=1.0), device=dev)
        kw.update(near=2.0, far=6.0)
        tr = RenderTrainer(kw, lrate=1e-2, lrate_decay=250, graph=graph)
        losses = [float(tr.step(24, 32, 40.0, rays, target)[0]) for _ in range(8)]
        out.append((tr, losses))
    (te, le), (tg, lg) = out
    assert tg._graph is not None
    assert (te._draws, te.opt_step, te.global_step) == (tg._draws, tg.opt_step, tg.global_step) and te.current_lr() == tg.current_lr()
    assert le[-1] < le[0] and lg[-1] < lg[0], (le, lg)    # both runs descend (random targets: slowly)
    for a, b in zip(le, lg):
        assert abs(a - b) < 2e-2 * abs(a), (le, lg)
