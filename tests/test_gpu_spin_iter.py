"""GPU: the three-render SPIn-NeRF iteration (run_nerf.py:1455-1521) against the CPU oracle with
the same injected randoms, fp32 path: loss value and parameter gradients of both networks."""
import importlib

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def test_spin_iteration_loss_and_gradients_match_oracle():
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    H, W, focal, near, far = 20, 24, 30.0, 2.0, 6.0
    Nc, Nf, N = 64, 32, 40
    sd_c = O.init_nerf_params(seed=3)
    sd_f = O.init_nerf_params(seed=4)

    def mk(sd):
        n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="fp32").cuda()
        n.load_state_dict(sd)
        return n
    net_c, net_f = mk(sd_c), mk(sd_f)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=1.0, ndc=False, lindisp=False, near=near, far=far)
    tr = train.RenderTrainer(kw, lrate=5e-4)

    g = torch.Generator().manual_seed(0)
    c2w = torch.eye(4)[:3, :4].clone(); c2w[2, 3] = 4.0
    ro, rd = O.get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)

    def batch():
        sel = torch.randperm(H * W, generator=g)[:N]
        return torch.stack([ro[sel], rd[sel]], 0)
    rays_clf, rays_all, rays_inp = batch(), batch(), batch()
    t_clf, t_all = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
    d_inp = torch.rand(N, generator=g) * 0.3 + 0.1

    def rnd(seed):
        gg = torch.Generator().manual_seed(seed)
        return {"t_rand": torch.rand(N, Nc, generator=gg), "u": torch.rand(N, Nf, generator=gg),
                "noise_c": torch.randn(N, Nc, generator=gg), "noise_f": torch.randn(N, Nc + Nf, generator=gg)}
    rnds = [rnd(1), rnd(2), rnd(3)]

    # ---- oracle ----
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    okw = dict(sd_coarse=pc, sd_fine=pf, N_samples=Nc, N_importance=Nf, perturb=1.0,   # (noise comes pre-scaled)
               white_bkgd=False, lindisp=False, use_viewdirs=True, ndc=False, near=near, far=far, retraw=True)
    rgb, disp, acc, depth, ex = O.render(H, W, focal, rays=rays_clf, randoms=rnds[0], **okw)
    rgb_c, _, _, _, ex_c = O.render(H, W, focal, rays=rays_all, randoms=rnds[1], detach_weights=True, **okw)
    _, disp_i, _, _, ex_i = O.render(H, W, focal, rays=rays_inp, randoms=rnds[2], **okw)
    ref = (O.img2mse(rgb, t_clf) + O.img2mse(rgb_c, t_all) + O.img2mse(ex_c["rgb0"], t_all) + O.img2mse(ex["rgb0"], t_clf)
           + O.img2mse(disp_i, d_inp) + O.img2mse(ex_i["disp0"], d_inp))
    ref.backward()

    # ---- HIP path ----
    cu = lambda t: t.cuda()
    loss, outs = tr.spin_loss(H, W, focal, cu(rays_clf), cu(t_clf), cu(rays_all), cu(t_all), cu(rays_inp), cu(d_inp),
                              randoms=[{k: cu(v) for k, v in r.items()} for r in rnds])
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-4 * abs(float(ref.detach())), (float(loss.detach()), float(ref.detach()))
    loss.backward()
    for net, p in ((net_c, pc), (net_f, pf)):
        got = net.named_views(net.flat.grad)
        for k, v in p.items():
            a, b = got[k].cpu().double().reshape(-1), v.grad.double().reshape(-1)
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert rel < 5e-3, f"{k}: relative L2 error {rel:.2e}"

    # the batched form (first and third render in one launch sequence) gives the same loss and gradients
    g_ref = [n.flat.grad.clone() for n in (net_c, net_f)]
    for n in (net_c, net_f):
        n.flat.grad = None
    loss_b, _ = tr.spin_loss(H, W, focal, cu(rays_clf), cu(t_clf), cu(rays_all), cu(t_all), cu(rays_inp), cu(d_inp),
                             randoms=[{k: cu(v) for k, v in r.items()} for r in rnds], batched=True)
    assert abs(float(loss_b.detach()) - float(loss.detach())) < 1e-6 * abs(float(loss.detach()))
    loss_b.backward()
    for n, gr in zip((net_c, net_f), g_ref):
        assert float((n.flat.grad - gr).norm() / gr.norm()) < 1e-5

    # the optimiser step runs and reports a PSNR
    l2, psnr = tr.spin_iteration(H, W, focal, cu(rays_clf), cu(t_clf), cu(rays_all), cu(t_all), cu(rays_inp), cu(d_inp))
    assert np.isfinite(float(l2)) and np.isfinite(float(psnr)) and tr.global_step == 1
    # NaN disparity target: the geometry term is dropped like the reference does
    l3, _ = tr.spin_loss(H, W, focal, cu(rays_clf), cu(t_clf), cu(rays_all), cu(t_all), cu(rays_inp),
                         torch.full((N,), float("nan")).cuda())
    assert np.isfinite(float(l3))


def _spin_setup(precision="fp32", N=40, Nc=64, Nf=32, gain=1.0):
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    H, W, focal, near, far = 20, 24, 30.0, 2.0, 6.0
    sd_c = O.init_nerf_params(seed=3, gain=gain) if gain != 1.0 else O.init_nerf_params(seed=3)
    sd_f = O.init_nerf_params(seed=4, gain=gain) if gain != 1.0 else O.init_nerf_params(seed=4)

    def mk(sd):
        n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision=precision).cuda()
        n.load_state_dict(sd)
        return n
    net_c, net_f = mk(sd_c), mk(sd_f)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=1.0, ndc=False, lindisp=False, near=near, far=far)
    tr = train.RenderTrainer(kw, lrate=5e-4)
    g = torch.Generator().manual_seed(0)
    c2w = torch.eye(4)[:3, :4].clone(); c2w[2, 3] = 4.0
    ro, rd = O.get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)

    def batch():
        sel = torch.randperm(H * W, generator=g)[:N]
        return torch.stack([ro[sel], rd[sel]], 0)
    rays = [batch(), batch(), batch()]
    t_clf, t_all = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
    d_inp = torch.rand(N, generator=g) * 0.3 + 0.1

    def rnd(seed):
        gg = torch.Generator().manual_seed(seed)
        return {"t_rand": torch.rand(N, Nc, generator=gg), "u": torch.rand(N, Nf, generator=gg),
                "noise_c": torch.randn(N, Nc, generator=gg), "noise_f": torch.randn(N, Nc + Nf, generator=gg)}
    rnds = [rnd(1), rnd(2), rnd(3)]
    return dict(S=S, tr=tr, nets=(net_c, net_f), sd=(sd_c, sd_f), hwf=(H, W, focal), near=near, far=far, Nc=Nc, Nf=Nf, N=N,
                rays=rays, t_clf=t_clf, t_all=t_all, d_inp=d_inp, rnds=rnds)


def test_direct_spin_iteration_matches_oracle_and_the_autograd_route(monkeypatch):
    """Round 5: the iteration on the step's library route (RenderTrainer._spin_direct: one render of the concatenated rays
    with the loss as three terms, one backward launch sequence, no torch autograd) — loss and parameter gradients against the
    oracle's three renders (the gates of the autograd-route test above), against the autograd route itself, and the NaN guard
    of the geometry term (run_nerf.py:1518-1521)."""
    c = _spin_setup()
    tr, (net_c, net_f), (sd_c, sd_f) = c["tr"], c["nets"], c["sd"]
    H, W, focal = c["hwf"]
    cu = lambda t: t.cuda()
    rays, rnds = c["rays"], c["rnds"]
    curnd = [{k: cu(v) for k, v in r.items()} for r in rnds]
    args = (H, W, focal, cu(rays[0]), cu(c["t_clf"]), cu(rays[1]), cu(c["t_all"]), cu(rays[2]), cu(c["d_inp"]))

    # ---- oracle ----
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    okw = dict(sd_coarse=pc, sd_fine=pf, N_samples=c["Nc"], N_importance=c["Nf"], perturb=1.0, white_bkgd=False, lindisp=False,
               use_viewdirs=True, ndc=False, near=c["near"], far=c["far"], retraw=True)
    rgb, disp, acc, depth, ex = O.render(H, W, focal, rays=rays[0], randoms=rnds[0], **okw)
    rgb_c, _, _, _, ex_c = O.render(H, W, focal, rays=rays[1], randoms=rnds[1], detach_weights=True, **okw)
    _, disp_i, _, _, ex_i = O.render(H, W, focal, rays=rays[2], randoms=rnds[2], **okw)
    ref = (O.img2mse(rgb, c["t_clf"]) + O.img2mse(rgb_c, c["t_all"]) + O.img2mse(ex_c["rgb0"], c["t_all"])
           + O.img2mse(ex["rgb0"], c["t_clf"]) + O.img2mse(disp_i, c["d_inp"]) + O.img2mse(ex_i["disp0"], c["d_inp"]))
    ref.backward()
    ref_psnr = -10.0 * float(torch.log10(O.img2mse(rgb, c["t_clf"]).detach()))

    # ---- the direct route (taken by default) ----
    p0 = [n.flat.detach().clone() for n in (net_c, net_f)]
    assert tr._spin_direct_ok(args, dict(randoms=curnd))
    loss, psnr = tr.spin_iteration(*args, randoms=curnd)
    assert abs(float(loss) - float(ref.detach())) < 2e-4 * abs(float(ref.detach())), (float(loss), float(ref.detach()))
    assert abs(float(psnr) - ref_psnr) < 1e-3
    assert tr.global_step == 1
    g_direct = [n.flat.grad.clone() for n in (net_c, net_f)]
    for net, p in ((net_c, pc), (net_f, pf)):
        got = net.named_views(net.flat.grad)
        for k, v in p.items():
            a, b = got[k].cpu().double().reshape(-1), v.grad.double().reshape(-1)
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert rel < 5e-3, f"{k}: relative L2 error {rel:.2e}"
    # every map the three renders return: the concatenated render's rows
    h = tr._last_spin
    N = c["N"]
    assert float((h.rgb[:N].cpu() - rgb.detach()).abs().max()) < 2e-4
    assert float((h.rgb[N:2 * N].cpu() - rgb_c.detach()).abs().max()) < 2e-4
    assert float(((h.disp[2 * N:].cpu() - disp_i.detach()) / disp_i.detach()).abs().max()) < 2e-3

    # ---- the autograd route on the same parameters and draws ----
    def restore():
        for n, p in zip((net_c, net_f), p0):
            with torch.no_grad():
                n.flat.copy_(p)
            n.mark_weights_changed()
            n.flat.grad = None
    restore()
    monkeypatch.setenv("SNR_NO_DIRECT_SPIN", "1")
    assert not tr._spin_direct_ok(args, dict(randoms=curnd))
    loss_a, psnr_a = tr.spin_iteration(*args, randoms=curnd)
    monkeypatch.delenv("SNR_NO_DIRECT_SPIN")
    assert abs(float(loss_a) - float(loss)) < 2e-6 * abs(float(loss)), (float(loss_a), float(loss))
    assert abs(float(psnr_a) - float(psnr)) < 1e-4
    for n, gd in zip((net_c, net_f), g_direct):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 2e-5

    # ---- NaN guard: a NaN among the disparity targets drops the geometry term and its gradient, nothing else ----
    restore()
    bad = c["d_inp"].clone(); bad[3] = float("nan")
    args_bad = args[:8] + (cu(bad),)
    loss_n, _ = tr.spin_iteration(*args_bad, randoms=curnd)
    ref_ab = float((O.img2mse(rgb, c["t_clf"]) + O.img2mse(rgb_c, c["t_all"]) + O.img2mse(ex_c["rgb0"], c["t_all"])
                    + O.img2mse(ex["rgb0"], c["t_clf"])).detach())
    assert np.isfinite(float(loss_n)) and abs(float(loss_n) - ref_ab) < 2e-4 * ref_ab
    g_n = [n.flat.grad.clone() for n in (net_c, net_f)]
    assert all(bool(torch.isfinite(g).all()) for g in g_n)
    restore()
    monkeypatch.setenv("SNR_NO_DIRECT_SPIN", "1")
    tr.spin_iteration(*args_bad, randoms=curnd)
    monkeypatch.delenv("SNR_NO_DIRECT_SPIN")
    for n, gd in zip((net_c, net_f), g_n):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 2e-5
    # ... and with in-kernel draws the iteration simply runs
    l2, psnr2 = tr.spin_iteration(*args)
    assert np.isfinite(float(l2)) and np.isfinite(float(psnr2))


@pytest.mark.parametrize("N", [1, 37, 130])
def test_direct_spin_iteration_with_term_boundaries_inside_workgroups(N):
    """The loss terms' ray ranges need not be multiples of the compositing kernels' 4 rays per workgroup: with N rays per
    render the boundaries at N and 2 N fall inside a workgroup (its rays then add to different terms' slots).  Direct route
    against the autograd route, fp32, injected draws."""
    import os
    c = _spin_setup(N=N)
    tr, (net_c, net_f) = c["tr"], c["nets"]
    H, W, focal = c["hwf"]
    cu = lambda t: t.cuda()
    curnd = [{k: cu(v) for k, v in r.items()} for r in c["rnds"]]
    args = (H, W, focal, cu(c["rays"][0]), cu(c["t_clf"]), cu(c["rays"][1]), cu(c["t_all"]), cu(c["rays"][2]), cu(c["d_inp"]))
    p0 = [n.flat.detach().clone() for n in (net_c, net_f)]
    loss, psnr = tr.spin_iteration(*args, randoms=curnd)
    g_direct = [n.flat.grad.clone() for n in (net_c, net_f)]
    for n, p in zip((net_c, net_f), p0):
        with torch.no_grad():
            n.flat.copy_(p)
        n.mark_weights_changed()
        n.flat.grad = None
    os.environ["SNR_NO_DIRECT_SPIN"] = "1"
    try:
        loss_a, psnr_a = tr.spin_iteration(*args, randoms=curnd)
    finally:
        del os.environ["SNR_NO_DIRECT_SPIN"]
    assert abs(float(loss_a) - float(loss)) < 3e-6 * abs(float(loss)) and abs(float(psnr_a) - float(psnr)) < 1e-4
    for n, gd in zip((net_c, net_f), g_direct):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 3e-5


def test_direct_spin_iteration_bf16_runs_on_the_merged_backward():
    """bf16 (the bench's precision): the direct iteration's gradients against the autograd route's, which runs the three
    renders one after the other; the two differ only by the split-K partition of the bf16 partial sums (documented 6e-3)."""
    import os
    c = _spin_setup(precision="bf16", N=64)
    tr, (net_c, net_f) = c["tr"], c["nets"]
    H, W, focal = c["hwf"]
    cu = lambda t: t.cuda()
    curnd = [{k: cu(v) for k, v in r.items()} for r in c["rnds"]]
    args = (H, W, focal, cu(c["rays"][0]), cu(c["t_clf"]), cu(c["rays"][1]), cu(c["t_all"]), cu(c["rays"][2]), cu(c["d_inp"]))
    p0 = [n.flat.detach().clone() for n in (net_c, net_f)]
    loss, _ = tr.spin_iteration(*args, randoms=curnd)
    g_direct = [n.flat.grad.clone() for n in (net_c, net_f)]
    for n, p in zip((net_c, net_f), p0):
        with torch.no_grad():
            n.flat.copy_(p)
        n.mark_weights_changed()
        n.flat.grad = None
    os.environ["SNR_NO_DIRECT_SPIN"] = "1"
    try:
        loss_a, _ = tr.spin_iteration(*args, randoms=curnd)
    finally:
        del os.environ["SNR_NO_DIRECT_SPIN"]
    assert abs(float(loss_a) - float(loss)) < 1e-5 * abs(float(loss))
    for n, gd in zip((net_c, net_f), g_direct):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 6e-3


def test_colmap_depth_render_and_prepare_export_and_lpips_hookup(tmp_path, monkeypatch):
    """The rest of the iteration (run_nerf.py:1473-1507, 1523-1561, 1563-1609): the render with the COLMAP `depths=` column
    and its depth loss against the oracle (loss value and gradients), the perceptual term's patch renders with a stand-in
    distance, and the --prepare disparity export."""
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    H, W, focal, near, far = 20, 24, 30.0, 2.0, 6.0
    Nc, Nf, N = 64, 32, 40
    sd_c, sd_f = O.init_nerf_params(seed=3, gain=2.0), O.init_nerf_params(seed=4, gain=2.0)

    def mk(sd):
        n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="fp32").cuda()
        n.load_state_dict(sd)
        return n
    net_c, net_f = mk(sd_c), mk(sd_f)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=False, raw_noise_std=1.0, ndc=False, lindisp=False, near=near, far=far)
    tr = train.RenderTrainer(kw, lrate=5e-4)
    g = torch.Generator().manual_seed(0)
    c2w = torch.eye(4)[:3, :4].clone(); c2w[2, 3] = 4.0
    ro, rd = O.get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)

    def batch():
        sel = torch.randperm(H * W, generator=g)[:N]
        return torch.stack([ro[sel], rd[sel]], 0)
    rays_clf, rays_all, rays_dep = batch(), batch(), batch()
    t_clf, t_all = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
    t_dep = torch.rand(N, generator=g) * 2 + 3
    wts = torch.rand(N, generator=g)

    def rnd(seed):
        gg = torch.Generator().manual_seed(seed)
        return {"t_rand": torch.rand(N, Nc, generator=gg), "u": torch.rand(N, Nf, generator=gg),
                "noise_c": torch.randn(N, Nc, generator=gg), "noise_f": torch.randn(N, Nc + Nf, generator=gg)}
    rnds = [rnd(1), rnd(2), None, rnd(4)]
    cu = lambda t: t.cuda()
    for mode, ref_term in (("weighted", lambda d: torch.mean(((d - t_dep) ** 2) * wts)),
                           ("relative", lambda d: torch.mean(((d - t_dep) / t_dep) ** 2)),
                           ("mse", lambda d: O.img2mse(d, t_dep))):
        pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
        pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
        okw = dict(sd_coarse=pc, sd_fine=pf, N_samples=Nc, N_importance=Nf, perturb=1.0, white_bkgd=False, lindisp=False,
                   use_viewdirs=True, ndc=False, near=near, far=far, retraw=True)
        rgb, _, _, _, ex = O.render(H, W, focal, rays=rays_clf, randoms=rnds[0], **okw)
        rgb_c, _, _, _, ex_c = O.render(H, W, focal, rays=rays_all, randoms=rnds[1], detach_weights=True, **okw)
        _, _, _, depth_col, _ = O.render(H, W, focal, rays=rays_dep, depths=t_dep, randoms=rnds[3], **okw)
        ref = (O.img2mse(rgb, t_clf) + O.img2mse(rgb_c, t_all) + O.img2mse(ex_c["rgb0"], t_all) + O.img2mse(ex["rgb0"], t_clf)
               + 0.1 * ref_term(depth_col))
        ref.backward()
        for n in (net_c, net_f):
            n.flat.grad = None
        loss, outs = tr.spin_loss(H, W, focal, cu(rays_clf), cu(t_clf), cu(rays_all), cu(t_all),
                                  randoms=[{k: cu(v) for k, v in r.items()} if r else None for r in rnds],
                                  colmap_depth=dict(rays=cu(rays_dep), target=cu(t_dep), weights=cu(wts), depth_lambda=0.1,
                                                    mode=mode))
        assert abs(float(loss.detach()) - float(ref.detach())) < 5e-4 * abs(float(ref.detach())), (mode, float(loss.detach()), float(ref.detach()))
        assert tuple(outs["colmap"][0].shape) == (N,)
        loss.backward()
        for net, p in ((net_c, pc), (net_f, pf)):
            got = net.named_views(net.flat.grad)
            for k, v in p.items():
                a, b = got[k].cpu().double().reshape(-1), v.grad.double().reshape(-1)
                rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
                assert rel < 1e-2, f"{mode} {k}: relative L2 error {rel:.2e}"

    # ---- perceptual term: patch renders with gradients, a stand-in distance (mean squared difference per image) ----
    kw_test = dict(kw, perturb=False, raw_noise_std=0.)
    poses = torch.stack([c2w, c2w + torch.tensor([[0, 0, 0, 0.2], [0, 0, 0, 0], [0, 0, 0, 0]])]).cuda()
    images = torch.rand(2, H, W, 3, generator=g).cuda()
    masks = np.zeros((2, H, W)); masks[:, 4:16, 5:20] = 1
    calls = []

    def dist_fn(pred, target):
        assert pred.shape == target.shape == (1, 3, H // 4, W // 4) and pred.requires_grad
        assert float(pred.detach().abs().max()) <= 1.0 + 1e-5 and float(target.abs().max()) <= 1.0 + 1e-5
        calls.append(1)
        return ((pred - target) ** 2).mean(dim=(1, 2, 3))
    # The patch corner is drawn with random.randint inside the mask's bounding box (run_nerf.py:197-209): X in 4..10, Y in 5..13.
    # These untrained gain-2 networks have a density <= 0 on 433 of the 480 pixels (every fine weight exactly 0 there), so 18 of
    # the 63 corners of pose 0 and 5 of pose 1 give an ALL-EMPTY 5 x 6 patch — whose perceptual term has, correctly, a gradient
    # of exactly zero.  Rounds 1-5 drew the corner from the process's unseeded `random` state and asserted "gradient > 0": with
    # both poses' patches empty (18/63 x 5/63 = 2.3 % of the draws) the assertion failed — THE intermittent failure of the full
    # suite (round 5: once in ~25 runs; round 6: pass 18 of a 24-pass soak, full output in profiles/r06_soak.txt; the 40 seeds
    # round 5 tried had a 40 % chance of missing it).  A defect of the test, not of a kernel.  Now every corner is visited, for
    # both poses at once, and the property that holds is asserted: the gradient is non-zero exactly when a patch is not empty.
    import random
    n_empty = n_lit = 0

    def acc_sum(pose, X, Y):
        with torch.no_grad():   # acc_map = sum of the (non-negative) weights of the fine pass
            return float(S.render(H, W, focal, chunk=1024, c2w=pose[:3, :4], patch=(X, Y, H // 4, W // 4), **kw_test)[2].abs().sum())
    corners = [(X, Y) for X in range(4, 11) for Y in range(5, 14)]
    # pose 0 sweeps every corner while pose 1 sits on one of ITS empty corners, then the other way round (the two poses' empty
    # corners do not overlap: CPU oracle, 18 and 5 of 63)
    fixed = {1: next(c for c in corners if acc_sum(poses[1], *c) == 0.0), 0: next(c for c in corners if acc_sum(poses[0], *c) == 0.0)}
    for moving in (0, 1):
        for c in corners:
            pair = (c, fixed[1]) if moving == 0 else (fixed[0], c)
            draws = iter([pair[0][0], pair[0][1], pair[1][0], pair[1][1]])     # render_path draws X then Y, pose by pose
            monkeypatch.setattr(random, "randint", lambda a, b: next(draws))
            lit = acc_sum(poses[moving], *c) > 0.0
            for n in (net_c, net_f):
                n.flat.grad = None
            calls.clear()
            term = tr.lpips_term(dist_fn, poses, images, masks, (H, W, focal), kw_test)
            assert len(calls) == 2 and 0 < float(term.detach()) < 4.0 / 100 + 1e-6
            term.backward()
            got = float(net_f.flat.grad.abs().max()) > 0
            assert got == lit, (f"pose {moving} at corner {c} (the other pose on an empty patch): patch {'not ' if lit else ''}empty, "
                                f"fine-network gradient {'non-' if got else ''}zero")
            n_empty += not lit
            n_lit += lit
    monkeypatch.undo()
    assert n_empty >= 10 and n_lit >= 60, (n_empty, n_lit)   # the scene has both kinds (CPU oracle: 18 + 5 empty, 45 + 58 lit)

    # ---- --prepare: disparity maps and masks for the depth inpainter ----
    tr.export_disparities(poses, (H, W, focal), kw_test, masks, str(tmp_path / "prep"), render_factor=2)
    import os
    assert sorted(os.listdir(tmp_path / "prep")) == ["img000.png", "img001.png", "label"]
    assert sorted(os.listdir(tmp_path / "prep" / "label")) == ["img000.png", "img001.png"]
    assert open(tmp_path / "prep" / "img000.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"


@pytest.mark.parametrize("Nc,Nf", [(96, 32), (65, 16)])
def test_library_routes_with_coarse_sample_counts_the_fused_compositing_kernel_does_not_cover(monkeypatch, Nc, Nf):
    """the coarse compositing + hierarchical-sampling kernel holds a ray's coarse samples in one 64-lane
    wave (3 <= N_samples <= 64); beyond that it must DECLINE (SNR_ERR_UNSUPPORTED) so that the library route falls back to its
    two-kernel form, not fail the step with SNR_ERR_SHAPE.  RenderTrainer.step and spin_iteration with N_samples = 96 (and 65)
    and a fine pass (the reference itself cannot run N_samples = 2 with a fine pass: sample_pdf indexes an empty bin list): loss and parameter gradients of the library route against the render() + autograd route (itself held to
    the oracle above), and the step's loss against the oracle's three renders."""
    c = _spin_setup(Nc=Nc, Nf=Nf, N=24)
    tr, (net_c, net_f), (sd_c, sd_f) = c["tr"], c["nets"], c["sd"]
    H, W, focal = c["hwf"]
    cu = lambda t: t.cuda()
    rays, rnds = c["rays"], c["rnds"]
    curnd = [{k: cu(v) for k, v in r.items()} for r in rnds]
    args = (H, W, focal, cu(rays[0]), cu(c["t_clf"]), cu(rays[1]), cu(c["t_all"]), cu(rays[2]), cu(c["d_inp"]))
    p0 = [n.flat.detach().clone() for n in (net_c, net_f)]

    def restore():
        for n, p in zip((net_c, net_f), p0):
            with torch.no_grad():
                n.flat.copy_(p)
            n.mark_weights_changed()
            n.flat.grad = None

    # ---- the SPIn-NeRF iteration: direct route vs autograd route vs oracle ----
    assert tr._spin_direct_ok(args, dict(randoms=curnd))
    loss, psnr = tr.spin_iteration(*args, randoms=curnd)
    g_direct = [n.flat.grad.clone() for n in (net_c, net_f)]
    assert all(bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0 for g in g_direct)
    okw = dict(sd_coarse=sd_c, sd_fine=sd_f, N_samples=Nc, N_importance=Nf, perturb=1.0, white_bkgd=False, lindisp=False,
               use_viewdirs=True, ndc=False, near=c["near"], far=c["far"], retraw=True)
    with torch.no_grad():
        rgb, _, _, _, ex = O.render(H, W, focal, rays=rays[0], randoms=rnds[0], **okw)
        rgb_c, _, _, _, ex_c = O.render(H, W, focal, rays=rays[1], randoms=rnds[1], detach_weights=True, **okw)
        _, disp_i, _, _, ex_i = O.render(H, W, focal, rays=rays[2], randoms=rnds[2], **okw)
        ref = (O.img2mse(rgb, c["t_clf"]) + O.img2mse(rgb_c, c["t_all"]) + O.img2mse(ex_c["rgb0"], c["t_all"])
               + O.img2mse(ex["rgb0"], c["t_clf"]) + O.img2mse(disp_i, c["d_inp"]) + O.img2mse(ex_i["disp0"], c["d_inp"]))
    assert abs(float(loss) - float(ref)) < 2e-4 * abs(float(ref)), (float(loss), float(ref))
    restore()
    monkeypatch.setenv("SNR_NO_DIRECT_SPIN", "1")
    loss_a, psnr_a = tr.spin_iteration(*args, randoms=curnd)
    monkeypatch.delenv("SNR_NO_DIRECT_SPIN")
    assert abs(float(loss_a) - float(loss)) < 2e-6 * abs(float(loss)), (float(loss_a), float(loss))
    for n, gd in zip((net_c, net_f), g_direct):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 2e-5

    # ---- the plain step: library route vs autograd route ----
    restore()
    l1, rgb1 = tr.step(H, W, focal, args[3], args[4], randoms=curnd[0])
    g1 = [n.flat.grad.clone() for n in (net_c, net_f)]
    restore()
    monkeypatch.setenv("SNR_NO_DIRECT_STEP", "1")
    l2, rgb2 = tr.step(H, W, focal, args[3], args[4], randoms=curnd[0])
    monkeypatch.delenv("SNR_NO_DIRECT_STEP")
    assert abs(float(l1) - float(l2)) < 2e-6 * abs(float(l2)), (float(l1), float(l2))
    assert float((rgb1 - rgb2).abs().max()) < 1e-6
    for n, gd in zip((net_c, net_f), g1):
        assert float((n.flat.grad - gd).norm() / gd.norm()) < 2e-5
