"""CPU: RenderTrainer checkpoints use the reference's layout (run_nerf.py:1626-1636): a plain torch.optim.Adam over
per-layer tensors in grad_vars order accepts the saved optimizer state and continues exactly like the trainer's
own Adam does; round trip through save/load restores everything."""
import importlib

import torch


def _cpu_adam(params, grads, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    g = grads * grad_scale
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    params.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt() + eps, value=-lr)


def test_checkpoint_is_loadable_by_a_per_layer_adam(tmp_path, monkeypatch):
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    monkeypatch.setattr(train.ops, "adam_step_", _cpu_adam)
    torch.manual_seed(0)
    nets = [S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True) for _ in range(2)]
    tr = train.RenderTrainer({"network_fn": nets[0], "network_fine": nets[1]}, lrate=1e-3)
    g = torch.Generator().manual_seed(1)

    def set_grads():
        for n in nets:
            n.flat.grad = torch.randn(n.flat.shape, generator=g) * 1e-2
    for _ in range(3):
        set_grads(); tr.apply_gradients()
    path = str(tmp_path / "000003.tar")
    tr.save_checkpoint(path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    assert ck["global_step"] == 3 and list(ck["network_fn_state_dict"])[0] == "pts_linears.0.weight"

    # the reference side: per-layer parameters (coarse, then fine), torch.optim.Adam, load_state_dict
    params = [torch.nn.Parameter(v.clone()) for sd in (ck["network_fn_state_dict"], ck["network_fine_state_dict"])
              for v in sd.values()]
    opt = torch.optim.Adam(params=params, lr=1e-3, betas=(0.9, 0.999))
    opt.load_state_dict(ck["optimizer_state_dict"])
    # one more step on both sides with the same gradients
    set_grads()
    o = 0
    for n in nets:
        for view in n.named_views(n.flat.grad).values():
            params[o].grad = view.clone(); o += 1
    for gr in opt.param_groups:
        gr["lr"] = tr.lrate * (0.1 ** (4 / (tr.lrate_decay * 1000)))   # the trainer's step 4 uses the decayed rate
    opt.step()
    tr.apply_gradients()
    o = 0
    for n in nets:
        for view in n.named_views().values():
            assert torch.allclose(view, params[o], atol=1e-7, rtol=1e-5); o += 1

    # round trip
    tr2 = train.RenderTrainer({"network_fn": S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True),
                               "network_fine": S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)}, lrate=1e-3)
    tr2.load_checkpoint(path)
    assert tr2.global_step == 3
    ck = torch.load(path, weights_only=False)   # (the optimizer above stepped on the first copy's tensors)
    ck2 = tr2.state_dict()
    for a, b in zip(ck["optimizer_state_dict"]["state"].values(), ck2["optimizer_state_dict"]["state"].values()):
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
    for k in ck["network_fine_state_dict"]:
        assert torch.equal(ck["network_fine_state_dict"][k], ck2["network_fine_state_dict"][k])


def test_resume_through_create_nerf_continues_like_an_uninterrupted_run(tmp_path, monkeypatch):
    """create_nerf reloads a reference-format checkpoint (run_nerf.py:448-462); RenderTrainer(optimizer=, start=) adopts
    the Adam moments, the step count and the decayed learning rate, so the next steps equal those of a run that was
    never interrupted.  A checkpoint whose optimizer state does not fit raises instead of restarting silently."""
    import argparse
    import pytest
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    monkeypatch.setattr(train.ops, "adam_step_", _cpu_adam)
    (tmp_path / "run").mkdir()
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=128,
                              N_samples=64, alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8,
                              netwidth_fine=256, netchunk=65536, lrate=1e-3, basedir=str(tmp_path), expname="run",
                              ft_path=None, no_reload=False, perturb=1.0, white_bkgd=True, raw_noise_std=1.0,
                              dataset_type="llff", no_ndc=True, lindisp=True, sigma_loss=False, no_coarse=False)
    cpu = torch.device("cpu")
    torch.manual_seed(3)
    kw, _, start, _, opt = S.create_nerf(args, device=cpu)
    assert start == 0
    tr = train.RenderTrainer(kw, lrate=1e-3, lrate_decay=1, optimizer=opt, start=start)   # fast decay: visible in 6 steps
    g = torch.Generator().manual_seed(1)
    grads = [[torch.randn(n.flat.shape, generator=g) * 1e-2 for n in tr.nets] for _ in range(6)]

    def run(t, steps):
        for gs in steps:
            for n, gr in zip(t.nets, gs):
                n.flat.grad = gr.clone()
            t.apply_gradients()
    run(tr, grads[:3])
    assert tr.global_step == 3 and tr.opt_step == 3
    # the reference's schedule (run_nerf.py:1611-1622, 1703): steps 1, 2 at lrate, step k at lrate * 0.1 ** ((k - 2) / 1000)
    assert abs(tr.current_lr() - 1e-3 * 0.1 ** (2 / 1000)) < 1e-12
    tr.save_checkpoint(str(tmp_path / "run" / "000003.tar"))
    run(tr, grads[3:])                                       # the uninterrupted run

    kw2, _, start2, _, opt2 = S.create_nerf(args, device=cpu)   # picks the checkpoint up
    assert start2 == 3
    tr2 = train.RenderTrainer(kw2, lrate=1e-3, lrate_decay=1, optimizer=opt2, start=start2)
    assert tr2.opt_step == 3 and tr2.global_step == 3 and abs(tr2.current_lr() - 1e-3 * 0.1 ** (2 / 1000)) < 1e-12
    run(tr2, grads[3:])
    for a, b in zip(tr.nets, tr2.nets):
        assert torch.equal(a.flat.detach(), b.flat.detach())
    for a, b in zip(tr.m + tr.v, tr2.m + tr2.v):
        assert torch.equal(a, b)

    # a state that does not cover these networks' parameters is an error, not a silent restart
    ck = torch.load(tmp_path / "run" / "000003.tar", weights_only=False)
    ck["optimizer_state_dict"]["state"].pop(5)
    torch.save(ck, tmp_path / "run" / "000004.tar")
    with pytest.raises(RuntimeError):
        S.create_nerf(args, device=cpu)


def test_nerf_rgb_draws_and_registers_what_the_reference_does():
    """helpers:159-191: NeRF_RGB never constructs alpha_linear — same seed, same weights as the reference for everything
    built after it, and no alpha_linear entries among the parameters Adam state is written for."""
    import spin_nerf_amd as S
    torch.manual_seed(11)
    a = S.NeRF_RGB(input_ch=63, input_ch_views=27, use_viewdirs=True)
    after_a = torch.rand(1)
    torch.manual_seed(11)
    for fin, fout in [(63, 256)] + [(256, 256)] * 4 + [(319, 256)] + [(256, 256)] * 2 + [(283, 128), (256, 256), (128, 3)]:
        torch.nn.Linear(fin, fout)          # the reference's construction order without alpha_linear
    assert torch.equal(after_a, torch.rand(1))
    v = a.named_views(a.flat.detach())
    assert float(v["alpha_linear.weight"].abs().max()) == 0.0 and float(v["alpha_linear.bias"].abs().max()) == 0.0
    assert not any(k.startswith("alpha_linear") for k in a.param_views()) and len(a.param_views()) == 22
    assert len(S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True).param_views()) == 24


def test_resume_from_a_per_layer_adam_without_state_for_the_unused_views_layer(monkeypatch):
    """with use_viewdirs=False the reference still registers views_linears.0 (helpers:86-90) but never gives it
    a gradient, so a genuine torch.optim.Adam state has no entries for those two tensors per network — its indices have
    holes.  Such a checkpoint resumes (zero moments for the idle layer); a hole anywhere else still raises."""
    import pytest
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    monkeypatch.setattr(train.ops, "adam_step_", _cpu_adam)
    torch.manual_seed(0)
    nets = [S.NeRF(input_ch=63, input_ch_views=0, output_ch=5, use_viewdirs=False) for _ in range(2)]
    # the reference side: per-layer tensors in state-dict order, gradients for every layer but views_linears.*
    names = [k for n in nets for k in n.named_views()]
    params = [torch.nn.Parameter(v.clone()) for n in nets for v in n.named_views().values()]
    opt = torch.optim.Adam(params=params, lr=1e-3, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(3)
    for _ in range(2):
        for k, p in zip(names, params):
            p.grad = None if k.startswith("views_linears") else torch.randn(p.shape, generator=g) * 1e-2
        opt.step()
    osd = opt.state_dict()
    assert len(osd["state"]) == len(params) - 4 and 16 not in osd["state"]      # two idle tensors per network: holes
    ck = {"global_step": 2, "optimizer_state_dict": osd,
          "network_fn_state_dict": {k: p.detach() for k, p in zip(names[:len(names) // 2], params[:len(params) // 2])},
          "network_fine_state_dict": {k: p.detach() for k, p in zip(names[len(names) // 2:], params[len(params) // 2:])}}
    tr = train.RenderTrainer({"network_fn": nets[0], "network_fine": nets[1]}, lrate=1e-3)
    tr.load_state_dict(ck)
    assert tr.opt_step == 2
    o = 0
    for n, m, v in zip(tr.nets, tr.m, tr.v):
        for (k, mv), vv in zip(n.param_views(m).items(), n.param_views(v).values()):
            e = osd["state"].get(o)
            if e is None:
                assert k.startswith("views_linears") and float(mv.abs().max()) == 0.0 and float(vv.abs().max()) == 0.0
            else:
                assert torch.equal(mv, e["exp_avg"].reshape(mv.shape)) and torch.equal(vv, e["exp_avg_sq"].reshape(vv.shape))
            o += 1
    # a hole in a layer that does train is refused
    bad = {"state": {k: v for k, v in osd["state"].items() if k != 3}, "param_groups": osd["param_groups"]}
    with pytest.raises(RuntimeError, match="no Adam state for parameter 3"):
        tr.load_state_dict(dict(ck, optimizer_state_dict=bad))
    # ... and so is state for a tensor these networks do not have
    st_extra = dict(osd["state"]); st_extra[len(params)] = osd["state"][0]
    extra = {"state": st_extra, "param_groups": osd["param_groups"]}
    with pytest.raises(RuntimeError, match="parameter indices"):
        tr.load_state_dict(dict(ck, optimizer_state_dict=extra))


def test_create_nerf_refuses_an_unsupported_network_shape_by_flag_name(tmp_path):
    """run_nerf.py:393-421 builds NeRF(D=args.netdepth, W=args.netwidth, ...); the HIP kernels implement the defaults (8 x 256,
    skips [4]).  Anything else is refused by create_nerf with the FLAG's name, before any allocation or launch."""
    import argparse
    import pytest
    import spin_nerf_amd as S
    (tmp_path / "run").mkdir()
    base = dict(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=128, N_samples=64,
                alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536,
                lrate=1e-3, basedir=str(tmp_path), expname="run", ft_path=None, no_reload=True, perturb=1.0,
                white_bkgd=True, raw_noise_std=1.0, dataset_type="llff", no_ndc=True, lindisp=True, sigma_loss=False,
                no_coarse=False)
    for flag, value in (("netwidth", 128), ("netdepth", 6), ("netwidth_fine", 512), ("netdepth_fine", 4), ("multires", 12)):
        with pytest.raises(NotImplementedError, match=f"--{flag} {value}"):
            S.create_nerf(argparse.Namespace(**dict(base, **{flag: value})), device=torch.device("cpu"))
    S.create_nerf(argparse.Namespace(**base), device=torch.device("cpu"))   # the defaults build


def test_load_weights_from_keras_transposes_into_the_flat_buffer():
    """DS_NeRF/run_nerf_helpers.py:129-156: [kernel [in, out], bias] pairs in the order pts_linears 0..7, feature_linear,
    views_linears.0, rgb_linear, alpha_linear."""
    import numpy as np
    import pytest
    import spin_nerf_amd as S
    net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)
    rs = np.random.RandomState(0)
    order = [f"pts_linears.{i}" for i in range(8)] + ["feature_linear", "views_linears.0", "rgb_linear", "alpha_linear"]
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    weights = []
    for name in order:
        fout, fin = shapes[name + ".weight"]
        weights += [rs.normal(size=(fin, fout)).astype(np.float32), rs.normal(size=(fout,)).astype(np.float32)]
    gen = net.weights_generation
    net.load_weights_from_keras(weights)
    sd = net.state_dict()
    for j, name in enumerate(order):
        assert np.array_equal(sd[name + ".weight"].numpy(), weights[2 * j].T)
        assert np.array_equal(sd[name + ".bias"].numpy(), weights[2 * j + 1])
    assert net.weights_generation == gen + 1       # the packed-weight cache is invalidated
    with pytest.raises(AssertionError):
        S.NeRF(input_ch=63, input_ch_views=0, use_viewdirs=False, output_ch=5).load_weights_from_keras(weights)
