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
