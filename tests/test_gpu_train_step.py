"""GPU: RenderTrainer.step — the exact code bench.py times — against the oracle's train_step
(DS_NeRF/run_nerf.py:1465-1490 render + img2mse(rgb) + img2mse(rgb0); :1611-1612 backward + Adam; :1616-1622 lr decay)
with the same injected randoms, fp32 path, bench.py's render configuration (no_ndc + lindisp + white_bkgd + perturb=1 +
raw_noise_std=1, viewdirs, 64 coarse + 128 fine samples), three consecutive steps."""
import importlib

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def test_trainer_step_matches_oracle_train_step_for_three_steps():
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    H, W, focal, near, far = 378, 504, 400.0, 1.2, 9.0
    Nc, Nf, N = 64, 128, 96
    sd_c, sd_f = O.init_nerf_params(seed=0), O.init_nerf_params(seed=1)

    def mk(sd):
        n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="fp32").cuda()
        n.load_state_dict(sd)
        return n
    net_c, net_f = mk(sd_c), mk(sd_f)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=1.0, ndc=False, lindisp=True, near=near, far=far)
    lrate, decay = 5e-4, 250
    tr = train.RenderTrainer(kw, lrate=lrate, lrate_decay=decay)

    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    opt = O.AdamState(list(pc.values()) + list(pf.values()), lr=lrate)
    okw = dict(H=H, W=W, focal=focal, chunk=1024 * 32, ndc=False, near=near, far=far, use_viewdirs=True, N_samples=Nc,
               N_importance=Nf, perturb=1.0, white_bkgd=True, lindisp=True)

    c2w = torch.eye(4)[:3, :4]
    ro, rd = O.get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    g = torch.Generator().manual_seed(5)
    p0 = [torch.cat([v.detach().reshape(-1) for v in p.values()]).clone() for p in (pc, pf)]
    for step in range(3):
        sel = torch.randperm(H * W, generator=g)[:N]
        rays = torch.stack([ro[sel], rd[sel]], 0)
        target = torch.rand(N, 3, generator=g)
        rnd = dict(t_rand=torch.rand(N, Nc, generator=g), u=torch.rand(N, Nf, generator=g),
                   noise_c=torch.randn(N, Nc, generator=g), noise_f=torch.randn(N, Nc + Nf, generator=g))
        # run_nerf.py:1611-1622, 1703: the optimiser steps with its current rate, THEN the rate for the next step is set from
        # the not-yet-incremented global_step: steps 1 and 2 run at lrate, step k at lrate * 0.1 ** ((k - 2) / decay_steps)
        opt.lr = lrate * (0.1 ** (max(step - 1, 0) / (decay * 1000)))
        assert abs(tr.current_lr() - opt.lr) < 1e-12
        ref_loss, ref_rgb = O.train_step(pc, pf, opt, rays, target, okw, randoms=rnd)
        loss, rgb = tr.step(H, W, focal, rays.cuda(), target.cuda(), randoms={k: v.cuda() for k, v in rnd.items()})
        assert abs(float(loss) - float(ref_loss)) < 1e-3 * abs(float(ref_loss)), (step, float(loss), float(ref_loss))
        d = (rgb.cpu() - ref_rgb).abs()
        # free-running fine stage (tests/test_gpu_render.py): the bulk tight, displaced samples bounded
        assert float((d < 3e-4).float().mean()) > 0.9 and float(d.max()) < 5e-2, (step, float(d.max()))
        if step == 0:
            # after the first step Adam's moments ARE the gradients (times 0.1 / 0.001 of their squares): the gradient
            # parity of tests/test_gpu_spin_iter.py (rel-L2 5e-3) element for element
            n_c = len(pc)
            for ni, (net, p) in enumerate(((net_c, pc), (net_f, pf))):
                gm, gv = net.named_views(tr.m[ni]), net.named_views(tr.v[ni])
                for j, k in enumerate(p):
                    for name, a, b in (("exp_avg", gm[k], opt.m[ni * n_c + j]), ("exp_avg_sq", gv[k], opt.v[ni * n_c + j])):
                        a, b = a.cpu().double().reshape(-1), b.double().reshape(-1)
                        rel = float((a - b).norm() / b.norm().clamp_min(1e-300))
                        # (the fine network sees the free-running resampled z_vals: tests/test_gpu_render.py's 3e-2)
                        assert rel < (1e-2 if ni == 0 else 3e-2), f"{k}: {name} after the first step differs by {rel:.2e}"
    assert tr.global_step == 3 and opt.t == 3
    # Later steps start from parameters that already differ in the elements Adam moved the other way (it moves every
    # element by ~lr per step whatever the size of its gradient, so a gradient within rounding of zero decides the sign:
    # measured 3 % of a bias tensor's update), and ReLU units near their threshold flip — the trajectories separate at
    # a bounded rate instead of agreeing to rounding.  After three steps: weights 1e-3, biases 5e-3 relative L2, and the
    # accumulated update points the same way.
    for ni, (net, p, start) in enumerate(((net_c, pc, p0[0]), (net_f, pf, p0[1]))):
        got = net.named_views(net.flat.detach())
        for j, (k, v) in enumerate(p.items()):
            a, b = got[k].cpu().double().reshape(-1), v.detach().double().reshape(-1)
            rel = float((a - b).norm() / b.norm())
            tol = 1e-3 if v.dim() == 2 else 5e-3
            assert rel < tol, f"{k}: parameters after 3 steps differ by {rel:.2e} (relative L2)"
        a = torch.cat([got[k].cpu().double().reshape(-1) for k in p]) - start.double()
        b = torch.cat([v.detach().double().reshape(-1) for v in p.values()]) - start.double()
        cos = float((a @ b) / (a.norm() * b.norm()))
        assert cos > 0.98, f"update direction: cosine {cos:.4f}"
