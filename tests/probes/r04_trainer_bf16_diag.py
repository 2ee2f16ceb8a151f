"""numbers behind tests/test_gpu_train_step.py::test_bf16_trainer_step_matches_the_oracle_with_bf16_rounding_emulation on the
regenerated (seeded) trained fixtures: per-ray rgb differences, loss, moments"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nerf_oracle as O
from helpers import load, render_case_nets, TRAINED_CASES
import spin_nerf_amd as S
train = importlib.import_module("spin-nerf_amd.train")
for name in TRAINED_CASES:
    g = load(name)
    sd_c, sd_f = render_case_nets(g)
    H, W, focal, near, far = int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), float(g["far"])
    Nc, Nf = 64, 128
    rays = torch.from_numpy(g["rays"]); target = torch.from_numpy(g["target"]); N = rays.shape[1]
    white, std = bool(g["white"]), float(g["noise_std"])
    def mk(sd):
        n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda(); n.load_state_dict(sd); return n
    net_c, net_f = mk(sd_c), mk(sd_f)
    def q(inputs, viewdirs, network_fn): return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=net_f, N_samples=Nc, network_fn=net_c,
              use_viewdirs=True, white_bkgd=white, raw_noise_std=std, ndc=False, lindisp=False, near=near, far=far)
    tr = train.RenderTrainer(kw, lrate=5e-4, lrate_decay=250)
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    opt = O.AdamState(list(pc.values()) + list(pf.values()), lr=5e-4)
    okw = dict(H=H, W=W, focal=focal, chunk=1024 * 32, ndc=False, near=near, far=far, use_viewdirs=True, N_samples=Nc,
               N_importance=Nf, perturb=1.0, white_bkgd=white, lindisp=False, mlp=O.nerf_forward_bf16emu)
    for seed in (9, 10, 11):
        gen = torch.Generator().manual_seed(seed)
        rnd = dict(t_rand=torch.rand(N, Nc, generator=gen), u=torch.rand(N, Nf, generator=gen),
                   noise_c=torch.randn(N, Nc, generator=gen) * std, noise_f=torch.randn(N, Nc + Nf, generator=gen) * std)
        pc2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd_c.items()}
        pf2 = {k: v.detach().clone().requires_grad_(True) for k, v in sd_f.items()}
        opt2 = O.AdamState(list(pc2.values()) + list(pf2.values()), lr=5e-4)
        ref_loss, ref_rgb = O.train_step(pc2, pf2, opt2, rays, target, okw, randoms=rnd)
        nc2, nf2 = mk(sd_c), mk(sd_f)
        kw2 = dict(kw, network_fn=nc2, network_fine=nf2)
        tr2 = train.RenderTrainer(kw2, lrate=5e-4, lrate_decay=250)
        loss, rgb = tr2.step(H, W, focal, rays.cuda(), target.cuda(), randoms={k: v.cuda() for k, v in rnd.items()})
        d = (rgb.cpu() - ref_rgb).abs().max(-1).values
        srt = torch.sort(d, descending=True).values
        rel = []
        for i, m in enumerate(tr2.m):
            ref_m = torch.cat([mm.reshape(-1) for mm in (opt2.m[:len(pc2)] if i == 0 else opt2.m[len(pc2):])])
            rel.append(float((m.cpu() - ref_m).norm() / ref_m.norm()))
        print(name, "seed", seed, "loss rel", abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)), "rgb per-ray top5", [round(float(x), 5) for x in srt[:5]],
              "median", float(srt[len(srt) // 2]), "moments rel", rel)
