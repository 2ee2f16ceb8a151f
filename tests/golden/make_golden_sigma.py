"""Golden vectors for SigmaLoss (DS_NeRF/loss.py:8-44).  Run ONLY in the build container:

    python tests/golden/make_golden_sigma.py

The reference draws torch.rand / torch.randn internally; the same draws are reproduced here by re-seeding, stored
as inputs, and injected into the build's implementations by the tests."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch
from make_golden import import_reference, npz
from oracle import nerf_oracle as O

H, R = import_reference()
import loss as L   # /root/reference/DS_NeRF/loss.py (on sys.path after import_reference)

sd = O.make_wild_params(seed=41)
net = H.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict(sd)
embed_fn, _ = H.get_embedder(10, 0)
embeddirs_fn, _ = H.get_embedder(4, 0)
run = lambda inputs, viewdirs, network_fn: R.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                         embeddirs_fn=embeddirs_fn, netchunk=65536)
rs = np.random.RandomState(42)
N, S = 24, 64
ro = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(N, 3)).astype(np.float32))
rd = torch.from_numpy(rs.normal(size=(N, 3)).astype(np.float32))
vd = rd / rd.norm(dim=-1, keepdim=True)
near = torch.full((N, 1), 0.5)
far = torch.full((N, 1), 6.0)
depths = torch.from_numpy(rs.uniform(1.0, 4.0, size=(N,)).astype(np.float32))
for name, perturb, std in (("sigma_loss_det", 0., 0.), ("sigma_loss_rand", 1., 1.)):
    torch.manual_seed(7)
    t_rand = torch.rand(N, S) if perturb > 0 else None
    noise = torch.randn(N, S) * std if std > 0 else None
    torch.manual_seed(7)
    sl = L.SigmaLoss(S, perturb, std)
    for p in net.parameters():
        p.grad = None
    out = sl.calculate_loss(ro, rd, vd, near, far, depths, run, net)
    out.sum().backward()
    g = net.pts_linears[0].weight.grad.reshape(-1)
    extra = {}
    if t_rand is not None:
        extra.update(t_rand=t_rand, noise=noise)
    npz(name, rays_o=ro, rays_d=rd, viewdirs=vd, near=near, far=far, depths=depths, perturb=perturb, std=std, loss=out,
        g_pts0=g[::7], g_pts0_norm=g.double().norm(), g_alpha=net.alpha_linear.weight.grad.reshape(-1), **extra)
