"""Golden vectors for NeRF_RGB (DS_NeRF/run_nerf_helpers.py:159-216): colour network + frozen density network.

Run ONLY in the build container:   python tests/golden/make_golden_rgb.py
Weights are re-derived from seeds by the oracle (make_golden.py's convention); the fixture holds the input, the
reference module's output and the gradients of sum(out * d_out) w.r.t. the colour network's parameters."""
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
sd_a = O.make_wild_params(seed=31)
sd_rgb = {k: v for k, v in O.make_wild_params(seed=32).items() if not k.startswith("alpha_linear")}
alpha = H.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
alpha.load_state_dict(sd_a)
net = H.NeRF_RGB(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, alpha_model=alpha)
net.load_state_dict({**sd_rgb, **{"alpha_model." + k: v for k, v in sd_a.items()}})
rs = np.random.RandomState(33)
pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(40, 3)).astype(np.float32))
dirs = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(40, 3)).astype(np.float32)), dim=-1)
x = torch.cat([O.embed(pts, 10), O.embed(dirs, 4)], -1)
d_out = torch.from_numpy(rs.normal(size=(40, 4)).astype(np.float32))
out = net(x)
(out * d_out).sum().backward()
grads = {}
for k, p in net.named_parameters():   # every 61st element + the norm, like make_golden.py's gradient fixtures
    if not k.startswith("alpha_model."):
        g = p.grad.reshape(-1)
        grads["g_" + k] = g[::61] if g.numel() > 4096 else g
        grads["g_" + k + ".norm"] = g.double().norm()
assert all(p.grad is None for k, p in net.named_parameters() if k.startswith("alpha_model."))
npz("nerf_rgb_vd", pts=pts, dirs=dirs, x=x, d_out=d_out, out=out, keys=np.array(sorted(net.state_dict().keys())), **grads)
