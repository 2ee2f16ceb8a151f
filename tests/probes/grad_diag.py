"""Diagnostic (not a test): relative L2 error / cosine of MLP parameter grads vs oracle autograd."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from test_gpu_kernels import _mlp_grad_case
for prec in ("fp32", "bf16"):
    for (n, s) in ((5, 7), (33, 64), (16, 192), (64, 192)):
        sd, net = _mlp_grad_case(S, True, prec, n, s, seed=5)
        got = net.named_views(net.flat.grad)
        line = []
        for k, p in sd.items():
            a, b = got[k].cpu().double().reshape(-1), p.grad.double().reshape(-1)
            rel = float((a - b).norm() / b.norm()); cos = float((a @ b) / (a.norm() * b.norm()))
            line.append(f"{k.replace('_linears','').replace('.weight','.w').replace('.bias','.b')}:{rel:.1e}/{1-cos:.0e}")
        print(prec, n, s, " ".join(line), flush=True)
