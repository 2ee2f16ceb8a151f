"""Diagnostic (not a test): per-key error statistics of render() vs the golden fixtures."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from helpers import load, RENDER_CASES
from test_gpu_render import build, run
for name in RENDER_CASES:
    g = load(name)
    for prec in ("fp32", "bf16"):
        net_c, net_f, kw = build(S, g, prec)
        with torch.no_grad():
            rgb, disp, acc, depth, ex = run(S, g, kw, True)
        out = dict(rgb=rgb, disp=disp, acc=acc, depth=depth, **{"x_" + k: v for k, v in ex.items()})
        line = []
        for k, v in out.items():
            ref = g[k]
            d = np.abs(v.cpu().numpy() - ref)
            line.append(f"{k}:max={d.max():.2e},>1e-4:{(d > 1e-4).mean() * 100:.2f}%")
        print(name, prec, " ".join(line), flush=True)
