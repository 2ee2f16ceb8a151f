"""Debug: per-tensor error of the bf16 backward (recompute path) vs the bf16-emulating oracle; where the bad elements are."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib, torch
import test_gpu_kernels as T
S = importlib.import_module("spin-nerf_amd")
from oracle import nerf_oracle as O
for rep in range(3):
  for vd in (True, False):
    for n_rays, sps in [(33, 64), (64, 192), (256, 192)]:
        sd, net = T._mlp_grad_case(S, vd, "bf16", n_rays, sps, seed=6 + rep, wild=False, mlp=O.nerf_forward_bf16emu)
        got = net.named_views(net.flat.grad)
        for k, p in sd.items():
            if p.grad is None: continue
            rel, cos = T._rel_l2(got[k], p.grad)
            if not (rel < 5e-2):
                g, r = got[k].detach().cpu().float(), p.grad
                bad = ((g - r).abs() > 0.5 * r.abs().max()) | ~torch.isfinite(g)
                if g.dim() == 2:
                    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
                    print(f"rep {rep} vd {vd} n_rays {n_rays} sps {sps} {k}: rel {rel:.2e} bad {int(bad.sum())} rows {rows[:24]} cols {cols[:24]}")
                    rr, cc = rows[0], cols[0]
                    print("    got", [f"{float(v):.3g}" for v in g[rr, cc:cc + 8]], "ref", [f"{float(v):.3g}" for v in r[rr, cc:cc + 8]])
                else:
                    print(f"rep {rep} vd {vd} n_rays {n_rays} sps {sps} {k}: rel {rel:.2e} bad {bad.nonzero().flatten().tolist()[:24]}")
print("done")
