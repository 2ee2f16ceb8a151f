"""Debug: per-tensor error of the bf16 backward (recompute path) vs the bf16-emulating oracle, several sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib, torch
import test_gpu_kernels as T
S = importlib.import_module("spin-nerf_amd")
from oracle import nerf_oracle as O
for n_rays, sps in [(33, 64), (64, 192), (256, 192)]:
    sd, net = T._mlp_grad_case(S, True, "bf16", n_rays, sps, seed=6, wild=False, mlp=O.nerf_forward_bf16emu)
    got = net.named_views(net.flat.grad)
    print("n_rays", n_rays, "sps", sps)
    for k, p in sd.items():
        if p.grad is None: continue
        rel, cos = T._rel_l2(got[k], p.grad)
        print(f"   {k:28s} rel {rel:.3e} cos {cos:.5f}")
    for k in ("pts_linears.2.weight", "pts_linears.4.weight"):
        g, r = got[k].detach().cpu(), sd[k].grad
        bad = ((g - r).abs() > 0.5 * r.abs().max()) | ~torch.isfinite(g)
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print("   ", k, "bad elements", int(bad.sum()), "rows", rows[:40], "cols", cols[:40])
    k = "pts_linears.2.weight"
    g, r = got[k].detach().cpu(), sd[k].grad
    e = (g - r).abs()
    print("    max ref", float(r.abs().max()), "col 3 errs:", [f"{float(v):.2e}" for v in e[:40, 3]])
    print("    col 4 errs:", [f"{float(v):.2e}" for v in e[:16, 4]])
    print("    got col 3:", [f"{float(v):.2e}" for v in g[:40, 3]])
    for k in ("pts_linears.2.weight", "pts_linears.4.weight"):
        g, r = got[k].detach().cpu(), sd[k].grad
        bad = ((g - r).abs() > 0.5 * r.abs().max()) | ~torch.isfinite(g)
        rows = bad.any(1).nonzero().flatten().tolist()[:6]
        for rr in rows:
            print("    ", k, "row", rr, "got cols 0..15:", [f"{float(v):.3g}" for v in g[rr, :16]])
            print("    ", k, "row", rr, "ref cols 0..15:", [f"{float(v):.3g}" for v in r[rr, :16]])
