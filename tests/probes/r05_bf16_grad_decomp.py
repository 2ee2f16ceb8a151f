"""Round 5: where does the bf16 gradient error on rays that are not opaque come from?
CPU only: the oracle's render() on the reference-trained fixtures, with the network evaluated under cumulative emulations of
what the HIP bf16 path rounds (oracle/nerf_oracle.py: nerf_forward_bf16emu is stage 2):

  0  fp32 (the reference's arithmetic; reproduces the fixture's stored gradients)
  1  + the encodings rounded to bf16 (weights are bf16-representable in these fixtures already)
  2  + every activation an MFMA consumes rounded to bf16; the ReLU masks are those of the rounded forward (flips)
  3  + every d z (and d raw) rounded to bf16 on its way back (what dgrad stores and the weight gradients consume)
  4  + split-K: the weight gradient is a sum of per-split partial sums, each rounded to bf16 once (27 interleaved splits)

For each stage: relative L2 error of the FINE and COARSE networks' parameter gradients against stage 0, whole network and
the worst tensors, plus the render's own error (rgb, acc).  The GPU's measured numbers stand beside it in the output file."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nerf_oracle as O          # noqa: E402
from helpers import load, T, render_case_nets, chunked_pytest_randoms, fixture_loss   # noqa: E402

torch.set_num_threads(8)


def q(t):
    return t.to(torch.bfloat16).to(torch.float32)


class RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return q(g)


class LinSplit(torch.autograd.Function):
    """y = x W^T + b with the weight / bias gradient as the sum of bf16-rounded per-split partial sums; 32-sample tile t
    belongs to split t % n_splits (the kernels' interleaved sweep)."""
    @staticmethod
    def forward(ctx, x, W, b, n_splits):
        ctx.save_for_backward(x, W)
        ctx.n_splits = n_splits
        return F.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        n = x.shape[0]
        tile = torch.arange(n) // 32
        gW = torch.zeros_like(W)
        gb = torch.zeros(W.shape[0])
        for s in range(ctx.n_splits):
            m = (tile % ctx.n_splits) == s
            if m.any():
                gW += q(g[m].t() @ x[m])
                gb += q(g[m].sum(0))
        return g @ W, gW, gb, None


def make_mlp(stage, n_splits=27):
    r_in, r_act, r_grad, split = stage >= 1, stage >= 2, stage >= 3, stage >= 4
    qa = q if r_act else (lambda t: t)

    def lin(h, W, b):
        z = LinSplit.apply(h, W, b, n_splits) if split else F.linear(h, W, b)
        return RoundGrad.apply(z) if r_grad else z

    def mlp(sd, x, input_ch=63, input_ch_views=27, skips=(4,), use_viewdirs=True):
        input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
        if r_in:
            input_pts, input_views = q(input_pts), q(input_views)
        h = input_pts
        for i in range(8):
            h = qa(F.relu(lin(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"])))
            if i in skips:
                h = torch.cat([input_pts, h], -1)
        alpha = lin(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
        feature = qa(lin(h, sd["feature_linear.weight"], sd["feature_linear.bias"]))
        h = torch.cat([feature, input_views], -1)
        h = qa(F.relu(lin(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"])))
        rgb = lin(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
        return torch.cat([rgb, alpha], -1)
    return mlp


def run(name, stage):
    g = load(name)
    sd_c, sd_f = render_case_nets(g)
    for sd in (sd_c, sd_f):
        for v in sd.values():
            v.requires_grad_(True)
    Nf = int(g["Nf"])
    n_rays = g["rgb"].reshape(-1, 3).shape[0]
    rnd = chunked_pytest_randoms(n_rays, int(g["chunk"]), 64, Nf, float(g["perturb"]), float(g["noise_std"]))
    kw = dict(N_samples=64, N_importance=Nf, perturb=float(g["perturb"]), white_bkgd=bool(g["white"]), lindisp=bool(g["lindisp"]),
              retraw=True, need_alpha=bool(g["need_alpha"]), detach_weights=bool(g["detach"]))
    rgb, disp, acc, depth, ex = O.render(rays=T(g["rays"]), H=int(g["H"]), W=int(g["W"]), focal=float(g["focal"]), chunk=int(g["chunk"]),
                                         ndc=bool(g["ndc"]), near=float(g["near"]), far=float(g["far"]), use_viewdirs=True,
                                         sd_coarse=sd_c, sd_fine=sd_f, randoms=rnd, mlp=make_mlp(stage), **kw)
    target = T(g["target"])
    loss = fixture_loss(g, lambda x: O.img2mse(x, target), rgb, ex.get("rgb0"), disp)
    loss.backward()
    grads = {pfx: {k: v.grad.detach().clone() for k, v in sd.items() if v.grad is not None} for pfx, sd in (("c", sd_c), ("f", sd_f))}
    return dict(loss=float(loss), rgb=rgb.detach(), acc=acc.detach(), grads=grads, acc_ref=g["acc"].reshape(-1))


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    names = ["render_trained_black_vd", "render_trained_fine_vd"]
    labels = ["fp32", "+ bf16 encodings", "+ bf16 activations (mask flips)", "+ bf16 d z", "+ bf16 split-K partials (27 splits)"]
    for name in names:
        base = run(name, 0)
        acc = base["acc_ref"]
        print(f"== {name}: {len(acc)} rays, acc min {acc.min():.3f}, {int((acc < 0.99).sum())} rays below 0.99; loss {base['loss']:.6f}")
        flat0 = {p: torch.cat([v.reshape(-1) for v in base["grads"][p].values()]) for p in "cf"}
        prev = None
        for st in range(1, 5):
            r = run(name, st)
            line = f"  stage {st} {labels[st]:38s}"
            for p, nm in (("c", "coarse"), ("f", "fine")):
                flat = torch.cat([v.reshape(-1) for v in r["grads"][p].values()])
                worst = max(((rel(r["grads"][p][k], base["grads"][p][k]), k) for k in base["grads"][p] if k.endswith("weight")))
                line += f" | {nm} whole {rel(flat, flat0[p]):.4f}  worst weight {worst[0]:.4f} ({worst[1]})"
                if prev is not None:
                    flat_prev = torch.cat([v.reshape(-1) for v in prev["grads"][p].values()])
                    line += f"  step {rel(flat, flat_prev):.4f}"
            line += f" | rgb {float((r['rgb'] - base['rgb']).abs().max()):.2e} acc {float((r['acc'] - base['acc']).abs().max()):.2e} loss {abs(r['loss'] / base['loss'] - 1):.2e}"
            print(line, flush=True)
            prev = r


if __name__ == "__main__":
    main()
