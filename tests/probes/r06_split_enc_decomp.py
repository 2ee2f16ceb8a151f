"""Round 6: what split-bf16 (hi + lo) positional encodings would buy, on the CPU oracle, BEFORE the kernel
work: the bf16 emulation of tests/probes/r05_bf16_grad_decomp.py (activations rounded, ReLU flips) with the encodings
  a) rounded to bf16 (the shipped path),  b) as hi + lo bf16 pairs (16 mantissa bits),  c) exact fp32,
and d) split encodings AND hi + lo split of the first hidden activation h0 (to see where the next floor is).
Relative L2 error of each network's parameter gradient against fp32, per fixture."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
import r05_bf16_grad_decomp as D   # noqa: E402
from oracle import nerf_oracle as O   # noqa: E402

q = D.q


def make_mlp(enc, act=True):
    qa = q if act else (lambda t: t)
    qe = {"bf16": q, "split": lambda t: q(t) + q(t - q(t)), "fp32": lambda t: t}[enc]

    def mlp(sd, x, input_ch=63, input_ch_views=27, skips=(4,), use_viewdirs=True):
        input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
        input_pts, input_views = qe(input_pts), qe(input_views)
        h = input_pts
        for i in range(8):
            h = qa(F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"])))
            if i in skips:
                h = torch.cat([input_pts, h], -1)
        alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
        feature = qa(F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"]))
        h = torch.cat([feature, input_views], -1)
        h = qa(F.relu(F.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"])))
        rgb = F.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
        return torch.cat([rgb, alpha], -1)
    return mlp


def run(name, mlp):
    real = D.make_mlp
    D.make_mlp = lambda stage, n_splits=27: mlp
    try:
        return D.run(name, 2)
    finally:
        D.make_mlp = real


if __name__ == "__main__":
    for name in ["render_trained_black_vd", "render_trained_fine_vd"]:
        base = D.run(name, 0)
        flat0 = {p: torch.cat([v.reshape(-1) for v in base["grads"][p].values()]) for p in "cf"}
        for label, mlp in (("bf16 encodings + bf16 activations (shipped)", make_mlp("bf16")),
                           ("split encodings + bf16 activations", make_mlp("split")),
                           ("fp32 encodings + bf16 activations", make_mlp("fp32")),
                           ("split encodings, fp32 activations", make_mlp("split", act=False))):
            r = run(name, mlp)
            line = f"{name:26s} {label:46s}"
            for p, nm in (("c", "coarse"), ("f", "fine")):
                flat = torch.cat([v.reshape(-1) for v in r["grads"][p].values()])
                worst = max(((D.rel(r["grads"][p][k], base["grads"][p][k]), k) for k in base["grads"][p] if k.endswith("weight")))
                line += f" | {nm} whole {D.rel(flat, flat0[p]):.4f} worst {worst[0]:.4f} ({worst[1]})"
            line += f" | rgb {float((r['rgb'] - base['rgb']).abs().max()):.2e}"
            print(line, flush=True)
