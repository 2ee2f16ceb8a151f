"""Diagnostic: the 32 encoded features the forward kernel saves vs the oracle's hash_encode."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from oracle import hashgrid_oracle as H
from test_gpu_hashgrid import make, samples
L = S._lib
lib = L.load()
sd, net = make(S, 4)
for spread in (40.0, 2.0):
    pts, dirs = samples(5, 41, 16, spread)
    n = 41 * 16
    p = pts.reshape(-1, 3).cuda().contiguous()
    vd = dirs.cuda().contiguous()
    raw = torch.empty(n, 4, device="cuda")
    act = torch.zeros(lib.snr_hashgrid_act_bytes(n), dtype=torch.uint8, device="cuda")
    L.check(lib.snr_hashgrid_forward(L.ptr(net.flat.detach()), L.ptr(net.packed_weights()), L.ptr(p), None, 0, None, L.ptr(vd), 3,
                                     n, 16, L.ptr(raw), L.ptr(act), L.stream()), "fwd")
    a = act.cpu().view(torch.bfloat16).reshape(-1, 2, 2, 32, 8).float()    # [tile][q][g][s][e]
    enc = a.permute(0, 3, 1, 2, 4).reshape(-1, 32)[:n]                      # [tile*32 + s][16q + 8g + e]
    x01 = H.to_unit_cube(pts.reshape(-1, 3))
    ref = H.hash_encode(x01, sd["encoder.params"])
    refq = ref.to(torch.bfloat16).float()
    d = (enc - refq).abs()
    print("spread", spread, "exact match frac", float((d == 0).float().mean()), "max abs", float(d.max()), "ref max", float(ref.abs().max()))
    per_level = d.reshape(n, 16, 2).amax((0, 2))
    print("   per-level max abs diff", [round(float(v), 5) for v in per_level])
    print("   vs unrounded ref: rms", float((enc - ref).pow(2).mean().sqrt()))
