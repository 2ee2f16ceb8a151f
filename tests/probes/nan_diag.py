"""do the bf16 backward results depend on the initial contents of the workspaces (i.e. is unwritten memory read)?"""
import sys, os, importlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import test_gpu_kernels as T
S = importlib.import_module("spin-nerf_amd")
ops = importlib.import_module("spin-nerf_amd.ops"); nerf = importlib.import_module("spin-nerf_amd.nerf")
real_empty = torch.empty
def filled(val):
    def f(*a, **k):
        t = real_empty(*a, **k)
        if t.dtype == torch.uint8: t.fill_(val)
        return t
    return f
res = {}
for vd in (True, False):
    for val in (0, 0xFF, 0x7F):
        torch.empty = filled(val)
        try:
            sd, net = T._mlp_grad_case(S, vd, "bf16", int(os.environ.get("NR", 33)), 64, seed=6, wild=False, mlp=T.O.nerf_forward_bf16emu)
        finally:
            torch.empty = real_empty
        g = net.flat.grad.clone()
        res[(vd, val)] = g
        print(vd, hex(val), "nonfinite", int((~torch.isfinite(g)).sum()), "norm", float(torch.nan_to_num(g).norm()))
    a, b = res[(vd, 0)], res[(vd, 0xFF)]
    print(vd, "max |diff| between fills:", float((torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max()))
