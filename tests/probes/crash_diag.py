import sys, os, importlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import test_gpu_kernels as T
from helpers import load, T as TT, mlp_case_params
S = importlib.import_module("spin-nerf_amd")
name, prec, train, n = sys.argv[1], sys.argv[2], sys.argv[3] == "1", int(sys.argv[4])
g = load(name); vd = bool(g["use_viewdirs"]); sd = mlp_case_params(g)
net = T.make_net(S, sd, vd, prec, out_ch=4 if vd else 5)
pts = torch.randn(n, 1, 3, device="cuda"); dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1) if vd else None
net.packed_weights(); torch.cuda.synchronize(); print("packed ok", flush=True)
print("launch", name, prec, train, n, flush=True)
with torch.set_grad_enabled(train):
    out = net.query(pts, dirs)
torch.cuda.synchronize()
print("ok", float(out.abs().max()), flush=True)
