"""Diagnostic: per-kernel time of snr_mlp_backward (bf16) for the library named by SNR_LIB."""
import os, sys, torch, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = importlib.import_module("spin-nerf_amd._lib")
if os.environ.get("SNR_LIB"):
    L.LIB_PATH = os.environ["SNR_LIB"]
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
M = int(os.environ.get("M", 196608))
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
pts = torch.randn(M // 192, 192, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)
L.prof_enable(True)
for i in range(6):
    net.flat.grad = None
    out = net.query(pts, vd)
    out.backward(torch.randn_like(out))
    if i == 1: L.prof_read()
torch.cuda.synchronize()
p = L.prof_read()
print(os.environ.get("SNR_LIB", "base")[-24:], M, {k: round(v[0] / v[1], 4) for k, v in p.items()})
