"""Cycle counters around the block waits of the chained kernels (library built with -DSNR_TIMING, named by SNR_LIB):
per wave, how long the counted DMA wait and the barrier take per block, against the kernel's total."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("spin-nerf_amd._lib")
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
raw_lib = ctypes.CDLL(L.LIB_PATH)
M = int(os.environ.get("M", 196608))
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
pts = torch.randn(M // 192, 192, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)

def read(fn, name):
    buf = (ctypes.c_ulonglong * 8)()
    getattr(raw_lib, fn)(buf)
    tw, tb, _, na, tk, nw = [buf[i] for i in range(6)]
    if nw:
        print(f"{name:10s} per wave: kernel {tk/nw:9.0f} ticks, {na/nw:5.0f} blocks, DMA wait {tw/nw:8.0f} ({tw/na:5.0f}/block), "
              f"barrier {tb/nw:8.0f} ({tb/na:5.0f}/block) -> {100*(tw+tb)/tk:.1f} % of the kernel at the block entry")
for _ in range(2):
    out = net.query(pts, vd); torch.cuda.synchronize(); read("snr_debug_read_fwd", "fwd train")
    out.backward(torch.randn_like(out)); torch.cuda.synchronize(); read("snr_debug_read", "dgrad")
    net.flat.grad = None
with torch.no_grad():
    net.query(pts, vd); torch.cuda.synchronize(); read("snr_debug_read_fwd", "fwd infer")
