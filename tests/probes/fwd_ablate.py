"""Diagnostic: time snr_mlp_forward (bf16, inference + training mode) for the library named by SNR_LIB."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib
L = importlib.import_module("spin-nerf_amd._lib")
if os.environ.get("SNR_LIB"):
    L.LIB_PATH = os.environ["SNR_LIB"]
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
M = 196608
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
pts = torch.randn(M, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(1024, 3, device="cuda"), dim=-1)
raw = torch.empty(M, 4, device="cuda")
packed = net.packed_weights()
act = torch.empty(lib.snr_mlp_act_bytes(net.cfg, M), dtype=torch.uint8, device="cuda")
for name, a in (("inference", None), ("train", act)):
    for _ in range(3):
        lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(a), L.stream())
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n):
        lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(a), L.stream())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{os.environ.get('SNR_LIB','base'):>50s} {name:9s} {dt*1e3:.3f} ms  {M*1186816/dt/1e12:.0f} TF")
    if hasattr(lib, "snr_debug_read_fwd"):
        buf = (ctypes.c_ulonglong * 8)()
        lib.snr_debug_read_fwd(buf)
        tw, tb, ti, na, tk, nw = [buf[i] for i in range(6)]
        print(f"    per wave: pass prologue {buf[2]/nw:.0f} cyc, skip-layer encoding {buf[6]/nw:.0f}, view-direction encoding {buf[7]/nw:.0f}  (of kernel {tk/nw:.0f}; {M/256/ (nw/8):.1f} passes per workgroup)")
        print(f"    per wave: kernel {tk/nw:.0f} cyc, acquires {na/nw:.0f}, wait {tw/nw:.0f} ({tw/na:.0f}/acq), barrier {tb/nw:.0f} ({tb/na:.0f}/acq), issue {ti/nw:.0f} ({ti/na:.0f}/acq)  [s_memtime ticks]")
