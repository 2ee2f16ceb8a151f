"""Round 5: what does the weight DMA really cost?  The "no DMA" ablation (SNR_ABLATE=8) multiplies with whatever an
un-written LDS holds — zeros — and zero operands run at a higher clock (MI355X_MICROARCH.md: DVFS give-back).  SNR_ABLATE=64
issues the DMA during each workgroup's FIRST pass only: with a persistent grid (SNR_CHAIN_GRID=256) and 786 432 samples per
launch 11 of 12 passes run without DMA on real-valued weights.  Inference forward, shipped kernel; library chosen by SNR_LIB."""
import os, sys, torch, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SNR_CHAIN_GRID"] = "256"
L = importlib.import_module("spin-nerf_amd._lib")
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
torch.manual_seed(0)
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
packed = net.packed_weights()
M = int(os.environ.get("M", "786432"))
TRAIN = os.environ.get("TRAIN") == "1"
pts = torch.randn(M, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)
raw = torch.empty(M, 4, device="cuda")
act = torch.empty(lib.snr_mlp_act_bytes(net.cfg, M), dtype=torch.uint8, device="cuda") if TRAIN else None
def go(n):
    for _ in range(n):
        lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(act), L.stream())
go(5)
ts = []
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(10); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(os.path.basename(os.environ.get("SNR_LIB", "base")), ("training" if TRAIN else "inference") + " forward of %d samples: %s ms  (raw finite: %s, |raw| max %.3g)" % (M, ["%.4f" % t for t in ts], bool(torch.isfinite(raw).all()), float(raw.abs().max())))
