"""Round 5: the chain2 kernels (SNR_CHAIN2=1) against the shipped chain kernels (SNR_CHAIN2=0), same process: bit-identical
outputs, then the launch times of both (HIP events around 20 launches each, interleaved A/B/A/B)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib
L = importlib.import_module("spin-nerf_amd._lib")
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
torch.manual_seed(0)
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
with torch.no_grad():
    net.flat.mul_(1.7)   # wider activations than the default initialisation
net.mark_weights_changed() if hasattr(net, "mark_weights_changed") else None
packed = net.packed_weights()


def mode(v):
    os.environ["SNR_CHAIN2"] = str(v)
    lib.snr_tunables_reload()


def fwd(M, train=False):
    pts = torch.randn(M, 3, device="cuda") * 1.5
    vd = torch.nn.functional.normalize(torch.randn((M + 191) // 192, 3, device="cuda"), dim=-1)
    out = []
    for v in (0, 1):
        mode(v)
        raw = torch.full((M, 4), float("nan"), device="cuda")
        act = torch.zeros(lib.snr_mlp_act_bytes(net.cfg, M), dtype=torch.uint8, device="cuda") if train else None
        L.check(lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(act), L.stream()), "fwd")
        torch.cuda.synchronize()
        out.append((raw, act))
    return out


ok = True
for train in (False, True) if "--train" in sys.argv else (False,):
    for M in ((196608,) if "--quick" in sys.argv else (1, 31, 32, 33, 256, 257, 5000, 65536, 196608, 196608 + 77)):
        (r0, a0), (r1, a1) = fwd(M, train)
        same = torch.equal(r0.view(torch.int32), r1.view(torch.int32))
        msg = f"train={int(train)} M={M:7d} raw bit-identical: {same}"
        if not same:
            d = (r0 - r1).abs()
            msg += f"  max diff {d.max().item():.3e} nan {torch.isnan(r1).sum().item()} first bad sample {int((d.amax(1) > 0).nonzero()[0])}"
        if train:
            sa = torch.equal(a0, a1)
            msg += f"  saved activations identical: {sa}"
            if not sa:
                bad = (a0 != a1).nonzero().flatten()
                msg += f" ({bad.numel()} bytes differ, first at {int(bad[0])} of {a0.numel()})"
            same = same and sa
        print(msg, flush=True)
        ok = ok and same

M = 196608
pts = torch.randn(M, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(1024, 3, device="cuda"), dim=-1)
raw = torch.empty(M, 4, device="cuda")
act = torch.empty(lib.snr_mlp_act_bytes(net.cfg, M), dtype=torch.uint8, device="cuda")
for name, a in (("inference", None),) + ((("train", act),) if "--train" in sys.argv else ()):
    res = {0: [], 1: []}
    for rep in range(4):
        for v in (0, 1):
            mode(v)
            for _ in range(3):
                lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(a), L.stream())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts), None, 0, None, L.ptr(vd), 3, M, 192, L.ptr(raw), L.ptr(a), L.stream())
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 20)
    for v in (0, 1):
        t = sorted(res[v])
        print(f"{name:9s} SNR_CHAIN2={v}: median {t[len(t)//2]:.4f} ms  min {t[0]:.4f}  ({M*1186816/t[len(t)//2]/1e9:.0f} TFLOP/s)  all {['%.4f' % x for x in res[v]]}")
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
