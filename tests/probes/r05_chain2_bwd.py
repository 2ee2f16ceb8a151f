"""Round 5: chain2 dgrad (SNR_CHAIN2=1) against the shipped dgrad kernel (SNR_CHAIN2=0): bit-identical parameter gradients
(the weight-gradient pass consumes every d z section the dgrad kernel writes), then per-kernel times of both."""
import os, sys, torch, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = importlib.import_module("spin-nerf_amd._lib")
S = importlib.import_module("spin-nerf_amd")
lib = L.load()
torch.manual_seed(0)
net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()


def mode(v):
    os.environ["SNR_CHAIN2"] = str(v)
    lib.snr_tunables_reload()


ok = True
sizes = (196608,) if "--quick" in sys.argv else (192, 192 * 3, 192 * 40, 65536 // 64 * 64, 196608)
for M in sizes:
    rays = M // 192
    pts = torch.randn(rays, 192, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(rays, 3, device="cuda"), dim=-1)
    go = None
    grads = []
    for v in (0, 1):
        mode(v)
        net.flat.grad = None
        out = net.query(pts, vd)
        if go is None:
            go = torch.randn_like(out)
        out.backward(go)
        torch.cuda.synchronize()
        grads.append(net.flat.grad.clone())
    same = torch.equal(grads[0].view(torch.int32), grads[1].view(torch.int32))
    d = (grads[0] - grads[1]).abs().max().item()
    print(f"M={M:7d} gradients bit-identical: {same}  (max abs diff {d:.3e}, |g| max {grads[0].abs().max().item():.3e}, nan {torch.isnan(grads[1]).sum().item()})", flush=True)
    ok = ok and same

M = 196608
pts = torch.randn(M // 192, 192, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)
L.prof_enable(True)
for rep in range(3):
    for v in (0, 1):
        mode(v)
        for i in range(8):
            net.flat.grad = None
            out = net.query(pts, vd)
            out.backward(torch.ones_like(out))
            if i == 1:
                torch.cuda.synchronize(); L.prof_read()
        torch.cuda.synchronize()
        p = L.prof_read()
        print(f"SNR_CHAIN2={v}", {k: round(val[0] / val[1], 4) for k, val in p.items()}, flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
