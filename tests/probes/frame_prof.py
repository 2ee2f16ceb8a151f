"""Diagnostic: where a full 378x504 inference frame goes (per-kernel HIP-event times, wall clock, chunk size sweep)."""
import os, sys, time, contextlib, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
import bench
dev = torch.device("cuda")
H, W, focal, near, far = 378, 504, 400.0, 1.2, 9.0
torch.manual_seed(0)
import argparse
args = bench.make_args(argparse.Namespace(precision="bf16", n_rand=1024, n_coarse=64, n_fine=128))
with contextlib.redirect_stdout(io.StringIO()):
    kw_train, kw_test, *_ = S.create_nerf(args, device=dev)
kw_test.update(near=near, far=far)
c2w = torch.eye(4)[:3, :4].to(dev)
for chunk in (32768, 65536, 190512):
    with torch.no_grad():
        S.render(H, W, focal, chunk=chunk, c2w=c2w, **kw_test)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            S.render(H, W, focal, chunk=chunk, c2w=c2w, **kw_test)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        S._lib.prof_enable(True); S._lib.prof_read()
        S.render(H, W, focal, chunk=chunk, c2w=c2w, **kw_test)
        torch.cuda.synchronize()
        prof = S._lib.prof_read(); S._lib.prof_enable(False)
    tot = sum(ms for ms, c in prof.values())
    print(f"chunk {chunk}: wall {wall:.2f} ms; kernels {tot:.2f} ms:", {k: (round(ms, 3), c) for k, (ms, c) in prof.items()}, flush=True)
