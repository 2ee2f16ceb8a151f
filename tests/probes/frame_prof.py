"""Diagnostic: per-kernel time of one full 378x504 frame render (bf16 inference)."""
import os, sys, time, torch, argparse, tempfile, contextlib, io, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("spin-nerf_amd")
sys.argv = [sys.argv[0]]
import bench
ns = argparse.Namespace(n_fine=128, n_coarse=64, precision="bf16")
with contextlib.redirect_stdout(io.StringIO()):
    kw_train, kw_test, *_ = S.create_nerf(bench.make_args(ns), device=torch.device("cuda"))
kw_test.update(near=1.2, far=9.0)
c2w = torch.eye(4)[:3, :4].cuda()
with torch.no_grad():
    for chunk in (32768, 65536, 190512):
        S.render(378, 504, 400.0, chunk=chunk, c2w=c2w, **kw_test); torch.cuda.synchronize()
        S._lib.prof_enable(True); S._lib.prof_read()
        t0 = time.perf_counter()
        S.render(378, 504, 400.0, chunk=chunk, c2w=c2w, **kw_test); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p = S._lib.prof_read(); S._lib.prof_enable(False)
        print("chunk", chunk, "frame ms", round(dt * 1e3, 1), {k: round(v[0], 2) for k, v in p.items()}, "kernel sum", round(sum(v[0] for v in p.values()), 1))
