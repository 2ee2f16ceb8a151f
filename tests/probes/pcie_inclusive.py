"""Diagnostic: the training step with the batch handed over from HOST memory every step (what the reference's loop does:
next(raysRGB_iter).to(device)) against batches already resident — the PCIe-inclusive rate DESIGN.md quotes."""
import os, sys, time, importlib, contextlib, io, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
import bench
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
dev = torch.device("cuda")
ns = argparse.Namespace(n_fine=128, n_coarse=64, precision="bf16")
with contextlib.redirect_stdout(io.StringIO()):
    kw, *_ = S.create_nerf(bench.make_args(ns), device=dev)
kw.update(near=1.2, far=9.0)
tr = RenderTrainer(kw)
H, W, f = 378, 504, 400.0
b = bench.synthetic_batches(8, 1024, H, W, f, 5, dev)
host = [(r.cpu().pin_memory(), t.cpu().pin_memory()) for r, t in b]
pageable = [(r.cpu(), t.cpu()) for r, t in b]
for name, src in (("resident", b), ("pinned host", host), ("pageable host", pageable)):
    for i in range(10):
        r, t = src[i % 8]
        tr.step(H, W, f, r.to(dev, non_blocking=True), t.to(dev, non_blocking=True))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for i in range(n):
        r, t = src[i % 8]
        tr.step(H, W, f, r.to(dev, non_blocking=True), t.to(dev, non_blocking=True))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:14s}: {dt * 1e3:.4f} ms/step  {1024 / dt:,.0f} rays/s", flush=True)
