"""Diagnostic: host-side time of one trainer.step (tiny batch -> the GPU never limits) vs the bench batch."""
import os, sys, time, importlib, contextlib, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spin_nerf_amd as S
import bench
RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
import argparse
ns = argparse.Namespace(n_fine=128, n_coarse=64, precision="bf16")
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    kw, kwt, *_ = S.create_nerf(bench.make_args(ns), device=dev)
kw.update(near=1.2, far=9.0)
tr = RenderTrainer(kw)
H, W, f = 378, 504, 400.0
for n_rand in (32, 1024):
    b = bench.synthetic_batches(4, n_rand, H, W, f, 5, dev)
    for i in range(5):
        tr.step(H, W, f, *b[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for i in range(n):
        tr.step(H, W, f, *b[i % 4])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n_rand {n_rand}: host loop {1e3 * (t1 - t0) / n:.3f} ms/step, with final sync {1e3 * (t2 - t0) / n:.3f} ms/step")
