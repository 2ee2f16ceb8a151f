"""Kernel split of the 378x504 frame (bench.py's ms_per_frame path).  Run plain for the wall clock, or under
`rocprofv3 --kernel-trace --stats` for the per-kernel times; SNR_PROF-style event timing is printed as well."""
import os, sys, time, argparse, contextlib, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import spin_nerf_amd as S

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=1024 * 32)
ap.add_argument("--frames", type=int, default=3)
a = ap.parse_args()
ns = argparse.Namespace(precision="bf16", n_rand=1024, n_coarse=64, n_fine=128)
device = torch.device("cuda", 0)
H, W, focal, near, far = 378, 504, 400.0, 1.2, 9.0
torch.manual_seed(0)
args = bench.make_args(ns)
with contextlib.redirect_stdout(io.StringIO()):
    kw_train, kw_test, start, grad_vars, _ = S.create_nerf(args, device=device)
kw_test.update(near=near, far=far)
c2w = torch.eye(4)[:3, :4].to(device)
with torch.no_grad():
    S.render(H, W, focal, chunk=a.chunk, c2w=c2w, **kw_test)
    torch.cuda.synchronize()
    S._lib.prof_enable(True); S._lib.prof_read()
    t = time.perf_counter()
    for _ in range(a.frames):
        S.render(H, W, focal, chunk=a.chunk, c2w=c2w, **kw_test)
    torch.cuda.synchronize()
    wall_prof = (time.perf_counter() - t) / a.frames * 1e3
    prof = S._lib.prof_read(); S._lib.prof_enable(False)
    t = time.perf_counter()
    for _ in range(a.frames):
        S.render(H, W, focal, chunk=a.chunk, c2w=c2w, **kw_test)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / a.frames * 1e3
print(f"chunk {a.chunk}: frame {wall:.2f} ms (with event timing on: {wall_prof:.2f} ms)")
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    ms = v[0] / a.frames
    tot += ms
    print(f"   {k:28s} {ms:8.3f} ms / frame   {v[1] / a.frames:6.1f} launches")
print(f"   sum of library kernels      {tot:8.3f} ms / frame")
