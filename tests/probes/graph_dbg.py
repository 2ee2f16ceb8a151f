"""Diagnostic: parameters / gradients of the captured-graph route of RenderTrainer.step against the eager route, step by step,
and the device-side snr_step_state against the host counters (found the one-ulp difference of pow() between host and
device that made the two routes drift apart before the bias corrections were formed by squaring on both sides)."""
import sys, importlib, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
import test_gpu_train_step as T
train = importlib.import_module("spin-nerf_amd.train")
(a, b), hwf, rays, target, _ = T._two_trainers("bf16", 128)
ta, tb = a[0], train.RenderTrainer(b[0].kw, lrate=5e-4, graph=True)
for i in range(4):
    ta.step(*hwf, rays, target); tb.step(*hwf, rays, target)
    for k, (na, nb) in enumerate(zip(a[1], b[1])):
        d = (na.flat.detach() - nb.flat.detach()).abs()
        gd = (na.flat.grad - nb.flat.grad).abs()
        print(f"step {i} net {k}: param max diff {float(d.max()):.3e} frac>1e-6 {float((d > 1e-6).float().mean()):.4f}; grad max diff {float(gd.max()):.3e} grad max {float(nb.flat.grad.abs().max()):.3e}")
    if tb._graph is not None:
        st = S._lib.StepState.from_buffer_copy(bytes(tb._graph["state"].cpu().numpy()))
        print("  device state:", st.offset_base, st.opt_step, st.global_step, st.lr, st.bc1, st.bc2_sqrt, " host:", tb._draws, tb.opt_step, tb.global_step, tb._lr)
