"""Diagnostic: the analytic-sphere test scene with a wider field of view (all of the sampled volume observed by the
training cameras) — do held-out views then come out right?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
for focal in (110.0, 80.0):
    T.FOCAL = focal
    for prec in ("fp32", "bf16"):
        for noise in (1.0, 0.0):
            p, held = T.train(prec, 2000, seed=0, noise=noise)
            print(f"focal {focal} {prec} noise {noise}: train last400 {np.mean(p[-400:]):.2f} dB, training camera {T.train.seen_view_psnr:.2f} dB, held-out {held:.2f} dB", flush=True)
