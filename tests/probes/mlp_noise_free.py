"""Diagnostic: does the big-MLP path train WITHOUT density noise on the analytic sphere (black background)?  Before the
compositing-backward fix (0 * NaN on rays that hit nothing) noise-free runs died into the all-empty state."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_train import train
for prec in ("fp32", "bf16"):
    for seed in (0, 1, 2):
        p, t = train(prec, 1200, seed=seed, noise=0.0)
        print(f"{prec} seed {seed} noise 0: it100 {np.mean(p[90:110]):.2f} it600 {np.mean(p[590:610]):.2f} last400 {np.mean(p[-400:]):.2f} dB, held-out view {t:.2f} dB", flush=True)
