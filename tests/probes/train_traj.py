"""Diagnostic: PSNR / acc trajectory of HIP training on the analytic sphere scene."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
import importlib
S = importlib.import_module("spin-nerf_amd")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
white = len(sys.argv) > 3 and sys.argv[3] == "white"
noise = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ps, tp = T.train(prec, iters, white=white, noise=noise, seed=seed)
for i in range(0, iters, max(1, iters // 15)):
    print(i, round(float(np.mean(ps[i:i + 20])), 2))
print("final", round(float(np.mean(ps[-50:])), 2), "held-out", round(tp, 2))
