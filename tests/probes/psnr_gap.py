"""Diagnostic: bf16 vs fp32 training quality at equal iterations on the analytic sphere (BASELINE.json: matched PSNR),
paired by seed (same initial weights, same ray batches, same in-kernel random draws)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
iters = int(os.environ.get("ITERS", 4000))
seeds = [int(s) for s in os.environ.get("SEEDS", "0,1,2,3,4,5,6,7,8,9,10,11").split(",")]
rows = []
for seed in seeds:
    r = {"seed": seed}
    for prec in ("bf16", "fp32"):
        ps, held = T.train(prec, iters, seed=seed)
        r[prec] = (float(np.mean(ps[-50:])), float(np.mean(ps[-500:])), held, float(T.train.seen_view_psnr),
                   float(np.mean(ps[-1000:-500])))
    rows.append(r)
    print(json.dumps(r), flush=True)
for j, name in enumerate(("train last 50", "train last 500", "held-out view", "training camera, deterministic full frame",
                          "train steps -1000..-500")):
    b = np.array([r["bf16"][j] for r in rows]); f = np.array([r["fp32"][j] for r in rows]); d = b - f
    print(f"{name}: bf16 {b.mean():.2f} (sd {b.std(ddof=1):.2f})  fp32 {f.mean():.2f} (sd {f.std(ddof=1):.2f})  "
          f"paired difference {d.mean():+.2f} dB, sd {d.std(ddof=1):.2f}, standard error {d.std(ddof=1) / np.sqrt(len(d)):.2f}")
