"""Measure the bf16 path's errors on the reference-trained fixture (gates in tests/test_gpu_render.py = 2x these)."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_render as R
from helpers import load, T, chunked_pytest_randoms
import spin_nerf_amd as S
from helpers import fixture_loss, TRAINED_CASES
for name in (sys.argv[1:] or TRAINED_CASES):
  print("=====", name)
  g = load(name)
  n = g["rgb"].reshape(-1, 3).shape[0]
  for prec in ("fp32", "bf16"):
      net_c, net_f, kw = R.build(S, g, prec)
      with torch.no_grad():
          rgb, disp, acc, depth, ex = R.run(S, g, kw, True)
      d = lambda a, b: float(np.abs(R.npy(a) - b).max())
      print(prec, "coarse rgb0", d(ex["rgb0"], g["x_rgb0"]), "acc0", d(ex["acc0"], g["x_acc0"]), "free-running rgb", d(rgb, g["rgb"]), "acc", d(acc, g["acc"]))
      rays = R.pack_rays(S, g)
      z = T(g["x_z_vals"]).reshape(n, -1).cuda()
      with torch.no_grad():
          raw = net_f.query_rays(rays, z, rays[:, -3:])
      ref_raw = g["x_raw"].reshape(n, z.shape[1], -1)
      e = np.abs(R.npy(raw) - ref_raw)
      print(prec, "teacher-forced raw: rgb channels max", e[..., :3].max(), "(range", np.abs(ref_raw[..., :3]).max(), ") sigma max", e[..., 3].max(), "(range", np.abs(ref_raw[..., 3]).max(), ") sigma rel", (e[..., 3] / (np.abs(ref_raw[..., 3]) + 1.0)).max())
      rnd = chunked_pytest_randoms(n, int(g["chunk"]), 64, int(g["Nf"]), float(g["perturb"]), float(g["noise_std"]))
      with torch.no_grad():
          r2, d2, a2, w2, dp2, _ = S.raw2outputs(raw, z, rays[:, 3:6], white_bkgd=bool(g["white"]), noise=rnd["noise_f"].cuda() if rnd["noise_f"] is not None else None, rays=rays)
      print(prec, "teacher-forced rgb", d(r2, g["rgb"].reshape(n, 3)), "acc", d(a2, g["acc"].reshape(n)), "weights", d(w2, g["x_weights"].reshape(n, -1)),
            "opacity at the reference's half-way sample", (lambda cr, ch: float(np.abs(np.take_along_axis(ch, (cr >= 0.5).argmax(-1)[:, None], 1) - np.take_along_axis(cr, (cr >= 0.5).argmax(-1)[:, None], 1))[cr[:, -1] >= 0.5].max()))(np.cumsum(g["x_weights"].reshape(n, -1), -1), np.cumsum(R.npy(w2), -1)),
            "acc before the last sample", float(np.abs(R.npy(w2)[:, :-1].sum(-1) - g["x_weights"].reshape(n, -1)[:, :-1].sum(-1)).max()),
            "(reference range", float(g["x_weights"].reshape(n, -1)[:, :-1].sum(-1).min()), float(g["x_weights"].reshape(n, -1)[:, :-1].sum(-1).max()), ")",
            "depth", d(dp2, g["depth"].reshape(n)), "disp rel", float(np.nanmax(np.abs(R.npy(d2) - g["disp"].reshape(n)) / np.abs(g["disp"].reshape(n)))), "NaN pattern equal", bool((np.isnan(R.npy(d2)) == np.isnan(g["disp"].reshape(n))).all()))
      # gradients
      net_c, net_f, kw = R.build(S, g, prec)
      rgb, disp, acc, depth, ex = R.run(S, g, kw, True)
      target = T(g["target"]).cuda()
      loss = fixture_loss(g, lambda x: S.img2mse(x, target), rgb, ex["rgb0"], disp)
      print(prec, "loss", float(loss), "ref", float(g["loss"]))
      loss.backward()
      for pfx, net in (("gc_", net_c), ("gf_", net_f)):
          worst = 0
          for k, gr in net.named_views(net.flat.grad).items():
              if pfx + k not in g: continue
              gg = gr.reshape(-1).cpu()
              sub = gg[::61] if gg.numel() > 4096 else gg
              ref = torch.from_numpy(g[pfx + k])
              rel = float((sub - ref).norm() / (ref.norm() + 1e-30))
              nrm = abs(float(gg.double().norm()) / float(g[pfx + k + ".norm"]) - 1)
              worst = max(worst, rel)
              if rel > 1e-2: print("   ", prec, pfx + k, "rel", rel, "norm ratio-1", nrm)
          print(prec, pfx, "worst rel-L2 of the sampled gradient", worst)
