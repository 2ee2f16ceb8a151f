"""Diagnostic: run-to-run reproducibility of the hash-grid backward on fixed inputs.  The MLP weight gradients involve no
atomics (split-K partial sums reduced in a fixed order), so they must be bit-identical from run to run; any difference or
non-finite value is a race or a read of uninitialised memory.  The table gradient (atomics) is checked for finiteness and
against the first run to 1e-4 relative."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spin_nerf_amd as S
from oracle import hashgrid_oracle as H
dev = torch.device("cuda")
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S_ = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sd = H.init_params(3)
sd["encoder.params"] = sd["encoder.params"] * 3e3
net = S.NeRF_TCNN().to(dev)
net.load_state_dict(sd)
rs = np.random.RandomState(0)
rays = torch.zeros(n_rays, 11)
rays[:, :3] = torch.from_numpy(rs.normal(scale=0.3, size=(n_rays, 3)).astype(np.float32)) + torch.tensor([0., 0., 4.])
d = torch.from_numpy((rs.normal(size=(n_rays, 3)) * [0.3, 0.3, 0.1] + [0, 0, -1]).astype(np.float32))
rays[:, 3:6] = d
rays[:, 6], rays[:, 7] = 2.0, 6.0
rays[:, 8:11] = d / d.norm(dim=-1, keepdim=True)
rays = rays.to(dev)
z = torch.sort(torch.from_numpy(rs.uniform(2.0, 6.0, size=(n_rays, S_)).astype(np.float32)), -1)[0].to(dev)
d_raw = torch.from_numpy(rs.normal(scale=1e-4, size=(n_rays, S_, 4)).astype(np.float32)).to(dev)
vd = rays[:, -3:]
g_entries = net.table_entries * 2
ref = None
bad = 0
for r in range(reps):
    # dirty the allocator's free blocks so that stale-memory reads show up as garbage, not as the previous (correct) values
    junk = torch.full((64 << 20,), float("nan"), device=dev)
    del junk
    raw, sv = net.train_forward(rays, z, vd)
    g = net.train_backward(sv, d_raw)
    mlp = g[g_entries:]
    if ref is None:
        ref_raw, ref, ref_tab = raw.clone(), mlp.clone(), g[:g_entries].clone()
        continue
    fin = bool(torch.isfinite(g).all())
    same = bool(torch.equal(mlp, ref)) and bool(torch.equal(raw, ref_raw))
    tab_ok = float((g[:g_entries] - ref_tab).abs().max()) <= 1e-4 * float(ref_tab.abs().max())
    if not (fin and same and tab_ok):
        bad += 1
        diff = (mlp != ref) | ~torch.isfinite(mlp)
        idx = torch.nonzero(diff).flatten()
        print(f"rep {r}: finite {fin} mlp identical {same} table ok {tab_ok}; {int(diff.sum())} differing MLP entries, first at "
              f"{idx[:6].tolist()} (sigma block 0..3071, color block 3072..10239); raw identical {bool(torch.equal(raw, ref_raw))}", flush=True)
        if bad >= 10:
            break
print(f"{reps} repetitions at {n_rays} x {S_}: {bad} bad")
