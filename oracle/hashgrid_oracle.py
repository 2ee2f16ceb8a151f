"""CPU oracle for the hash-grid network of BASELINE config 5 (NeRF_TCNN, DS_NeRF/run_nerf_helpers_tcnn.py:13-113).

TEST INFRASTRUCTURE ONLY (same rule as nerf_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may
import it).

Parity status: UNPINNED.  The arithmetic of this path lives in a third-party CUDA dependency that is not in the
reference tree and cannot run here: `tinycudann`, installed from
git+https://github.com/NVlabs/tiny-cuda-nn/#subdirectory=bindings/torch with NO version or commit pin
(requirements.txt:13); the reference has no test, fixture or golden vector for it.  What is restated
below is therefore tiny-cuda-nn's PUBLISHED algorithm (Müller et al., "Instant Neural Graphics Primitives with a
Multiresolution Hash Encoding", SIGGRAPH 2022, §3, and the library's documented encodings / networks), anchored on the
reference's own call sites for every configuration value:

  HashGrid encoding     n_levels 16, 2 features per level, log2_hashmap_size 19, base_resolution 16,
                        per_level_scale exp2(log2(2048 * 100 / 16) / 15)                       (tcnn.py:34-46)
      level l: scale = base * per_level_scale^l - 1, resolution = ceil(scale) + 1; a level is stored densely
      (index = x + y*res + z*res^2) while res^3 (rounded up to a multiple of 8) fits into 2^19 entries, else hashed with
      the paper's spatial hash  (x * 1) xor (y * 2654435761) xor (z * 805459861)  mod 2^19  (uint32 arithmetic);
      position = x * scale + 0.5, trilinear interpolation of the 8 surrounding entries; output = levels concatenated.
      Table initialised U(-1e-4, 1e-4).
  SphericalHarmonics    degree 4 = 16 real SH basis functions of the direction (2*d - 1)       (tcnn.py:64-70)
  FullyFusedMLP         ReLU hidden layers, no output activation, NO biases; every layer's input / output width
                        padded to a multiple of 16 — the 31-wide colour input is padded with the constant 1, the 3-wide
                        colour output to 16 rows of which 3 are read; Xavier-uniform initialisation  (tcnn.py:48-58, 74-84)
  forward               x01 = (x + 100) / 200 -> grid -> 32 -> 64 -> 16 (sigma = h[0], geo = h[1:16]);
                        d01 = (d + 1) / 2 -> SH(16); cat(SH, geo)(31) -> 64 -> 64 -> 3; out = cat(color, sigma)   (tcnn.py:86-113)

tiny-cuda-nn computes in fp16 with fp16 accumulation inside its fused MLP; this restatement is fp32 (and has a bf16
rounding emulation matching the HIP kernel's rounding points).  Agreement of the HIP path with THIS file is what the tests
establish; agreement with tiny-cuda-nn itself can only be claimed at the level of the published definition.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

N_LEVELS, N_FEAT, LOG2_T, BASE_RES, BOUND = 16, 2, 19, 16, 100.0
PER_LEVEL_SCALE = float(np.exp2(np.log2(2048 * BOUND / 16) / (16 - 1)))     # run_nerf_helpers_tcnn.py:34
PRIMES = (1, 2654435761, 805459861)


def level_table():
    """[(scale, resolution, n_entries, offset, hashed)] per level and the total number of entries."""
    out, off = [], 0
    for l in range(N_LEVELS):
        scale = float(np.exp2(l * np.log2(PER_LEVEL_SCALE)) * BASE_RES - 1.0)
        res = int(math.ceil(scale)) + 1
        dense = min(res ** 3, 2 ** 32 - 1)
        n = (dense + 7) // 8 * 8
        n = min(n, 1 << LOG2_T)
        out.append((np.float32(scale), res, n, off, res ** 3 > n))
        off += n
    return out, off


def init_params(seed=0) -> Dict[str, torch.Tensor]:
    """state-dict-shaped parameters: 'encoder.params' (grid, [entries * 2]), 'sigma_net.params' (32x64 + 64x16 rows
    [out][in]), 'encoder_dir.params' (empty), 'color_net.params' (32x64 + 64x64 + 64x16)."""
    rs = np.random.RandomState(seed)
    _, total = level_table()

    def xavier(fout, fin):
        s = math.sqrt(6.0 / (fin + fout))
        return rs.uniform(-s, s, size=(fout, fin)).astype(np.float32)
    grid = rs.uniform(-1e-4, 1e-4, size=total * N_FEAT).astype(np.float32)
    sig = np.concatenate([xavier(64, 32).ravel(), xavier(16, 64).ravel()])
    col = np.concatenate([xavier(64, 32).ravel(), xavier(64, 64).ravel(), xavier(16, 64).ravel()])
    return {"encoder.params": torch.from_numpy(grid), "sigma_net.params": torch.from_numpy(sig),
            "encoder_dir.params": torch.zeros(0), "color_net.params": torch.from_numpy(col)}


def to_unit_cube(x: torch.Tensor) -> torch.Tensor:
    """x = (x + bound) / (2 * bound) (run_nerf_helpers_tcnn.py:93) the way torch evaluates a division by a Python
    scalar on the GPU the reference runs on: one fp32 addition, then a multiplication by the fp32 reciprocal.  The last
    bit of this value is 1 % of a cell at the finest level (scale 2e5), so the kernel follows the same two roundings."""
    return (x.to(torch.float32) + np.float32(BOUND)) * np.float32(1.0 / (2 * BOUND))


def hash_encode(x01: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """x01 [N,3] in [0,1] -> [N, 32]; differentiable w.r.t. `grid` (index_select + weighted sum)."""
    levels, total = level_table()
    tab = grid.reshape(total, N_FEAT)
    outs = []
    x = x01.to(torch.float32)
    for scale, res, n, off, hashed in levels:
        lvl = tab[off:off + n]                 # (slicing first keeps autograd's scatter per level instead of per table)
        # tiny-cuda-nn forms the grid position with ONE rounding (fmaf(scale, x, 0.5f)); at the finest levels pos ~ 2e5 and
        # a second rounding would move the interpolation weights by up to 1 %: the fp64 product and sum are exact
        pos = (x.double() * float(scale) + 0.5).float()
        pg = torch.floor(pos)
        frac = pos - pg
        pg = pg.to(torch.int64)
        acc = torch.zeros(x.shape[0], N_FEAT, dtype=tab.dtype)
        for corner in range(8):
            o = [(corner >> d) & 1 for d in range(3)]
            c = [pg[:, d] + o[d] for d in range(3)]
            w = torch.ones(x.shape[0], dtype=torch.float32)
            for d in range(3):
                w = w * (frac[:, d] if o[d] else (1.0 - frac[:, d]))
            if hashed:
                h = torch.zeros_like(c[0])
                for d in range(3):
                    h = h ^ ((c[d] * PRIMES[d]) & 0xFFFFFFFF)
                idx = h % n
            else:
                idx = (c[0] + c[1] * res + c[2] * res * res) % n
            acc = acc + w.to(tab.dtype)[:, None] * lvl[idx]
        outs.append(acc)
    return torch.cat(outs, -1)


def sh4(d01: torch.Tensor) -> torch.Tensor:
    """degree-4 real spherical harmonics (16 values) of the direction 2*d01 - 1 (tiny-cuda-nn's SphericalHarmonics)."""
    v = d01 * 2.0 - 1.0
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [torch.full_like(x, 0.28209479177387814),
         -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
         1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
         -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
         0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
         0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
         0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
         0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return torch.stack(o, -1)


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nerf_tcnn_forward(sd: Dict[str, torch.Tensor], inp: torch.Tensor, bf16emu=False) -> torch.Tensor:
    """NeRF_TCNN.forward (run_nerf_helpers_tcnn.py:86-113): inp [N,6] = (position, unit direction) -> [N,4] =
    (color 3 — no activation, sigma).  `bf16emu` rounds the weights and every matrix-unit input to bf16 (fp32
    accumulation), the HIP kernel's rounding points."""
    q = _bf16 if bf16emu else (lambda t: t)
    x, d = inp[:, :3], inp[:, 3:]
    x01 = to_unit_cube(x)
    enc = q(hash_encode(x01, sd["encoder.params"]))
    ws = sd["sigma_net.params"]
    w1s, w2s = ws[:64 * 32].reshape(64, 32), ws[64 * 32:].reshape(16, 64)
    h = q(F.relu(F.linear(enc, q(w1s))))
    hs = F.linear(h, q(w2s))
    sigma, geo = hs[:, 0], hs[:, 1:]
    sh = sh4((d + 1) / 2)
    inc = q(torch.cat([sh, geo, torch.ones_like(geo[:, :1])], -1))        # 31 inputs + the constant-1 padding column
    wc = sd["color_net.params"]
    w1c, w2c, w3c = wc[:2048].reshape(64, 32), wc[2048:2048 + 4096].reshape(64, 64), wc[6144:].reshape(16, 64)
    h = q(F.relu(F.linear(inc, q(w1c))))
    h = q(F.relu(F.linear(h, q(w2c))))
    color = F.linear(h, q(w3c))[:, :3]
    return torch.cat([color, sigma[:, None]], -1)


def run_network(sd, inputs, viewdirs, bf16emu=False):
    """network_query_fn of create_nerf_tcnn (run_nerf.py:499-524): identity embedders, inputs [N,S,3], viewdirs [N,3]."""
    flat = inputs.reshape(-1, 3)
    dirs = viewdirs[:, None].expand(inputs.shape).reshape(-1, 3)
    out = nerf_tcnn_forward(sd, torch.cat([flat, dirs], -1), bf16emu)
    return out.reshape(list(inputs.shape[:-1]) + [4])
