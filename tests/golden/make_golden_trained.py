"""Golden fixture on GENUINELY TRAINED networks, generated from the REFERENCE.

Run ONLY in the build container (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden_trained.py

The reference's own ``NeRF`` modules (coarse + fine, default init, seeds 0 / 1) are trained with the reference's own
``render()`` + ``img2mse`` and ``torch.optim.Adam(lr=5e-4)`` for 200 steps of 192 rays x (64 + 64) samples on an analytic
scene (a normal-shaded unit sphere in front of a white background, six cameras on a ring; ``raw_noise_std=1`` and
``perturb=1`` like the reference's configs).  Then the weights are ROUNDED TO BF16 — the fixture stores them as uint16 bit
patterns, half the size, and every precision mode of the HIP path sees exactly the weights the reference rendered with —
loaded back into the reference modules, and one ``render()`` of 48 rays (``pytest=True``: the reference's deterministic
random draws) is recorded with its parameter gradients, exactly like ``make_golden.py``'s render cases.

Nothing of the reference's source is stored: inputs, weights the reference trained, and the reference's outputs.

Round 4: the run is REGENERABLE — ``torch.manual_seed`` fixes the reference's ``torch.rand`` /
``torch.randn`` draws inside ``render()`` and the whole script runs on ONE thread (multi-threaded reductions are
order-dependent) — and it writes a SECOND fixture, ``render_trained_black_vd``: the same sphere in front of a BLACK
background without ``white_bkgd``, where rays that miss or graze the sphere end with ``acc < 1`` (on the white scene the
networks learn white fog: ``acc == 1`` on every ray, so nothing there tests a semi-transparent or empty ray).
``python tests/golden/make_golden_trained.py --check`` regenerates both into a temporary directory and compares them with
the committed files.
"""
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from make_golden import import_reference, npz

Hh, Ww, FOCAL, NEAR, FAR = 96, 128, 230.0, 2.0, 6.0


def sphere_scene(rays_o, rays_d, white=True):
    d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    b = (rays_o * d).sum(-1)
    c = (rays_o * rays_o).sum(-1) - 1.0
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp(min=0))
    n = rays_o + d * t[..., None]
    col = 0.5 + 0.5 * n
    return torch.where(hit[..., None], col, torch.ones_like(col) if white else torch.zeros_like(col))


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)


def make(name, white, out_dir, n_steps=200, seed=1234, noise_std=1.0, disp_loss=True):
    """Train the reference's modules on the sphere scene (white or black background), store weights + one render()."""
    H, R = import_reference()
    from oracle import nerf_oracle as O
    # multi-threaded reductions are order-dependent: one thread regenerates bit for bit (GOLDEN_THREADS: exploration only)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "1")))
    torch.manual_seed(seed)           # the reference's render() draws torch.rand / torch.randn from the global stream
    np.random.seed(seed)
    nets = []
    for s in (0, 1):
        net = H.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        net.load_state_dict(O.init_nerf_params(seed=s))
        nets.append(net)
    net_c, net_f = nets
    embed_fn, _ = H.get_embedder(10, 0)
    embeddirs_fn, _ = H.get_embedder(4, 0)

    def network_query_fn(inputs, viewdirs, network_fn):
        return R.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=65536)

    kw = dict(network_query_fn=network_query_fn, perturb=1.0, N_importance=64, network_fine=net_f, N_samples=64,
              network_fn=net_c, use_viewdirs=True, white_bkgd=bool(white), raw_noise_std=noise_std, ndc=False, near=NEAR, far=FAR,
              lindisp=False)
    rays_all, tgt_all = [], []
    for k in range(6):
        a = 2 * math.pi * k / 6
        eye = torch.tensor([4 * math.sin(a), 0.6, 4 * math.cos(a)])
        z = eye / eye.norm()
        x = torch.linalg.cross(torch.tensor([0., 1., 0.]), z); x = x / x.norm()
        y = torch.linalg.cross(z, x)
        c2w = torch.cat([torch.stack([x, y, z], 1), eye[:, None]], 1)
        ro, rd = H.get_rays(Hh, Ww, FOCAL, c2w)
        rays_all.append(torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0))
        tgt_all.append(sphere_scene(ro.reshape(-1, 3), rd.reshape(-1, 3), white))
    rays_all = torch.cat(rays_all, 1)
    tgt_all = torch.cat(tgt_all, 0)
    opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=5e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(77)
    n_rand = 192
    psnr = []
    for it in range(n_steps):
        sel = torch.randint(0, rays_all.shape[1], (n_rand,), generator=g)
        rgb, disp, acc, depth, extras = R.render(Hh, Ww, FOCAL, chunk=4096, rays=rays_all[:, sel], retraw=True, **kw)
        loss = H.img2mse(rgb, tgt_all[sel]) + H.img2mse(extras['rgb0'], tgt_all[sel])
        opt.zero_grad()
        loss.backward()
        opt.step()
        psnr.append(float(-10 * torch.log10(H.img2mse(rgb, tgt_all[sel]))))
        if it % 20 == 0 or it == n_steps - 1:
            print(f"[{name}] step {it:3d} loss {float(loss):.4f} psnr {psnr[-1]:.2f} dB", flush=True)

    # ---- weights -> bf16, reloaded into the reference ----
    arrs = {}
    for pfx, net in (("wc_", net_c), ("wf_", net_f)):
        sd = {k: v.detach().to(torch.bfloat16).float() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        for k, v in sd.items():
            arrs[pfx + k] = bf16_bits(v)
    # ---- one deterministic render + gradients, like make_golden.py's render cases ----
    rs = np.random.RandomState(4321)
    sel = torch.from_numpy(rs.permutation(rays_all.shape[1])[:48])   # rays that see the sphere, graze it, and miss it
    rays = rays_all[:, sel]
    kw_fix = dict(kw, N_importance=128)
    for p in list(net_c.parameters()) + list(net_f.parameters()):
        p.grad = None
    rgb, disp, acc, depth, extras = R.render(Hh, Ww, FOCAL, chunk=20, rays=rays, retraw=True, pytest=True, **kw_fix)
    target = tgt_all[sel]
    loss = H.img2mse(rgb, target) + H.img2mse(extras['rgb0'], target)
    if disp_loss:   # (a ray that hits nothing has acc = 0 and disp = 1 / (0 / 0) = NaN in the reference: no disparity term then)
        loss = loss + 0.1 * H.img2mse(disp, torch.zeros_like(disp))
    loss.backward()
    arrs.update(H=Hh, W=Ww, focal=FOCAL, rays=rays, ndc=0, lindisp=0, Nf=128, vd=1, perturb=1.0, noise_std=noise_std,
                white=int(bool(white)), disp_loss=int(bool(disp_loss)), near=NEAR, far=FAR, detach=0, use_c2w=0, need_alpha=0, och=4, chunk=20, rgb=rgb,
                disp=disp, acc=acc, depth=depth, target=target, loss=loss, train_psnr_last20=float(np.mean(psnr[-20:])),
                train_steps=n_steps, torch_seed=seed)
    for k, v in extras.items():
        arrs["x_" + k] = v
    for pfx, net in (("gc_", net_c), ("gf_", net_f)):
        for k, p in net.named_parameters():
            if p.grad is not None:
                gr = p.grad.reshape(-1)
                arrs[pfx + k] = gr[::61] if gr.numel() > 4096 else gr
                arrs[pfx + k + ".norm"] = gr.double().norm()
    npz(name, out_dir=out_dir, **arrs)
    raw = extras['raw']
    print(f"{name}: last-20 train PSNR {np.mean(psnr[-20:]):.2f} dB; fixture raw range [{float(raw.min()):.2f}, {float(raw.max()):.2f}], "
          f"acc in [{float(acc.min()):.3f}, {float(acc.max()):.3f}], rays with acc < 0.99: {int((acc < 0.99).sum())} / {acc.numel()}, "
          f"with acc < 0.5: {int((acc < 0.5).sum())}")


# name, white background, raw_noise_std (training and fixture render), disparity term in the fixture's loss
CASES = (("render_trained_fine_vd", True, 1.0, True), ("render_trained_black_vd", False, 0.0, False))


def main():
    check = "--check" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_dir = HERE
    if check:
        import tempfile
        out_dir = tempfile.mkdtemp(prefix="golden_trained_")
    for name, white, noise_std, disp_loss in CASES:
        if only and name not in only:
            continue
        make(name, white, out_dir, noise_std=noise_std, disp_loss=disp_loss)
        if check:
            a, b = np.load(os.path.join(out_dir, name + ".npz")), np.load(os.path.join(HERE, name + ".npz"))
            assert set(a.files) == set(b.files), (name, set(a.files) ^ set(b.files))
            worst = 0.0
            for k in a.files:
                x, y = a[k].astype(np.float64), b[k].astype(np.float64)
                # rays that hit nothing have NaN disparities, as in the reference (run_nerf_helpers.py:388): the NaN
                # patterns must agree, the finite elements are compared
                assert x.shape == y.shape and np.array_equal(np.isnan(x), np.isnan(y)), (name, k, "NaN pattern")
                fin = ~np.isnan(y)
                x, y = x[fin], y[fin]
                d = float(np.abs(x - y).max() / max(float(np.abs(y).max()), 1e-30)) if x.size else 0.0
                worst = max(worst, d)
                assert d <= 1e-5, (name, k, d)
            print(f"{name}: regenerated == committed (largest relative difference {worst:.1e} over {len(a.files)} arrays)")


if __name__ == "__main__":
    main()
