"""Generate golden input/output vectors from the REFERENCE implementation.

Run ONLY in the build container (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports ``/root/reference/DS_NeRF`` read-only (stubbing the modules that are not
installed: cv2, torchvision, imageio, tkinter, lpips, tinycudann, configargparse,
tensorboard — recipe from SURVEY.md §8c), calls the reference functions on seeded
inputs and writes ``tests/golden/*.npz``.  Nothing of the reference's source is
stored: fixtures hold inputs and expected outputs only.  Network weights are not
stored either; they are re-derived from (seed, gain) by
``oracle.nerf_oracle.init_nerf_params`` / ``make_wild_params`` (numpy legacy
MT19937 stream, frozen by numpy's compatibility policy).
"""
import os
import sys
import types
import importlib.machinery

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    for n in ["cv2", "torchvision", "imageio", "tkinter", "lpips", "tinycudann", "configargparse"]:
        _stub(n)
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    torch.cuda.set_device = lambda *a, **k: None
    sys.path.insert(0, "/root/reference/DS_NeRF")
    import run_nerf_helpers as H
    import run_nerf as R
    torch.autograd.set_detect_anomaly(False)
    return H, R


def npz(name, out_dir=None, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(out_dir or HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def ref_net(H, sd, use_viewdirs=True, output_ch=4, input_ch=63, input_ch_views=27):
    net = H.NeRF(D=8, W=256, input_ch=input_ch, output_ch=output_ch, skips=[4],
                 input_ch_views=input_ch_views, use_viewdirs=use_viewdirs)
    missing = net.load_state_dict(sd, strict=False)
    # without viewdirs the reference still owns an (unused) views_linears.0; nothing else may differ
    assert not missing.unexpected_keys, missing
    return net


def main():
    H, R = import_reference()
    from oracle import nerf_oracle as O
    torch.manual_seed(0)
    rs = np.random.RandomState(1234)

    # ---------------- embedder (helpers:22-70) ----------------
    x = torch.from_numpy(rs.uniform(-4, 4, size=(64, 3)).astype(np.float32))
    x[0] = 0.0
    x[1] = torch.tensor([1.0, -1.0, 1.0])
    x[2] = torch.tensor([37.5, -12.25, 100.0])
    e10, d10 = H.get_embedder(10, 0)
    e4, d4 = H.get_embedder(4, 0)
    npz("embed", x=x, emb10=e10(x), emb4=e4(x), dim10=d10, dim4=d4)

    # ---------------- NeRF MLP forward (helpers:74-127) ----------------
    for tag, mk in [("default", lambda vd: O.init_nerf_params(seed=0, use_viewdirs=vd, output_ch=5 if not vd else 4, input_ch_views=27 if vd else 0)),
                    ("wild", lambda vd: O.make_wild_params(seed=1, use_viewdirs=vd, output_ch=5 if not vd else 4, input_ch_views=27 if vd else 0))]:
        for vd in (True, False):
            sd = mk(vd)
            net = ref_net(H, sd, use_viewdirs=vd, output_ch=5 if not vd else 4,
                          input_ch_views=27 if vd else 0)
            pts = torch.from_numpy(rs.uniform(-3, 3, size=(96, 3)).astype(np.float32))
            dirs = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(96, 3)).astype(np.float32)), dim=-1)
            xin = torch.cat([e10(pts), e4(dirs)], -1) if vd else e10(pts)
            with torch.no_grad():
                out = net(xin)
            npz(f"mlp_{tag}_{'vd' if vd else 'novd'}", pts=pts, dirs=dirs, x=xin, out=out,
                seed=0 if tag == "default" else 1, wild=int(tag == "wild"), use_viewdirs=int(vd))

    # ---------------- raw2outputs (helpers:350-401) ----------------
    def r2o_case(name, S, white, detach, noise_std, kind):
        N = 24
        raw = torch.from_numpy(rs.normal(scale=2.0, size=(N, S, 4)).astype(np.float32))
        if kind == "zero_sigma":
            raw[..., 3] = -1.0
        elif kind == "huge_sigma":
            raw[..., 3] = raw[..., 3].abs() * 200 + 50
        z = torch.sort(torch.from_numpy(rs.uniform(0.1, 6.0, size=(N, S)).astype(np.float32)), -1)[0]
        d = torch.from_numpy(rs.normal(scale=1.7, size=(N, 3)).astype(np.float32))
        raw_g = raw.clone().requires_grad_(True)
        rgb, disp, acc, w, depth, alpha = H.raw2outputs(raw_g, z, d, raw_noise_std=noise_std, white_bkgd=white,
                                                        pytest=True, need_alpha=True, detach_weights=detach)
        noise = None
        if noise_std > 0:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(N, S) * noise_std)
        # fixed upstream grads -> d raw
        g = {k: torch.from_numpy(rs.normal(size=tuple(v.shape)).astype(np.float32))
             for k, v in dict(rgb=rgb, disp=disp, acc=acc, w=w, depth=depth).items()}
        loss = sum((g[k] * v).sum() for k, v in dict(rgb=rgb, disp=disp, acc=acc, w=w, depth=depth).items())
        loss.backward()
        npz(name, raw=raw, z=z, d=d, noise=noise if noise is not None else np.zeros((0,), np.float32),
            white=int(white), detach=int(detach), rgb=rgb, disp=disp, acc=acc, w=w, depth=depth, alpha=alpha,
            g_rgb=g["rgb"], g_disp=g["disp"], g_acc=g["acc"], g_w=g["w"], g_depth=g["depth"], d_raw=raw_g.grad)

    r2o_case("r2o_s64", 64, False, False, 0.0, "normal")
    r2o_case("r2o_s192_white_noise", 192, True, False, 1.0, "normal")
    r2o_case("r2o_s192_detach", 192, True, True, 0.0, "normal")
    r2o_case("r2o_s64_zero_sigma", 64, True, False, 0.0, "zero_sigma")
    r2o_case("r2o_s64_huge_sigma", 64, False, False, 0.0, "huge_sigma")
    r2o_case("r2o_s5", 5, False, False, 0.5, "normal")

    # ---------------- sample_pdf (helpers:304-347) ----------------
    def pdf_case(name, nb, Nf, det, kind):
        N = 40
        bins = torch.sort(torch.from_numpy(rs.uniform(0.0, 5.0, size=(N, nb)).astype(np.float32)), -1)[0]
        w = torch.from_numpy(rs.uniform(0, 1, size=(N, nb - 1)).astype(np.float32))
        if kind == "delta":
            w = torch.zeros_like(w)
            w[torch.arange(N), torch.from_numpy(rs.randint(0, nb - 1, size=N))] = 5.0
        elif kind == "uniform":
            w = torch.ones_like(w)
        elif kind == "zeros":
            w = torch.zeros_like(w)
        out = H.sample_pdf(bins, w, Nf, det=det, pytest=True)
        if det:
            u = torch.Tensor(np.broadcast_to(np.linspace(0., 1., Nf), (N, Nf)).copy())
        else:
            np.random.seed(0)
            u = torch.Tensor(np.random.rand(N, Nf))
        npz(name, bins=bins, w=w, u=u, det=int(det), out=out)

    pdf_case("pdf_rand", 63, 128, False, "normal")
    pdf_case("pdf_det", 63, 128, True, "normal")
    pdf_case("pdf_delta", 63, 128, False, "delta")
    pdf_case("pdf_delta_det", 63, 64, True, "delta")
    pdf_case("pdf_uniform", 63, 128, True, "uniform")
    pdf_case("pdf_zeros", 63, 128, False, "zeros")
    pdf_case("pdf_small", 7, 5, False, "normal")

    # ---------------- rays (helpers:249-260, 283-300) ----------------
    Hh, Ww, focal = 12, 16, 20.0
    ang = 0.3
    c2w = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0.2], [0, 1, 0, -0.1], [-np.sin(ang), 0, np.cos(ang), 0.5]],
                       dtype=torch.float32)
    ro, rd = H.get_rays(Hh, Ww, focal, c2w)
    no, nd = H.ndc_rays(Hh, Ww, focal, 1., ro, rd)
    npz("rays", H=Hh, W=Ww, focal=focal, c2w=c2w, rays_o=ro, rays_d=rd, ndc_o=no, ndc_d=nd)

    # ---------------- end-to-end render() with pytest=True (run_nerf.py:90-165, 593-737) ----------------
    def render_case(name, ndc, lindisp, Nf, vd, perturb, noise_std, white, near, far, detach=False,
                    use_c2w=False, need_alpha=False, grads=False):
        Hh, Ww, focal = (10, 12, 15.0) if use_c2w else (18, 24, 30.0)
        och = 4 if vd else (5 if Nf > 0 else 4)
        sd_c = O.make_wild_params(seed=11, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0)
        sd_f = O.make_wild_params(seed=12, use_viewdirs=vd, output_ch=och, input_ch_views=27 if vd else 0) if Nf > 0 else None
        net_c = ref_net(H, sd_c, vd, och, input_ch_views=27 if vd else 0)
        net_f = ref_net(H, sd_f, vd, och, input_ch_views=27 if vd else 0) if Nf > 0 else None
        embed_fn, _ = H.get_embedder(10, 0)
        embeddirs_fn, _ = H.get_embedder(4, 0)

        def network_query_fn(inputs, viewdirs, network_fn):
            return R.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                 embeddirs_fn=embeddirs_fn if vd else None, netchunk=65536)

        kw = dict(network_query_fn=network_query_fn, perturb=perturb, N_importance=Nf, network_fine=net_f,
                  N_samples=64, network_fn=net_c, use_viewdirs=vd, white_bkgd=white, raw_noise_std=noise_std,
                  ndc=ndc, near=near, far=far)
        if not ndc:
            kw['lindisp'] = lindisp
        ro, rd = H.get_rays(Hh, Ww, focal, c2w)
        if ndc:
            # LLFF-style: camera looking down -z from z>0 so NDC is well defined
            pass
        sel = torch.from_numpy(rs.permutation(Hh * Ww)[:48])
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
        if use_c2w:
            out = R.render(Hh, Ww, focal, chunk=100, c2w=c2w[:3, :4], retraw=True, pytest=True,
                           need_alpha=need_alpha, detach_weights=detach, **kw)
        else:
            out = R.render(Hh, Ww, focal, chunk=20, rays=rays, retraw=True, pytest=True,
                           need_alpha=need_alpha, detach_weights=detach, **kw)
        rgb, disp, acc, depth, extras = out
        arrs = dict(H=Hh, W=Ww, focal=focal, c2w=c2w, rays=rays, ndc=int(ndc), lindisp=int(lindisp), Nf=Nf,
                    vd=int(vd), perturb=perturb, noise_std=noise_std, white=int(white), near=near, far=far,
                    detach=int(detach), use_c2w=int(use_c2w), need_alpha=int(need_alpha), och=och,
                    chunk=100 if use_c2w else 20,
                    rgb=rgb, disp=disp, acc=acc, depth=depth)
        for k, v in extras.items():
            arrs["x_" + k] = v
        if grads:
            target = torch.from_numpy(rs.uniform(0, 1, size=tuple(rgb.shape)).astype(np.float32))
            loss = H.img2mse(rgb, target)
            if 'rgb0' in extras:
                loss = loss + H.img2mse(extras['rgb0'], target)
            loss = loss + 0.1 * H.img2mse(disp, torch.zeros_like(disp))
            loss.backward()
            arrs["target"] = target
            arrs["loss"] = loss
            for pfx, net in (("gc_", net_c), ("gf_", net_f)):
                if net is None:
                    continue
                for k, p in net.named_parameters():
                    if p.grad is not None:
                        # big matrices: a fixed stride-61 subsample + the L2 norm keep fixtures small
                        g = p.grad.reshape(-1)
                        arrs[pfx + k] = g[::61] if g.numel() > 4096 else g
                        arrs[pfx + k + ".norm"] = g.double().norm()
        npz(name, **arrs)

    # NOTE on chunking + pytest: every render_rays chunk re-seeds numpy, so chunk c of size n sees
    # rand(n, S) from seed 0; test code must rebuild the randoms per chunk the same way.
    render_case("render_ndc_fine_vd", True, False, 128, True, 1.0, 1.0, False, 0., 1., grads=True)
    render_case("render_lindisp_fine_vd", False, True, 128, True, 1.0, 1.0, True, 1.2, 9.0, grads=True)
    render_case("render_lindisp_fine_vd_detach", False, True, 128, True, 1.0, 0.0, True, 1.2, 9.0, detach=True, grads=True)
    render_case("render_ndc_coarse_vd", True, False, 0, True, 1.0, 0.0, False, 0., 1., grads=True)
    render_case("render_noperturb_fine_vd_alpha", False, False, 128, True, 0.0, 0.0, True, 0.5, 6.0, need_alpha=True)
    render_case("render_ndc_fine_novd", True, False, 64, False, 1.0, 1.0, False, 0., 1.)
    render_case("render_c2w_fine_vd", False, True, 128, True, 0.0, 0.0, True, 1.2, 9.0, use_c2w=True)


if __name__ == "__main__":
    main()
