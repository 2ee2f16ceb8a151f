"""GPU: size-independent properties at BASELINE.json's full sizes (1024 rays x (64 + 128) samples = 196 608 +
65 536 MLP evaluations per step, bf16), where the CPU oracle would take minutes:

  * tiling invariance — the fused kernel on 196 608 samples in one launch equals the same samples pushed
    through in ragged pieces (different workgroup / tile assignment, padding tails);
  * determinism — two launches give bit-identical outputs and gradients (no atomics anywhere);
  * linearity of the backward in d raw — grad(a * d) = a * grad(d) and grad(d1 + d2) = grad(d1) + grad(d2)
    up to bf16 rounding of d z (the forward activations and ReLU flags are shared);
  * a slice of the big launch equals the CPU oracle's bf16 emulation on that slice (same rounding points);
  * the full training step at bench size: finite, loss goes down, both networks move.
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from helpers import T

pytestmark = pytest.mark.gpu

N_RAYS, N_C, N_F = 1024, 64, 128
M_FINE = N_RAYS * (N_C + N_F)


@pytest.fixture(scope="module")
def S():
    import spin_nerf_amd as S
    assert torch.cuda.is_available()
    S._lib.load()
    return S


def _net(S, seed, precision="bf16"):
    sd = O.make_wild_params(seed=seed)
    net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision=precision).cuda()
    net.load_state_dict(sd)
    return sd, net


def _inputs(seed, n_rays=N_RAYS, s=N_C + N_F):
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(n_rays, s, 3, generator=g) * 4 - 2).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1).cuda()
    return pts, dirs


def test_forward_is_tiling_invariant_and_deterministic(S):
    _, net = _net(S, 11)
    pts, dirs = _inputs(1)
    with torch.no_grad():
        whole = net.query(pts, dirs)
        again = net.query(pts, dirs)
        assert torch.equal(whole, again)
        # ragged pieces: 1, 7, 250, 33 rays ... (tile tails of 192 * k mod 256 samples)
        cuts = [0, 1, 8, 258, 291, 700, 1023, N_RAYS]
        parts = [net.query(pts[a:b], dirs[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(whole, torch.cat(parts, 0))
    assert torch.isfinite(whole).all()


def test_slice_of_full_launch_matches_bf16_emulation(S):
    sd, net = _net(S, 12)
    pts, dirs = _inputs(2)
    with torch.no_grad():
        whole = net.query(pts, dirs)
    rows = [0, 1, 511, 1023]
    p = pts[rows].cpu()
    d = dirs[rows].cpu()
    x = torch.cat([O.embed(p.reshape(-1, 3), 10), O.embed(d[:, None].expand(p.shape).reshape(-1, 3), 4)], -1)
    ref = O.nerf_forward_bf16emu(sd, x, use_viewdirs=True).reshape(len(rows), -1, 4)
    got = whole[rows].cpu()
    # same rounding points; what differs is fp32 summation order inside the MFMA and the hardware sin/cos
    err = (got - ref).abs()
    scale = ref.abs().mean()
    assert float(err.mean() / scale) < 2e-2, float(err.mean() / scale)
    assert float(err.max() / scale) < 0.5, float(err.max() / scale)


def _grad(net, pts, dirs, d_raw):
    net.flat.grad = None
    out = net.query(pts, dirs)
    out.backward(d_raw)
    return net.flat.grad.clone()


def test_backward_is_deterministic_and_linear_in_d_raw(S):
    _, net = _net(S, 13)
    pts, dirs = _inputs(3)
    g = torch.Generator().manual_seed(4)
    d1 = torch.randn(N_RAYS, N_C + N_F, 4, generator=g).cuda()
    d2 = torch.randn(N_RAYS, N_C + N_F, 4, generator=g).cuda()
    g1 = _grad(net, pts, dirs, d1)
    assert torch.equal(g1, _grad(net, pts, dirs, d1))          # no atomics, fixed split-K order
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    # exact scaling by a power of two (bf16 rounding commutes with it)
    assert torch.equal(_grad(net, pts, dirs, 4.0 * d1), 4.0 * g1)
    # additivity up to the bf16 rounding of d z at every layer
    g2 = _grad(net, pts, dirs, d2)
    g12 = _grad(net, pts, dirs, d1 + d2)
    rel = float((g12 - (g1 + g2)).norm() / (g1 + g2).norm())
    assert rel < 1e-2, rel


@pytest.mark.parametrize("precision,gate", [("fp32", 1e-5), ("bf16", 6e-3)])
def test_split_gradient_equals_whole(S, precision, gate):
    """Backward of 196 608 samples in one launch vs accumulated over two halves (a different split-K partition of the
    samples).  fp32 mode keeps its split-K partial sums in fp32: equal up to summation order (1e-5) — the strict check of
    the split-K machinery, which both modes share.  bf16 mode rounds every partial sum once to bf16 (half the bytes of the
    partial-sum round trip): with the random gradients of this test, whose per-sample contributions cancel almost
    completely, that shows as 2.6e-3 of the gradient's norm — unbiased, and far inside the 1-3 % by which the bf16
    network's gradient differs from the fp32 one."""
    _, net = _net(S, 14, precision)
    n_rays = N_RAYS if precision == "bf16" else N_RAYS // 4          # (the fp32 MFMA path is 16x slower)
    pts, dirs = _inputs(5, n_rays)
    d = torch.randn(n_rays, N_C + N_F, 4, generator=torch.Generator().manual_seed(6)).cuda()
    whole = _grad(net, pts, dirs, d)
    net.flat.grad = None
    h = n_rays // 2
    net.query(pts[:h], dirs[:h]).backward(d[:h])
    net.query(pts[h:], dirs[h:]).backward(d[h:])
    halves = net.flat.grad.clone()
    rel = float((whole - halves).norm() / whole.norm())
    assert rel < gate, rel


def _oracle_bf16_grads(sd, pts, dirs, d_raw):
    """autograd through the oracle's bf16-rounding emulation (DS_NeRF/run_nerf_helpers.py:104-127 with the kernels' rounding
    points) on the CPU: 196 608 samples take ~10 s on the test box's cores"""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    for v in sd.values():
        v.requires_grad_(True)
        v.grad = None
    ref = O.run_network(sd, pts.cpu(), dirs.cpu(), use_viewdirs=True, mlp=O.nerf_forward_bf16emu)
    (ref * d_raw.cpu()).sum().backward()
    return {k: v.grad for k, v in sd.items()}


def _assert_grads_close(net, ref, gate, what):
    got = net.named_views(net.flat.grad)
    worst = 0.0
    for k, g in ref.items():
        if g is None:
            continue
        a, b = got[k].detach().cpu().double().reshape(-1), g.double().reshape(-1)
        rel = float((a - b).norm() / b.norm())
        worst = max(worst, rel)
        assert rel < gate, f"{what}: {k}: relative L2 error vs the oracle's bf16 emulation {rel:.2e}"
    return worst


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_s", [N_C + N_F, N_C])
def test_bench_size_bf16_backward_matches_the_oracle(S, n_s):
    """the weight-gradient launch AT THE BENCH'S SIZE — 1024 rays x 192 samples (6144 tiles: the
    real slot apportionment, pacing and plain-workgroup shares of mlp_wgrad_pair.h) and the coarse 1024 x 64 launch —
    compared per parameter tensor with autograd through the oracle's bf16 emulation: relative L2 <= 5e-2, the gate of the
    small-size tests (tests/test_gpu_kernels.py: test_mlp_backward_bf16)."""
    sd, net = _net(S, 21)
    pts, dirs = _inputs(31, N_RAYS, n_s)
    # (+ 0.25: a zero-mean d raw makes the bias gradients — plain sums of the incoming gradients — cancel to ~sqrt(N), and the
    #  relative error of such a sum measures the draw, not the kernel)
    d = (torch.randn(N_RAYS, n_s, 4, generator=torch.Generator().manual_seed(32)) + 0.25).cuda()
    ref = _oracle_bf16_grads(sd, pts, dirs, d)
    _grad(net, pts, dirs, d)
    assert torch.isfinite(net.flat.grad).all()
    _assert_grads_close(net, ref, 5e-2, f"1024 x {n_s}")


@pytest.mark.timeout(900)
def test_bench_size_merged_backward_of_both_networks_matches_the_oracle(S):
    """The training step's real launch sequence: coarse (1024 x 64) and fine (1024 x 192) backward passes as ONE
    snr_mlp_backward_multi call — one chain launch, one weight-gradient launch (both networks' layer pairs on the pair slots,
    their plain jobs on the plain workgroups), one reduce — against the oracle's bf16 emulation per network and tensor
    (5e-2), and against the same two passes run one network at a time (a different split of the samples over the
    workgroups: bf16 partial sums, 6e-3 of the norm)."""
    ops = S.ops
    sd_f, net_f = _net(S, 22)
    sd_c, net_c = _net(S, 23)
    pts_f, dirs = _inputs(33, N_RAYS, N_C + N_F)
    pts_c = pts_f[:, :N_C].contiguous()
    g = torch.Generator().manual_seed(34)
    d_f = (torch.randn(N_RAYS, N_C + N_F, 4, generator=g) + 0.25).cuda()   # (non-zero mean: see the test above)
    d_c = (torch.randn(N_RAYS, N_C, 4, generator=g) + 0.25).cuda()
    L = S._lib
    lib = L.load()

    def fwd(net, pts):
        n = pts.shape[0] * pts.shape[1]
        raw = torch.empty(n, 4, device="cuda")
        act = torch.empty(lib.snr_mlp_act_bytes(net.cfg, n), dtype=torch.uint8, device="cuda")
        packed = net.packed_weights()
        L.check(lib.snr_mlp_forward(net.cfg, L.ptr(packed), L.ptr(pts.reshape(-1, 3).contiguous()), None, 0, None, L.ptr(dirs), 3, n,
                                    pts.shape[1], L.ptr(raw), L.ptr(act), L.stream()), "snr_mlp_forward")
        return (packed, act, n)
    sv_f, sv_c = fwd(net_f, pts_f), fwd(net_c, pts_c)
    g_f, g_c = ops.mlp_train_backward_multi([net_f, net_c], [sv_f, sv_c], [d_f.reshape(-1, 4), d_c.reshape(-1, 4)])
    one_f = ops.mlp_train_backward(net_f, sv_f, d_f.reshape(-1, 4))
    one_c = ops.mlp_train_backward(net_c, sv_c, d_c.reshape(-1, 4))
    torch.cuda.synchronize()
    assert torch.isfinite(g_f).all() and torch.isfinite(g_c).all()
    for merged, single in ((g_f, one_f), (g_c, one_c)):
        assert float((merged - single).norm() / single.norm()) < 6e-3
    for net, sd, pts, d, grad, what in ((net_f, sd_f, pts_f, d_f, g_f, "fine"), (net_c, sd_c, pts_c, d_c, g_c, "coarse")):
        ref = _oracle_bf16_grads(sd, pts, dirs, d)
        net.flat.grad = grad
        _assert_grads_close(net, ref, 5e-2, what + " network of the merged launch")


def test_training_step_at_bench_size(S):
    import argparse, contextlib, io, tempfile
    RenderTrainer = importlib.import_module("spin-nerf_amd.train").RenderTrainer
    dev = torch.device("cuda")
    torch.manual_seed(0)
    args = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=N_F, N_samples=N_C,
        alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536,
        lrate=5e-4, basedir=tempfile.mkdtemp(), expname="", ft_path=None, no_reload=True, perturb=1.0,
        white_bkgd=True, raw_noise_std=1.0, dataset_type="llff", no_ndc=True, lindisp=True, sigma_loss=False,
        no_coarse=False, precision="bf16")
    with contextlib.redirect_stdout(io.StringIO()):
        kw_train, *_ = S.create_nerf(args, device=dev)
    kw_train.update(near=1.2, far=9.0)
    tr = RenderTrainer(kw_train, lrate=5e-4, lrate_decay=250)
    H, W, focal = 378, 504, 400.0
    ro, rd = S.get_rays(H, W, focal, torch.eye(4, device=dev)[:3, :4])
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(5))[:N_RAYS].to(dev)
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).contiguous()
    target = torch.full((N_RAYS, 3), 0.25, device=dev)
    before = [n.flat.detach().clone() for n in tr.nets]
    losses = []
    for _ in range(30):
        loss, rgb = tr.step(H, W, focal, rays, target)
        losses.append(float(loss))
        assert rgb.shape == (N_RAYS, 3)
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    for n, b in zip(tr.nets, before):
        assert torch.isfinite(n.flat).all() and float((n.flat.detach() - b).abs().max()) > 0


def test_one_launch_beyond_4_gib_of_saved_activations(S):
    """1 048 576 samples in one launch: 5.7 GB of saved activations and 5.2 GB of d z, i.e. section offsets
    past 2^32 in the asm stores, the dgrad flag loads and the wgrad DMA addresses.  Must equal the same
    samples pushed through in four launches (forward bit-exact; gradients up to the rounding of the bf16 split-K partial
    sums — see test_split_gradient_equals_whole — where an addressing error would show as O(1))."""
    _, net = _net(S, 15)
    n_rays, s = 4096, 256
    pts, dirs = _inputs(7, n_rays, s)
    d = torch.randn(n_rays, s, 4, generator=torch.Generator().manual_seed(8)).cuda()
    lib = S._lib.load()
    assert lib.snr_mlp_act_bytes(net.cfg, n_rays * s) > 2 ** 32
    net.flat.grad = None
    whole = net.query(pts, dirs)
    whole.backward(d)
    g_whole = net.flat.grad.clone()
    net.flat.grad = None
    parts = []
    for a in range(0, n_rays, 1024):
        o = net.query(pts[a:a + 1024], dirs[a:a + 1024])
        o.backward(d[a:a + 1024])
        parts.append(o.detach())
    assert torch.equal(whole.detach(), torch.cat(parts, 0))
    rel = float((g_whole - net.flat.grad).norm() / g_whole.norm())
    assert torch.isfinite(g_whole).all() and rel < 6e-3, rel
