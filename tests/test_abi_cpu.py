"""CPU: the C-ABI shared library loads without a GPU and exports every symbol that
include/spinnerf_hip.h declares; size queries and argument validation work host-side; the Python
host fails loudly (no fallback) when asked to compute without a device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def S():
    import __graft_entry__ as g
    import spin_nerf_amd as S
    if not os.path.exists(S.LIB_PATH):
        g.build()
    return S


def declared_functions():
    src = open(os.path.join(ROOT, "include", "spinnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snr_[a-z0-9_]+)\s*\(", src)) - {"snr_mlp_config"})


def test_library_exports_every_declared_symbol(S):
    lib = ctypes.CDLL(S.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/spinnerf_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(S._lib.SIGNATURES) == names


def test_size_queries_and_validation_run_on_host(S):
    lib = S._lib.load()
    assert lib.snr_abi_version() == 4 == S._lib.ABI_VERSION
    cfg = S._lib.MlpConfig(10, 4, 0, 1, 4, S._lib.PREC_BF16)
    assert lib.snr_mlp_param_count(cfg) == 595844          # SURVEY.md §8 a6 [measured]
    assert lib.snr_mlp_packed_bytes(cfg) > 2 * 595844 * 2  # forward + transposed copies, bf16
    n = 1024 * 192
    assert lib.snr_mlp_act_bytes(cfg, n) > 0 and lib.snr_mlp_bwd_ws_bytes(cfg, n) > 0
    assert lib.snr_mlp_act_bytes(cfg, 0) == -2             # SNR_ERR_SHAPE
    bad = S._lib.MlpConfig(11, 4, 0, 1, 4, 0)
    assert lib.snr_mlp_param_count(bad) == -3              # SNR_ERR_UNSUPPORTED
    assert lib.snr_mlp_pack(cfg, None, None, None) == -1   # SNR_ERR_NULL
    assert lib.snr_sample_coarse(None, 11, 4, 64, 0, None, None, None) == -1
    assert lib.snr_pack_rays(None, None, None, 8, 4, 4, 1.0, 0, 1.0, 0.0, 1.0, None, None, None, 1, None, 11, None) == -1
    assert lib.snr_mse_pair(None, None, None, 8, None, None, None, None) == -1
    import ctypes
    one = ctypes.c_void_p(16)   # any non-null address: argument checks come before any device access
    assert lib.snr_pack_rays(one, one, None, 8, 4, 4, 1.0, 0, 1.0, 0.0, 1.0, None, None, None, 1, one, 8, None) == -2    # row too short for viewdirs
    assert lib.snr_pack_rays(one, one, None, 8, 4, 4, 1.0, 0, 1.0, 0.0, 1.0, None, None, one, 1, one, 11, None) == -2  # ... for depth + viewdirs
    assert lib.snr_pack_rays(one, one, None, 0, 4, 4, 1.0, 0, 1.0, 0.0, 1.0, None, None, None, 0, one, 8, None) == -2    # empty input
    assert lib.snr_embed(None, 4, 3, 10, one, None) == -1 and lib.snr_embed(one, 0, 3, 10, one, None) == -2
    assert lib.snr_mse_pair(one, None, one, 0, one, one, None, None) == -2
    assert lib.snr_mse_pair(one, one, one, 8, one, one, None, None) == -1                   # b without grad_b
    assert b"NULL" in lib.snr_status_string(-1)


def test_module_matches_reference_layout(S):
    """state-dict keys/shapes of the reference NeRF (helpers:86-102); 595 844 parameters."""
    torch.manual_seed(0)
    m = S.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    sd = m.state_dict()
    assert list(sd)[:4] == ["pts_linears.0.weight", "pts_linears.0.bias", "pts_linears.1.weight", "pts_linears.1.bias"]
    assert sd["pts_linears.5.weight"].shape == (256, 319) and sd["views_linears.0.weight"].shape == (128, 283)
    assert sd["alpha_linear.weight"].shape == (1, 256) and sd["rgb_linear.weight"].shape == (3, 128)
    assert sum(v.numel() for v in sd.values()) == 595844 == m.flat.numel()
    # same seed -> same init as a chain of nn.Linear built in the reference's order
    torch.manual_seed(0)
    first = torch.nn.Linear(63, 256)
    assert torch.equal(sd["pts_linears.0.weight"], first.weight)
    # round trip + strictness
    m2 = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)
    m2.load_state_dict(sd)
    assert torch.equal(m2.flat, m.flat)
    with pytest.raises(RuntimeError):
        m2.load_state_dict({k: v for k, v in sd.items() if k != "rgb_linear.bias"})
    with pytest.raises(NotImplementedError):
        S.NeRF(D=4, W=128, input_ch=63)


def test_no_cpu_fallback(S):
    """the product path refuses host tensors instead of computing on the CPU"""
    m = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)
    with pytest.raises(S.HipLibraryError):
        m.query(torch.zeros(2, 4, 3), torch.zeros(2, 3))
    with pytest.raises(S.HipLibraryError):
        S.sample_coarse(torch.zeros(4, 11), 8)


def test_missing_library_fails_loudly(S, monkeypatch):
    monkeypatch.setattr(S._lib, "_lib", None)
    monkeypatch.setattr(S._lib, "LIB_PATH", "/nonexistent/libspinnerf_hip.so")
    with pytest.raises(S.HipLibraryError, match="missing"):
        S._lib.load()


def test_embedder_and_create_nerf_contract(S, tmp_path):
    import argparse
    e, d = S.get_embedder(10, 0)
    assert d == 63 and S.get_embedder(4, 0)[1] == 27 and S.get_embedder(10, -1)[1] == 3
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=128,
                              N_samples=64, alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8,
                              netwidth_fine=256, netchunk=65536, lrate=5e-4, basedir=str(tmp_path), expname="",
                              ft_path=None, no_reload=True, perturb=1.0, white_bkgd=True, raw_noise_std=1.0,
                              dataset_type="llff", no_ndc=True, lindisp=True, sigma_loss=False, no_coarse=False)
    kw_train, kw_test, start, grad_vars, opt = S.create_nerf(args, device=torch.device("cpu"))
    # run_nerf.py:465-492
    assert set(kw_train) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                             "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"}
    assert kw_train["ndc"] is False and kw_test["perturb"] is False and kw_test["raw_noise_std"] == 0.
    assert start == 0 and len(grad_vars) == 2 and isinstance(opt, torch.optim.Adam)
    # checkpoint interchange (run_nerf.py:1626-1636): a per-layer state dict loads into the flat module
    ck = {"global_step": 7, "network_fn_state_dict": kw_train["network_fn"].state_dict(),
          "network_fine_state_dict": kw_train["network_fine"].state_dict(), "optimizer_state_dict": opt.state_dict()}
    torch.save(ck, tmp_path / "000007.tar")
    args.no_reload = False
    _, _, start2, _, _ = S.create_nerf(args, device=torch.device("cpu"))
    assert start2 == 7


def test_create_nerf_with_alpha_model_path(S, tmp_path):
    """--alpha_model_path (run_nerf.py:395-425): a frozen density network loaded from a checkpoint's
    network_fine_state_dict, NeRF_RGB colour networks for coarse and fine that share it."""
    import argparse
    torch.manual_seed(0)
    donor = S.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True)
    torch.save({"network_fine_state_dict": donor.state_dict()}, tmp_path / "alpha.tar")
    (tmp_path / "run").mkdir()
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=128,
                              N_samples=64, alpha_model_path=str(tmp_path / "alpha.tar"), netdepth=8, netwidth=256,
                              netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, basedir=str(tmp_path),
                              expname="run", ft_path=None, no_reload=True, perturb=1.0, white_bkgd=True,
                              raw_noise_std=1.0, dataset_type="llff", no_ndc=True, lindisp=True, sigma_loss=False,
                              no_coarse=False)
    kw_train, _, _, grad_vars, _ = S.create_nerf(args, device=torch.device("cpu"))
    c, f = kw_train["network_fn"], kw_train["network_fine"]
    assert isinstance(c, S.NeRF_RGB) and isinstance(f, S.NeRF_RGB) and c.alpha_model is f.alpha_model
    assert torch.equal(c.alpha_model.flat.detach(), donor.flat.detach())
    # like the reference, the (shared) density network's parameters ride along in both modules' parameter lists
    assert len(grad_vars) == 4 and "alpha_linear.weight" not in c.state_dict() and "alpha_model.alpha_linear.weight" in c.state_dict()
    args.no_coarse = True
    kw_train, _, _, grad_vars, _ = S.create_nerf(args, device=torch.device("cpu"))
    assert kw_train["network_fn"] is None and len(grad_vars) == 2


def test_hashgrid_sizes_and_module_contract_on_host(S):
    """config 5 (NeRF_TCNN): the C ABI's level table agrees with the oracle's restatement of the published definition,
    and the module exposes tiny-cuda-nn's state-dict keys (run_nerf_helpers_tcnn.py:36-84) — no GPU needed."""
    from oracle import hashgrid_oracle as H
    lib = S._lib.load()
    levels, total = H.level_table()
    assert lib.snr_hashgrid_table_entries() == total == 7034832
    assert lib.snr_hashgrid_param_count() == 2 * total + 3072 + 7168
    assert lib.snr_hashgrid_packed_bytes() == 44 * 1024
    assert lib.snr_hashgrid_act_bytes(1000) == 32 * 2048 and lib.snr_hashgrid_act_bytes(0) == -2
    assert lib.snr_hashgrid_bwd_ws_bytes(1024 * 192) > 1024 * 192 * 900
    assert lib.snr_hashgrid_forward(None, None, None, None, 0, None, None, 0, 8, 1, None, None, None) == -1
    assert [l[1] for l in levels[:4]] == [16, 31, 57, 107] and [bool(l[4]) for l in levels[:4]] == [False, False, False, True]
    torch.manual_seed(0)
    net = S.NeRF_TCNN()
    sd = net.state_dict()
    assert list(sd) == ["encoder.params", "sigma_net.params", "encoder_dir.params", "color_net.params"]
    assert sd["encoder.params"].numel() == 2 * total and float(sd["encoder.params"].abs().max()) <= 1e-4
    net2 = S.NeRF_TCNN()
    net2.load_state_dict(sd)
    assert torch.equal(net2.flat.detach(), net.flat.detach())
    with pytest.raises(NotImplementedError):
        S.NeRF_TCNN(hidden_dim=128)
    with pytest.raises(S.HipLibraryError):            # no CPU fallback on this path either
        net(torch.zeros(4, 6))


def test_fused_render_rays_layout_and_argument_checks_on_host(S):
    """snr_render_rays_fused_layout is pure host arithmetic: sections in order, 256-byte aligned, training adds the saved
    activations and the backward workspace; bad arguments are refused before anything touches a GPU."""
    lib = S._lib.load()
    L = S._lib
    cfg = L.MlpConfig(10, 4, 0, 1, 4, L.PREC_BF16)
    blob = ctypes.c_void_p(1)    # never dereferenced by the layout query
    net = L.Net(L.NET_MLP, cfg, blob, blob)
    hg = L.Net(L.NET_HASHGRID, L.MlpConfig(), blob, blob)
    rc = L.RenderConfig(64, 128, 1, 1, 1, 1.0)
    n = 1024
    out_i, out_t = L.RenderWsLayout(), L.RenderWsLayout()
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(net), ctypes.byref(net), n, 0, ctypes.byref(out_i)) == 0
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(net), ctypes.byref(net), n, 1, ctypes.byref(out_t)) == 0
    assert out_i.z_coarse == 0 and out_i.raw0 == n * 64 * 4 and out_i.weights0 == out_i.raw0 + n * 64 * 16
    assert out_i.z_vals > out_i.depth0 and out_i.raw == out_i.z_vals + n * 192 * 4 and out_i.d_raw == -1 and out_i.act == -1
    for k, _ in L.RenderWsLayout._fields_:
        v = getattr(out_t, k)
        assert v == -1 or v % 256 == 0, k
    act0, act = lib.snr_mlp_act_bytes(ctypes.byref(cfg), n * 64), lib.snr_mlp_act_bytes(ctypes.byref(cfg), n * 192)
    bw = lib.snr_mlp_bwd_ws_bytes(ctypes.byref(cfg), n * 192)
    assert out_t.total >= out_i.total + act0 + act + bw and out_t.bwd_ws + bw <= out_t.total
    # coarse only: no fine-pass sections; hash-grid networks: their own sizes
    rc0 = L.RenderConfig(64, 0, 0, 0, 0, 0.0)
    o = L.RenderWsLayout()
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc0), ctypes.byref(net), None, n, 1, ctypes.byref(o)) == 0
    assert o.z_vals == -1 and o.raw == -1 and o.act == -1 and o.act0 > 0
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(hg), ctypes.byref(hg), n, 1, ctypes.byref(o)) == 0
    assert o.total > lib.snr_hashgrid_bwd_ws_bytes(n * 192)
    # refusals
    bad = L.Net(7, cfg, blob, blob)
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(bad), None, n, 0, ctypes.byref(o)) == -3
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(net), None, 0, 0, ctypes.byref(o)) == -2
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), None, None, n, 0, ctypes.byref(o)) == -1
    assert lib.snr_render_rays_fused_layout(ctypes.byref(L.RenderConfig(1, 0, 0, 0, 0, 0.0)), ctypes.byref(net), None, n, 0,
                                            ctypes.byref(o)) == -2
    nopack = L.Net(L.NET_MLP, cfg, None, blob)
    assert lib.snr_render_rays_fused_layout(ctypes.byref(rc), ctypes.byref(nopack), None, n, 0, ctypes.byref(o)) == -1
    assert lib.snr_render_rays_fused_forward(ctypes.byref(rc), ctypes.byref(net), None, None, 11, n, None, None, None, None, 0, 0,
                                             None, None, n, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.snr_adam_step_dev(None, None, None, None, 8, None, 0.9, 0.999, 1e-8, 1.0, None) == -1
    assert lib.snr_step_state_advance(None, 5e-4, 250000.0, 0.9, 0.999, 4, None) == -1
    assert ctypes.sizeof(S._lib.StepState) == 40
    assert lib.snr_render_rays_fused_backward(ctypes.byref(rc), ctypes.byref(net), None, blob, 11, n, blob, blob, None, 0, 0,
                                              None) == -2       # no pass selected


def test_process_wide_settings_are_read_once_and_thread_safe(S, monkeypatch):
    """VERDICT r03 item 7.  The library's only process-wide state besides the profiling log: the SNR_* environment switches
    and each device's CU count, read ONCE (include/spinnerf_hip.h, conventions) — no launch counter, no per-call getenv.
    (a) eight host threads asking for the backward workspace size at once (the first call initialises the settings) all get
    the same answer; (b) changing the environment afterwards changes nothing until snr_tunables_reload(); (c) the new entry
    points validate their arguments on the host."""
    import ctypes
    import threading
    lib = S._lib.load()
    cfg = S._lib.MlpConfig(10, 4, 0, 1, 4, S._lib.PREC_BF16)
    n = 1024 * 192
    monkeypatch.delenv("SNR_PAIR_SLOTS", raising=False)
    lib.snr_tunables_reload()
    out = [None] * 8

    def ask(i):
        out[i] = [lib.snr_mlp_bwd_ws_bytes(cfg, n) for _ in range(200)]
    ts = [threading.Thread(target=ask, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    base = out[0][0]
    assert base > 0 and all(v == base for o in out for v in o)
    # the workspace bound covers every launch shape the switches can select (round 5, ADVICE r04: a size query must stay
    # valid across snr_tunables_reload and across devices): SNR_PAIR_SLOTS / SNR_PLAIN_WGS do not move it ...
    monkeypatch.setenv("SNR_PAIR_SLOTS", "64")
    monkeypatch.setenv("SNR_PLAIN_WGS", "200")
    lib.snr_tunables_reload()
    assert lib.snr_mlp_bwd_ws_bytes(cfg, n) == base
    monkeypatch.delenv("SNR_PAIR_SLOTS")
    monkeypatch.delenv("SNR_PLAIN_WGS")
    # ... while a switch that changes what the backward stores does, but not before the switches are read again
    monkeypatch.setenv("SNR_RECOMPUTE", "0")            # every layer's d z saved, plain split-K pass
    assert lib.snr_mlp_bwd_ws_bytes(cfg, n) == base
    lib.snr_tunables_reload()
    other = lib.snr_mlp_bwd_ws_bytes(cfg, n)
    assert other > 0 and other != base
    monkeypatch.delenv("SNR_RECOMPUTE")
    lib.snr_tunables_reload()
    assert lib.snr_mlp_bwd_ws_bytes(cfg, n) == base
    # new entry points: argument checks before any device access
    assert lib.snr_mlp_backward_multi(None, 1, None) == -1
    items = (S._lib.MlpBwdItem * 1)()
    assert lib.snr_mlp_backward_multi(items, 3, None) == -2 and lib.snr_mlp_backward_multi(items, 1, None) == -1
    assert lib.snr_adam_pack_multi(None, 1, 5e-4, 0.9, 0.999, 1e-8, 1, 1.0, None, None) == -1
    ap = (S._lib.AdamPackItem * 1)()
    assert lib.snr_adam_pack_multi(ap, 1, 5e-4, 0.9, 0.999, 1e-8, 0, 1.0, None, None) == -2   # step is 1-based
    rc = S._lib.RenderConfig(64, 128, 0, 0, 1, 1.0, 0)
    assert lib.snr_render_rays_fused_forward_terms(ctypes.byref(rc), None, None, None, 11, 8, None, None, None, None, 0, 0, None, None,
                                                   None, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.snr_render_step_prepare(ctypes.byref(rc), None, None, 8, 4, 4, 1.0, 0, 0.0, 1.0, 1, None, 11, None, 0, 0, None,
                                       None, None, None) == -1


def test_poison_switch_wraps_the_allocation_entry_points_and_is_off_by_default():
    """spin-nerf_amd/_debug.py (round 6): SNR_POISON_WS=1 fills every DEVICE buffer obtained through torch.empty / empty_like /
    new_empty with 0xFF bytes before the library sees it.  Without the variable nothing is wrapped; the wrappers leave host
    tensors alone (there is no GPU here: the fill itself is exercised by every soak pass on the GPU box)."""
    import importlib
    import subprocess
    import sys
    import torch
    dbg = importlib.import_module("spin-nerf_amd._debug")
    assert not dbg.poison_enabled() or os.environ.get("SNR_POISON_WS") not in (None, "", "0")
    code = ("import os, sys, importlib, torch; sys.path.insert(0, %r); e0 = torch.empty; "
            "d = importlib.import_module('spin-nerf_amd._debug'); "
            "assert (torch.empty is not e0) == (os.environ.get('SNR_POISON_WS') == '1'); "
            "t = torch.empty(4); u = torch.empty_like(t); v = t.new_empty(3); assert t.shape == (4,) and u.shape == (4,) and v.shape == (3,); "
            "assert d.install_poison() and torch.empty is not e0; print('ok')") % ROOT
    for val in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SNR_POISON_WS=val), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (val, r.stderr[-1500:])
