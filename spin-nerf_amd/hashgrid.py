"""Hash-grid radiance network backed by the HIP kernels of csrc/hashgrid.hip — the reference's default network
(``NeRF_TCNN``, DS_NeRF/run_nerf_helpers_tcnn.py:13-113; ``create_nerf_tcnn``, DS_NeRF/run_nerf.py:499-590; BASELINE
config 5).

The reference builds it from tiny-cuda-nn modules (HashGrid / SphericalHarmonics encodings, two FullyFusedMLPs); that
library is not part of the reference tree, so parity is UNPINNED: the kernels follow the published definition restated
in oracle/hashgrid_oracle.py and are tested against that restatement.  Same constructor defaults, same state-dict keys
(``encoder.params``, ``sigma_net.params``, ``encoder_dir.params`` (empty), ``color_net.params``), same forward
signature (``input`` [N, 6] = position, direction -> [N, 4] = colour without activation, sigma).

All parameters live in one flat fp32 buffer [table | sigma_net | color_net] so that Adam, the gradient all-reduce and
checkpoints treat it like the other networks of this package.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, f32c, ptr, stream

_SIGMA, _COLOR = 2048 + 1024, 2048 + 4096 + 1024


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, net, pts, rays, z_vals, viewdirs, n_samples, S):
        lib = _lib.load()
        packed = net.packed_weights()
        raw = torch.empty(n_samples, 4, device=flat.device, dtype=torch.float32)
        act = None
        if ctx.needs_input_grad[0]:
            act = torch.empty(lib.snr_hashgrid_act_bytes(n_samples), device=flat.device, dtype=torch.uint8)
        check(lib.snr_hashgrid_forward(ptr(flat.detach()), ptr(packed), ptr(pts), ptr(rays),
                                       rays.shape[1] if rays is not None else 0, ptr(z_vals), ptr(viewdirs),
                                       viewdirs.stride(0), n_samples, S, ptr(raw), ptr(act), stream()),
              "snr_hashgrid_forward")
        ctx.net, ctx.n, ctx.S = net, n_samples, S
        ctx.act, ctx.packed, ctx.inputs = act, packed, (pts, rays, z_vals, viewdirs)
        ctx.keys = (net.pack_generation, flat._version, net.weights_generation)
        ctx.set_materialize_grads(False)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        if d_raw is None:
            ctx.act = None
            return (None,) * 8
        lib = _lib.load()
        net = ctx.net
        if ctx.act is None:
            raise RuntimeError("the saved features of this evaluation were released by its first backward")
        if (net.pack_generation, net.flat._version, net.weights_generation) != ctx.keys:
            raise RuntimeError("the network's parameters changed between forward and backward: re-run the forward")
        pts, rays, z_vals, viewdirs = ctx.inputs
        g = torch.empty_like(net.flat)
        ws = torch.empty(lib.snr_hashgrid_bwd_ws_bytes(ctx.n), device=g.device, dtype=torch.uint8)
        check(lib.snr_hashgrid_backward(ptr(net.flat.detach()), ptr(ctx.packed), ptr(pts), ptr(rays),
                                        rays.shape[1] if rays is not None else 0, ptr(z_vals), ptr(viewdirs),
                                        viewdirs.stride(0), ptr(f32c(d_raw)), ctx.n, ctx.S, ptr(ctx.act), ptr(ws), ptr(g), 0,
                                        stream()), "snr_hashgrid_backward")
        ctx.act = None
        return (g,) + (None,) * 7


class NeRF_TCNN(nn.Module):
    use_viewdirs = True      # forward reads input[:, 3:] (run_nerf_helpers_tcnn.py:88)
    _UNREGISTERED = ()

    def __init__(self, encoding="HashGrid", encoding_dir="SphericalHarmonics", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=100, **kwargs):
        super().__init__()
        if (num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color, bound) != (2, 64, 15, 3, 64, 100):
            raise NotImplementedError("the HIP hash-grid kernels implement the reference's constants: num_layers=2, "
                                      "hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=100")
        self.bound, self.num_layers, self.hidden_dim, self.geo_feat_dim = bound, num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color, self.in_dim_color = num_layers_color, hidden_dim_color, 16 + geo_feat_dim
        try:
            entries = int(_lib.load().snr_hashgrid_table_entries())
        except _lib.HipLibraryError:
            entries = 7034832          # 4096 + 29792 + 185200 + 13 * 2^19 (oracle/hashgrid_oracle.py: level_table)
        self.table_entries = entries

        def xavier(fout, fin):         # tiny-cuda-nn's FullyFusedMLP initialisation
            s = math.sqrt(6.0 / (fin + fout))
            return (torch.rand(fout * fin) * 2 - 1) * s
        grid = (torch.rand(entries * 2) * 2 - 1) * 1e-4
        nets = [xavier(64, 32), xavier(16, 64), xavier(64, 32), xavier(64, 64), xavier(16, 64)]
        self.flat = nn.Parameter(torch.cat([grid] + nets))
        self._packed = None
        self._packed_key = None
        self.pack_generation = 0
        self.weights_generation = 0

    # ---- flat <-> named views (tiny-cuda-nn modules expose one `params` vector each) -----------------------------------
    def named_views(self, flat=None):
        flat = self.flat if flat is None else flat
        g = self.table_entries * 2
        return OrderedDict([("encoder.params", flat[:g]), ("sigma_net.params", flat[g:g + _SIGMA]),
                            ("encoder_dir.params", flat[g:g]), ("color_net.params", flat[g + _SIGMA:g + _SIGMA + _COLOR])])

    param_views = named_views

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_views(self.flat if keep_vars else self.flat.detach()).items():
            destination[prefix + k] = v if keep_vars else v.clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        views = self.named_views(self.flat.detach())
        for k, v in views.items():
            key = prefix + k
            if key not in state_dict:
                if strict and v.numel():
                    missing_keys.append(key)
                continue
            if state_dict[key].numel() != v.numel():
                error_msgs.append(f"size mismatch for {key}: {tuple(state_dict[key].shape)} vs {tuple(v.shape)}")
                continue
            with torch.no_grad():
                v.copy_(state_dict[key].reshape(v.shape).to(v.dtype))
        if strict:
            for key in state_dict:
                if key.startswith(prefix) and key[len(prefix):] not in views:
                    unexpected_keys.append(key)
        self.mark_weights_changed()

    # ---- packed MLP weights --------------------------------------------------------------------------------------------
    def packed_weights(self):
        lib = _lib.load()
        key = (self.flat.data_ptr(), self.flat._version, self.weights_generation)
        if self._packed is None or self._packed_key != key or self._packed.device != self.flat.device:
            if self._packed is None or self._packed.device != self.flat.device:
                self._packed = torch.empty(lib.snr_hashgrid_packed_bytes(), device=self.flat.device, dtype=torch.uint8)
            check(lib.snr_hashgrid_pack(ptr(self.flat.detach()), ptr(self._packed), stream()), "snr_hashgrid_pack")
            self._packed_key = key
            self.pack_generation += 1
        return self._packed

    def mark_weights_changed(self):
        self._packed_key = None
        self.weights_generation += 1

    # ---- evaluation ----------------------------------------------------------------------------------------------------
    def _eval(self, pts, rays, z_vals, viewdirs, M, S):
        if viewdirs is None:
            raise ValueError("NeRF_TCNN needs view directions (run_nerf_helpers_tcnn.py:88)")
        viewdirs = viewdirs.detach()
        if viewdirs.dtype != torch.float32 or viewdirs.stride(-1) != 1:
            viewdirs = f32c(viewdirs)
        flat = self.flat if torch.is_grad_enabled() else self.flat.detach()   # no_grad: inference kernel, nothing saved
        return _HashGrid.apply(flat, self, pts, rays, z_vals, viewdirs, M, S)

    # ---- the autograd-free training step (train.py: _step_direct; same contract as ops.mlp_train_forward / _backward) --
    def train_forward(self, rays, z_vals, viewdirs):
        lib = _lib.load()
        if viewdirs is None:
            raise ValueError("NeRF_TCNN needs view directions (run_nerf_helpers_tcnn.py:88)")
        packed = self.packed_weights()
        n, S = z_vals.numel(), z_vals.shape[1]
        raw = torch.empty(z_vals.shape[0], S, 4, device=z_vals.device, dtype=torch.float32)
        act = torch.empty(lib.snr_hashgrid_act_bytes(n), device=z_vals.device, dtype=torch.uint8)
        check(lib.snr_hashgrid_forward(ptr(self.flat.detach()), ptr(packed), None, ptr(rays), rays.shape[1], ptr(z_vals),
                                       ptr(viewdirs), viewdirs.stride(0), n, S, ptr(raw), ptr(act), stream()),
              "snr_hashgrid_forward")
        return raw, (packed, act, n, S, rays, z_vals, viewdirs)

    def train_backward(self, saved, d_raw):
        lib = _lib.load()
        packed, act, n, S, rays, z_vals, viewdirs = saved
        g = torch.empty_like(self.flat.data)
        ws = torch.empty(lib.snr_hashgrid_bwd_ws_bytes(n), device=g.device, dtype=torch.uint8)
        check(lib.snr_hashgrid_backward(ptr(self.flat.detach()), ptr(packed), None, ptr(rays), rays.shape[1], ptr(z_vals),
                                        ptr(viewdirs), viewdirs.stride(0), ptr(d_raw), n, S, ptr(act), ptr(ws), ptr(g), 0,
                                        stream()), "snr_hashgrid_backward")
        return g

    def query(self, inputs, viewdirs=None):
        S = inputs.shape[-2] if inputs.dim() > 1 else 1
        pts = f32c(inputs.detach().reshape(-1, 3))
        raw = self._eval(pts, None, None, viewdirs.reshape(-1, viewdirs.shape[-1]) if viewdirs is not None else None,
                         pts.shape[0], S)
        return raw.reshape(list(inputs.shape[:-1]) + [4])

    def query_rays(self, ray_batch, z_vals, viewdirs=None):
        rays, z = f32c(ray_batch.detach()), f32c(z_vals.detach())
        raw = self._eval(None, rays, z, viewdirs, z.numel(), z.shape[1])
        return raw.reshape(z.shape[0], z.shape[1], 4)

    def forward(self, input):
        """input [N, 6] = (x in [-bound, bound], unit direction) -> [N, 4] (run_nerf_helpers_tcnn.py:86-113)"""
        x = input.reshape(-1, input.shape[-1])
        raw = self._eval(f32c(x[:, :3].detach()), None, None, x[:, 3:6], x.shape[0], 1)
        return raw.reshape(list(input.shape[:-1]) + [4])


def create_nerf_tcnn(args, device=None):
    """create_nerf_tcnn (run_nerf.py:499-590): identity embedders, NeRF_TCNN coarse (+ fine) network, Adam, the render
    kwargs.  Like the reference, checkpoints are never reloaded on this path (`ckpts = []`, run_nerf.py:548)."""
    from .render import run_network
    device = device or torch.device("cuda")
    embed_fn = lambda inp: inp
    embeddirs_fn = (lambda inp: inp) if args.use_viewdirs else None
    model = model_fine = None
    grad_vars = []
    if getattr(args, "alpha_model_path", None) is None:
        model = NeRF_TCNN(encoding="hashgrid").to(device)
        grad_vars = list(model.parameters())
    if args.N_importance > 0:
        if getattr(args, "alpha_model_path", None) is None:
            model_fine = NeRF_TCNN(encoding="hashgrid").to(device)
        grad_vars += list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)
    network_query_fn._snr_fused = True

    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    print('Found ckpts', [])
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model, 'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    else:
        render_kwargs_train['ndc'] = True
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
