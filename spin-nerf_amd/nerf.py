"""NeRF MLP module backed by the fused HIP kernels.

Same constructor signature and state-dict key names as the reference's ``NeRF``
(DS_NeRF/run_nerf_helpers.py:74-127), so reference checkpoints load unchanged
(``network_fn_state_dict`` / ``network_fine_state_dict``, run_nerf.py:1626-1636).

Memory layout: all parameters live in ONE flat fp32 buffer (``self.flat``, the only registered
nn.Parameter, laid out in the reference module's parameter order) so that the optimizer step, the
data-parallel gradient all-reduce and the weight re-pack each touch a single contiguous 2.4 MB
range.  ``state_dict()`` / ``load_state_dict()`` translate to and from the per-layer keys.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream

_PREC = {"bf16": _lib.PREC_BF16, "fp32": _lib.PREC_FP32}
DEFAULT_PRECISION = "bf16"


class _LinearView:
    """weight/bias views into the flat buffer (what ``model.pts_linears[i]`` exposes)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


def _layer_specs(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs):
    """(key prefix, out_features, in_features) in the reference's registration order (helpers:86-102)."""
    specs = [("pts_linears.0", W, input_ch)]
    for i in range(D - 1):
        specs.append((f"pts_linears.{i + 1}", W, W + input_ch if i in skips else W))
    specs.append(("views_linears.0", W // 2, input_ch_views + W))
    if use_viewdirs:
        specs += [("feature_linear", W, W), ("alpha_linear", 1, W), ("rgb_linear", 3, W // 2)]
    else:
        specs.append(("output_linear", output_ch, W))
    return specs


class NeRF(nn.Module):
    _UNREGISTERED = ()   # layer names present in the flat buffer but not in the reference module (NeRF_RGB)

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 precision=None):
        super().__init__()
        if D != 8 or W != 256 or list(skips) != [4]:
            raise NotImplementedError(
                f"the HIP NeRF kernels implement netdepth=8, netwidth=256, skips=[4] (the reference defaults); "
                f"got D={D}, W={W}, skips={skips}")
        views_ok = (input_ch_views == 0 and not use_viewdirs) or (input_ch_views >= 3 and (input_ch_views - 3) % 6 == 0)
        if input_ch < 3 or (input_ch - 3) % 6 or not views_ok:
            raise NotImplementedError("input_ch / input_ch_views must be 3 + 6*multires (get_embedder, helpers:55-70)")
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.output_ch = list(skips), bool(use_viewdirs), output_ch
        self.precision = precision or DEFAULT_PRECISION
        # multires_views = -1 tells the C layout that views_linears.0 has no direction columns
        self.cfg = _lib.MlpConfig(
            multires=(input_ch - 3) // 6, multires_views=(input_ch_views - 3) // 6 if input_ch_views else -1,
            i_embed=0, use_viewdirs=int(self.use_viewdirs), out_ch=4 if use_viewdirs else output_ch,
            precision=_PREC[self.precision])
        self._specs = _layer_specs(D, W, input_ch, input_ch_views, output_ch, self.skips, self.use_viewdirs)
        # nn.Linear default init, drawn in the reference's construction order so that the same
        # torch seed yields the same network as the reference (helpers:86-102)
        chunks = []
        for name, fout, fin in self._specs:
            if name in self._UNREGISTERED:   # a block the kernels need but the reference module never constructs:
                chunks += [torch.zeros(fout * fin), torch.zeros(fout)]   # zeros, and no draw from the RNG stream
                continue
            lin = nn.Linear(fin, fout)
            chunks += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        self.flat = nn.Parameter(torch.cat(chunks))
        n_expected = None
        try:
            n_expected = _lib.load().snr_mlp_param_count(self.cfg)
        except _lib.HipLibraryError:
            pass  # library presence is enforced at first use
        if n_expected is not None and n_expected != self.flat.numel():
            raise RuntimeError(f"flat layout mismatch: {self.flat.numel()} vs C ABI {n_expected}")
        self._packed = None
        self._packed_key = None
        self.pack_generation = 0   # bumped on every (in-place) re-pack; autograd contexts check it (ops._Mlp)
        self.weights_generation = 0   # bumped by mark_weights_changed() (writes the version counter does not see)

    # ---- flat <-> named views --------------------------------------------------------------
    def named_views(self, flat=None):
        flat = self.flat if flat is None else flat
        out, o = OrderedDict(), 0
        for name, fout, fin in self._specs:
            out[name + ".weight"] = flat[o:o + fout * fin].view(fout, fin); o += fout * fin
            out[name + ".bias"] = flat[o:o + fout]; o += fout
        return out

    def param_views(self, flat=None):
        """views of the parameters the REFERENCE module registers, in its registration order (the layout of
        `grad_vars` and of the per-layer Adam state in checkpoints, run_nerf.py:398-434)"""
        return OrderedDict((k, v) for k, v in self.named_views(flat).items()
                           if k.rsplit(".", 1)[0] not in self._UNREGISTERED)

    def _linear(self, name):
        v = self.named_views(self.flat.detach())
        return _LinearView(v[name + ".weight"], v[name + ".bias"])

    @property
    def pts_linears(self):
        return [self._linear(f"pts_linears.{i}") for i in range(self.D)]

    @property
    def views_linears(self):
        return [self._linear("views_linears.0")]

    def __getattr__(self, name):
        if name in ("feature_linear", "alpha_linear", "rgb_linear", "output_linear") and "_specs" in self.__dict__:
            if any(s[0] == name for s in self._specs):
                return self._linear(name)
        return super().__getattr__(name)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_views(self.flat if keep_vars else self.flat.detach()).items():
            destination[prefix + k] = v if keep_vars else v.clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        views = self.named_views(self.flat.detach())
        for k, v in views.items():
            key = prefix + k
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(v.shape):
                error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} "
                                  f"from checkpoint, the shape in current model is {tuple(v.shape)}.")
                continue
            with torch.no_grad():
                v.copy_(src)
        if strict:
            for key in state_dict:
                if key.startswith(prefix) and key[len(prefix):] not in views:
                    unexpected_keys.append(key)

    # ---- packed weights ------------------------------------------------------------------------
    def packed_weights(self):
        """MFMA-fragment-order copy of the weights; rebuilt whenever the flat buffer changed
        (tensor version counter) — i.e. once per optimizer step."""
        lib = _lib.load()
        key = (self.flat.data_ptr(), self.flat._version, self.cfg.key())
        if self._packed is None or self._packed_key != key or self._packed.device != self.flat.device:
            nbytes = lib.snr_mlp_packed_bytes(self.cfg)
            if nbytes <= 0:
                check(int(nbytes), "snr_mlp_packed_bytes")
            if self._packed is None or self._packed.numel() != nbytes or self._packed.device != self.flat.device:
                self._packed = torch.empty(nbytes, device=self.flat.device, dtype=torch.uint8)
            check(lib.snr_mlp_pack(self.cfg, ptr(self.flat.detach()), ptr(self._packed), stream()), "snr_mlp_pack")
            self._packed_key = key
            self.pack_generation += 1
        return self._packed

    def mark_weights_changed(self):
        """call after writing ``flat`` through a raw pointer (in-place torch ops are detected by version)"""
        self._packed_key = None
        self.weights_generation += 1

    def note_packed_in_place(self):
        """the parameters AND the packed blob were rewritten together through raw pointers (ops.adam_pack_step_): the blob
        stays current, but autograd contexts that captured the old weights must notice"""
        self.pack_generation += 1
        self.weights_generation += 1

    def set_precision(self, precision):
        self.precision = precision
        self.cfg.precision = _PREC[precision]
        self._packed_key = None
        return self

    def load_weights_from_keras(self, weights):
        """Weights of the original (TensorFlow / Keras) NeRF release: a list of numpy arrays [kernel, bias] per layer, kernels
        as [in, out] — pts_linears 0..D-1, feature_linear, views_linears.0, rgb_linear, alpha_linear
        (DS_NeRF/run_nerf_helpers.py:129-156; the reference asserts use_viewdirs as well)."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        import numpy as np
        D = self.D
        order = [f"pts_linears.{i}" for i in range(D)] + ["feature_linear", "views_linears.0", "rgb_linear", "alpha_linear"]
        views = self.named_views(self.flat.detach())
        with torch.no_grad():
            for j, name in enumerate(order):
                w = torch.from_numpy(np.ascontiguousarray(np.transpose(weights[2 * j]))).to(self.flat)
                b = torch.from_numpy(np.ascontiguousarray(np.transpose(weights[2 * j + 1]))).to(self.flat)
                if tuple(w.shape) != tuple(views[name + ".weight"].shape) or b.numel() != views[name + ".bias"].numel():
                    raise ValueError(f"{name}: keras kernel {tuple(weights[2 * j].shape)} does not fit {tuple(views[name + '.weight'].shape)}")
                views[name + ".weight"].copy_(w)
                views[name + ".bias"].copy_(b.reshape(-1))
        self.mark_weights_changed()

    # ---- evaluation ----------------------------------------------------------------------------
    def query(self, inputs, viewdirs=None):
        """inputs [..., S, 3] sample positions, viewdirs [..., 3] per ray -> raw [..., S, out_ch]."""
        from .ops import mlp_query
        S = inputs.shape[-2] if inputs.dim() > 1 else 1
        raw = mlp_query(self, pts=inputs.reshape(-1, 3),
                        viewdirs=viewdirs.reshape(-1, viewdirs.shape[-1]) if viewdirs is not None else None,
                        samples_per_ray=S)
        return raw.reshape(list(inputs.shape[:-1]) + [raw.shape[-1]])

    def query_rays(self, ray_batch, z_vals, viewdirs=None):
        """pts = o + d z formed in-kernel from packed ray rows (run_nerf.py:670-671)."""
        from .ops import mlp_query
        raw = mlp_query(self, rays=ray_batch, z_vals=z_vals, viewdirs=viewdirs)
        return raw.reshape(z_vals.shape[0], z_vals.shape[1], raw.shape[-1])

    def forward(self, x):
        """Reference calling convention: x = cat(embed(pts), embed(dirs)) (helpers:104-105).  The
        embeddings carry their raw input in the first three columns (include_input=True,
        helpers:31-33); the kernel re-derives the encoding from those, so any x produced by the
        reference's embedders evaluates identically."""
        from .ops import mlp_query
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        dirs = x2[:, self.input_ch:self.input_ch + 3] if self.use_viewdirs else None
        raw = mlp_query(self, pts=x2[:, :3], viewdirs=dirs, samples_per_ray=1)
        return raw.reshape(list(lead) + [raw.shape[-1]])


class NeRF_RGB(NeRF):
    """Colour network whose density comes from a frozen, separately trained network (helpers:159-216): same
    trunk as NeRF, no `alpha_linear`; with view directions the output is cat(rgb, alpha_model(x)[..., 3]) with
    the density evaluated under no_grad (:202-203), without them it is `output_linear(h)` as in NeRF.

    The fused kernels evaluate a full NeRF, so the flat buffer keeps an `alpha_linear` block that stays zero
    (its output is replaced, hence its gradient is zero and Adam never moves it); the state dict follows the
    reference: no `alpha_linear.*`, and the `alpha_model.*` entries of the registered sub-module."""

    _UNREGISTERED = ("alpha_linear",)   # helpers:183-185: commented out in the reference's NeRF_RGB

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 alpha_model=None, precision=None):
        super().__init__(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=output_ch, skips=skips,
                         use_viewdirs=use_viewdirs, precision=precision)
        self.alpha_model = alpha_model

    def _own_views(self, flat):
        return self.param_views(flat)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self._own_views(self.flat if keep_vars else self.flat.detach()).items():
            destination[prefix + k] = v if keep_vars else v.clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        views = self._own_views(self.flat.detach())
        for k, v in views.items():
            key = prefix + k
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            if tuple(state_dict[key].shape) != tuple(v.shape):
                error_msgs.append(f"size mismatch for {key}")
                continue
            with torch.no_grad():
                v.copy_(state_dict[key])
        if strict:
            for key in state_dict:
                rest = key[len(prefix):]
                if key.startswith(prefix) and rest not in views and not rest.startswith("alpha_model."):
                    unexpected_keys.append(key)
        self._packed_key = None

    def _with_density(self, raw, density_fn):
        if not self.use_viewdirs:
            return raw
        if self.alpha_model is None:
            raise RuntimeError("NeRF_RGB with view directions needs an alpha_model (helpers:202-203)")
        with torch.no_grad():
            alpha = density_fn(self.alpha_model)[..., 3:4]
        return torch.cat([raw[..., :3], alpha], -1)

    def query(self, inputs, viewdirs=None):
        return self._with_density(super().query(inputs, viewdirs), lambda m: m.query(inputs, viewdirs))

    def query_rays(self, ray_batch, z_vals, viewdirs=None):
        return self._with_density(super().query_rays(ray_batch, z_vals, viewdirs),
                                  lambda m: m.query_rays(ray_batch, z_vals, viewdirs))

    def forward(self, x):
        return self._with_density(super().forward(x), lambda m: m(x))
