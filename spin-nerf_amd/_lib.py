"""ctypes binding of libspinnerf_hip.so (the C ABI declared in include/spinnerf_hip.h).

There is no CPU fallback: every wrapper raises if the library is missing or a call fails.
Tensors are passed as raw device pointers; the launch goes on torch's current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNR_LIB overrides the library path (kernel experiments: ablation builds under lib/ablate/)
LIB_PATH = os.environ.get("SNR_LIB") or os.path.join(_HERE, "lib", "libspinnerf_hip.so")

PREC_BF16, PREC_FP32 = 0, 1

_c = ctypes
_p = _c.c_void_p
_i = _c.c_int
_l = _c.c_int64
_f = _c.c_float


class MlpConfig(_c.Structure):
    _fields_ = [("multires", _i), ("multires_views", _i), ("i_embed", _i), ("use_viewdirs", _i),
                ("out_ch", _i), ("precision", _i)]

    def key(self):
        return (self.multires, self.multires_views, self.i_embed, self.use_viewdirs, self.out_ch, self.precision)


_CFG = _c.POINTER(MlpConfig)

NET_MLP, NET_HASHGRID = 0, 1


class Net(_c.Structure):            # snr_net
    _fields_ = [("kind", _i), ("mlp", MlpConfig), ("packed", _p), ("params", _p)]


class RenderConfig(_c.Structure):   # snr_render_config
    _fields_ = [("n_samples", _i), ("n_importance", _i), ("lindisp", _i), ("white_bkgd", _i), ("perturb", _i),
                ("raw_noise_std", _f), ("flags", _i)]


class RenderWsLayout(_c.Structure):  # snr_render_ws_layout
    _fields_ = [(k, _l) for k in ("z_coarse", "raw0", "weights0", "depth0", "z_vals", "raw", "weights", "z_samples", "d_raw0",
                                  "d_raw", "act0", "act", "bwd_ws", "bwd_ws0", "total")]


class LossTerm(_c.Structure):       # snr_loss_term
    _fields_ = [("first_ray", _l), ("n_rays", _l), ("kind", _i), ("target", _p), ("count", _l), ("slot", _i), ("slot_final", _i)]


class LossTerms(_c.Structure):      # snr_loss_terms
    _fields_ = [("n_terms", _i), ("term", LossTerm * 4), ("guard_term", _i)]


ERR_UNSUPPORTED = -3                # SNR_ERR_UNSUPPORTED
LOSS_RGB, LOSS_RGB_DETACHED, LOSS_DISP = 0, 1, 2
RENDER_Z_COARSE_READY, RENDER_LOSS4 = 1, 2


class StepState(_c.Structure):      # snr_step_state
    _fields_ = [("offset_base", _c.c_uint64), ("opt_step", _l), ("global_step", _l), ("lr", _f), ("bc1", _f),
                ("bc2_sqrt", _f), ("reserved", _f)]


class MlpBwdItem(_c.Structure):     # snr_mlp_bwd_item
    _fields_ = [("cfg", _CFG), ("packed", _p), ("params", _p), ("d_raw", _p), ("n_samples", _l), ("act", _p), ("ws", _p),
                ("grad_params", _p), ("accumulate", _i)]


class AdamPackItem(_c.Structure):   # snr_adam_pack_item
    _fields_ = [("cfg", _CFG), ("params", _p), ("grads", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("packed", _p)]


_NET, _RCFG = _c.POINTER(Net), _c.POINTER(RenderConfig)

# name -> (restype, argtypes); mirrors include/spinnerf_hip.h one to one
SIGNATURES = {
    "snr_abi_version": (_i, []),
    "snr_status_string": (_c.c_char_p, [_i]),
    "snr_mlp_param_count": (_l, [_CFG]),
    "snr_mlp_packed_bytes": (_l, [_CFG]),
    "snr_mlp_act_bytes": (_l, [_CFG, _l]),
    "snr_mlp_bwd_ws_bytes": (_l, [_CFG, _l]),
    "snr_mlp_pack": (_i, [_CFG, _p, _p, _p]),
    "snr_mlp_forward": (_i, [_CFG, _p, _p, _p, _i, _p, _p, _i, _l, _i, _p, _p, _p]),
    "snr_mlp_backward": (_i, [_CFG, _p, _p, _p, _l, _p, _p, _p, _i, _p]),
    "snr_mlp_backward_multi": (_i, [_c.POINTER(MlpBwdItem), _i, _p]),
    "snr_hashgrid_table_entries": (_l, []),
    "snr_hashgrid_param_count": (_l, []),
    "snr_hashgrid_packed_bytes": (_l, []),
    "snr_hashgrid_act_bytes": (_l, [_l]),
    "snr_hashgrid_bwd_ws_bytes": (_l, [_l]),
    "snr_hashgrid_pack": (_i, [_p, _p, _p]),
    "snr_hashgrid_forward": (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _l, _i, _p, _p, _p]),
    "snr_hashgrid_backward": (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _p, _l, _i, _p, _p, _p, _i, _p]),
    "snr_sample_coarse": (_i, [_p, _i, _l, _i, _i, _p, _p, _p]),
    "snr_composite_forward": (_i, [_p, _i, _p, _p, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "snr_composite_backward": (_i, [_p, _i, _p, _p, _i, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "snr_composite_alpha_forward": (_i, [_p, _i, _p, _p, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p]),
    "snr_composite_alpha_backward": (_i, [_p, _i, _p, _p, _i, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "snr_sample_fine": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _p, _p]),
    "snr_sample_coarse_rng": (_i, [_p, _i, _l, _i, _i, _c.c_uint64, _c.c_uint64, _p, _p]),
    "snr_sample_fine_rng": (_i, [_p, _p, _l, _i, _i, _c.c_uint64, _c.c_uint64, _p, _p, _p, _p]),
    "snr_composite_train": (_i, [_p, _i, _p, _p, _i, _p, _f, _c.c_uint64, _c.c_uint64, _l, _i, _i, _i, _p, _l, _p, _p, _p, _p,
                                 _p, _p, _p, _p, _p]),
    "snr_render_rays_fused_layout": (_i, [_RCFG, _NET, _NET, _l, _i, _c.POINTER(RenderWsLayout)]),
    "snr_adam_step_dev": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _p]),
    "snr_step_state_advance": (_i, [_p, _c.c_double, _c.c_double, _f, _f, _c.c_uint64, _p]),
    "snr_render_rays_fused_forward": (_i, [_RCFG, _NET, _NET, _p, _i, _l, _p, _p, _p, _p, _c.c_uint64, _c.c_uint64, _p, _p, _l,
                                           _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "snr_render_rays_fused_forward_terms": (_i, [_RCFG, _NET, _NET, _p, _i, _l, _p, _p, _p, _p, _c.c_uint64, _c.c_uint64, _p,
                                                 _c.POINTER(LossTerms), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "snr_render_step_prepare": (_i, [_RCFG, _p, _p, _l, _i, _i, _f, _i, _f, _f, _i, _p, _i, _p, _c.c_uint64, _c.c_uint64, _p, _p, _p, _p]),
    "snr_render_rays_fused_backward": (_i, [_RCFG, _NET, _NET, _p, _i, _l, _p, _p, _p, _i, _i, _p]),
    "snr_make_rays": (_i, [_i, _i, _f, _c.POINTER(_f), _i, _i, _i, _i, _i, _f, _f, _i, _p, _i, _p]),
    "snr_sample_pdf": (_i, [_p, _p, _p, _l, _i, _i, _p, _p]),
    "snr_pack_rays": (_i, [_p, _p, _p, _l, _i, _i, _f, _i, _f, _f, _f, _p, _p, _p, _i, _p, _i, _p]),
    "snr_embed": (_i, [_p, _l, _i, _i, _p, _p]),
    "snr_mse_pair": (_i, [_p, _p, _p, _l, _p, _p, _p, _p]),
    "snr_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _i, _f, _p]),
    "snr_adam_pack_multi": (_i, [_c.POINTER(AdamPackItem), _i, _f, _f, _f, _f, _i, _f, _p, _p]),
    "snr_prof_enable": (_i, [_i]),
    "snr_prof_kernel_count": (_i, []),
    "snr_prof_kernel_name": (_c.c_char_p, [_i]),
    "snr_prof_read": (_i, [_c.POINTER(_c.c_double), _c.POINTER(_l)]),
    "snr_tunables_reload": (_i, []),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


ABI_VERSION = 4   # include/spinnerf_hip.h: SNR_ABI_VERSION (bumped whenever a prototype or a shared struct changes)


def load():
    """Load the shared library (once).  Fails loudly: the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). spin-nerf_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.snr_abi_version() != ABI_VERSION:
        raise HipLibraryError("ABI version mismatch")
    _lib = lib
    return lib


def prof_enable(on=True):
    check(load().snr_prof_enable(int(bool(on))), "snr_prof_enable")


def prof_read():
    """{kernel name: (total_ms, launches)} of everything recorded since the last read."""
    lib = load()
    n = lib.snr_prof_kernel_count()
    ms, cnt = (_c.c_double * n)(), (_l * n)()
    check(lib.snr_prof_read(ms, cnt), "snr_prof_read")
    return {lib.snr_prof_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}


def check(status, what):
    if status != 0:
        msg = load().snr_status_string(int(status)).decode()
        raise HipLibraryError(f"{what} failed: {msg} (status {status})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be a contiguous fp32/raw CUDA tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("spin-nerf_amd kernels need device (HIP) tensors; got a CPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32c(t):
    """contiguous fp32 view/copy"""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
