"""spin-nerf_amd — MI355X-native volumetric renderer, drop-in for the render()/render_rays()/
network_query_fn surface of SamsungLabs/SPIn-NeRF's DS_NeRF/run_nerf.py.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it either as

    import spin_nerf_amd                      # alias module at the repo root
    importlib.import_module("spin-nerf_amd")  # the package itself
"""
from . import _debug          # SNR_POISON_WS=1: poison-filled torch.empty (debug)
from . import _lib
from ._lib import HipLibraryError, LIB_PATH
from .nerf import NeRF, NeRF_RGB
from .ops import raw2outputs, raw2outputs_mvseg, sample_pdf, sample_coarse, sample_fine, make_rays, mlp_query, adam_step_
from .render import (render, render_rays, batchify_rays, batchify, run_network, create_nerf, get_embedder, get_rays,
                     ndc_rays, Embedder)
from .hashgrid import NeRF_TCNN, create_nerf_tcnn
from .loss import SigmaLoss
from .poses import get_rays_np, get_rays_by_coord_np
from .path import (render_path, render_sharded, render_path_projection, render_test_ray, sample_sigma, convert_pose,
                   to8b, write_png)

img2mse = lambda x, y: ((x - y) ** 2).mean()                      # helpers:15
mse2psnr = lambda x: -10. * x.log() / 2.302585092994046           # helpers:17

__all__ = ["NeRF", "NeRF_RGB", "NeRF_TCNN", "create_nerf_tcnn", "render", "render_rays", "batchify_rays", "batchify", "run_network", "create_nerf",
           "get_embedder", "get_rays", "ndc_rays", "raw2outputs", "raw2outputs_mvseg", "sample_pdf", "sample_coarse", "sample_fine", "make_rays",
           "mlp_query", "adam_step_", "img2mse", "mse2psnr", "HipLibraryError", "LIB_PATH", "Embedder", "SigmaLoss", "get_rays_np", "get_rays_by_coord_np", "render_path", "render_sharded", "render_path_projection", "render_test_ray", "sample_sigma", "convert_pose", "to8b", "write_png"]
