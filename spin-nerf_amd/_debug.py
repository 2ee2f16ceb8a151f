"""Debug switches of the Python side (nothing here is on the product path unless its environment variable is set).

SNR_POISON_WS=1 — every device buffer this process obtains through ``torch.empty`` / ``torch.empty_like`` /
``Tensor.new_empty`` is filled with 0xFF bytes before anybody sees it: fp32 and bf16 NaNs, all-ones flag words, -1 indices.
The library never allocates (include/spinnerf_hip.h: the caller provides outputs and workspaces), so this covers every
output tensor, saved-activation buffer, backward workspace and fused-render workspace its kernels are handed.  A kernel that
reads a byte no kernel of the same call wrote then computes NaNs on EVERY run instead of on the runs where the caching
allocator happens to hand back a block with unlucky contents (VERDICT r05 "next round" item 1: the one unexplained failure of
the GPU suite only ever appeared inside a full-suite run).  The fill is an ordinary kernel on torch's current stream, i.e.
ordered in front of the library launches that follow on it.
"""
import os

import torch

_installed = False


def poison_enabled():
    return os.environ.get("SNR_POISON_WS", "0") not in ("", "0")


def _poison(t):
    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and t.is_contiguous() and not t.dtype.is_complex:
        try:
            t.view(torch.uint8).fill_(0xFF)
        except RuntimeError:          # (dtypes a byte view cannot express: bool and friends)
            t.fill_(True) if t.dtype == torch.bool else None
    return t


def install_poison():
    """Wrap the allocation entry points (idempotent).  Returns True when the wrappers are in place."""
    global _installed
    if _installed:
        return True
    _empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def empty(*a, **k):
        return _poison(_empty(*a, **k))

    def empty_like(*a, **k):
        return _poison(_empty_like(*a, **k))

    def new_empty(self, *a, **k):
        return _poison(_new_empty(self, *a, **k))
    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty
    _installed = True
    return True


if poison_enabled():
    install_poison()
