"""GPU: the chain2 kernels (round 5 experiment, mlp_chain2.h: 8 compute waves + 4 helper waves, SNR_CHAIN2=1) against the
shipped chain kernels.  Same arithmetic in the same order, so the comparison is bit for bit: raw of the inference forward,
parameter gradients through the chain2 dgrad (the weight-gradient pass consumes every d z section it writes)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import spin_nerf_amd as S
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    S._lib.load()
    return S


@pytest.fixture()
def chain2_switch(S):
    lib = S._lib.load()
    old = os.environ.get("SNR_CHAIN2")

    def mode(v):
        os.environ["SNR_CHAIN2"] = str(v)
        lib.snr_tunables_reload()
    yield mode
    if old is None:
        os.environ.pop("SNR_CHAIN2", None)
    else:
        os.environ["SNR_CHAIN2"] = old
    lib.snr_tunables_reload()


def _net(S):
    torch.manual_seed(3)
    net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="bf16").cuda()
    with torch.no_grad():
        net.flat.mul_(1.7)
    net.mark_weights_changed()
    return net


@pytest.mark.parametrize("n_rays", [1, 3, 40, 700])
def test_chain2_inference_forward_is_bit_identical(S, chain2_switch, n_rays):
    net = _net(S)
    pts = torch.randn(n_rays, 67, 3, device="cuda") * 1.5      # 67 samples per ray: ragged last tile
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device="cuda"), dim=-1)
    out = []
    for v in (0, 1):
        chain2_switch(v)
        with torch.no_grad():
            out.append(net.query(pts, vd).clone())
    assert torch.isfinite(out[0]).all()
    assert torch.equal(out[0].view(torch.int32), out[1].view(torch.int32))


@pytest.mark.parametrize("n_rays", [1, 40, 400])
def test_chain2_dgrad_gives_bit_identical_gradients(S, chain2_switch, n_rays):
    net = _net(S)
    pts = torch.randn(n_rays, 192, 3, device="cuda")
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device="cuda"), dim=-1)
    go = torch.randn(n_rays, 192, 4, device="cuda")
    grads = []
    for v in (0, 1):
        chain2_switch(v)
        net.flat.grad = None
        net.query(pts, vd).backward(go)
        grads.append(net.flat.grad.clone())
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0].view(torch.int32), grads[1].view(torch.int32))
