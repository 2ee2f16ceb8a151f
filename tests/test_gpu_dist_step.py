"""GPU, two ranks on the one device of the test box (gloo; RCCL refuses two ranks per device — the collective call is the
same `dist.all_reduce` either way): RenderTrainer.step on rank-sharded rays.  The all-reduced gradient of every network,
scaled by 1/world, must equal the gradient a single process computes on the whole global batch, and after the step all
replicas hold identical parameters ("8-rank result on a fixed global batch == 1-rank result")."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(kind="mlp"):
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import spin_nerf_amd as S
    from oracle import nerf_oracle as O
    train = importlib.import_module("spin-nerf_amd.train")
    H, W, focal, near, far = 20, 24, 30.0, 2.0, 6.0
    Nc, Nf, N = 64, 32, 48       # N = global batch, split in two
    nets = []
    for seed in (3, 4):
        if kind == "hash":      # the reference's default networks (create_nerf_tcnn): one flat [table | MLPs] buffer each
            from oracle import hashgrid_oracle as HG
            sd = HG.init_params(seed)
            sd["encoder.params"] = sd["encoder.params"] * 3e3
            n = S.NeRF_TCNN().cuda()
            n.load_state_dict(sd)
        else:
            n = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True, precision="fp32").cuda()
            n.load_state_dict(O.init_nerf_params(seed=seed))
        nets.append(n)

    def q(inputs, viewdirs, network_fn):
        return S.run_network(inputs, viewdirs, network_fn)
    q._snr_fused = True
    kw = dict(network_query_fn=q, perturb=1.0, N_importance=Nf, network_fine=nets[1], N_samples=Nc, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=True, raw_noise_std=1.0, ndc=False, lindisp=True, near=near, far=far)
    g = torch.Generator().manual_seed(0)
    c2w = torch.eye(4)[:3, :4].clone(); c2w[2, 3] = 4.0
    ro, rd = O.get_rays(H, W, focal, c2w)
    sel = torch.randperm(H * W, generator=g)[:N]
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    target = torch.rand(N, 3, generator=g)
    rnd = {"t_rand": torch.rand(N, Nc, generator=g), "u": torch.rand(N, Nf, generator=g),
           "noise_c": torch.randn(N, Nc, generator=g), "noise_f": torch.randn(N, Nc + Nf, generator=g)}
    return train, kw, nets, (H, W, focal), rays, target, rnd


def _worker(rank, world, port, out, kind):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train, kw, nets, hwf, rays, target, rnd = _setup(kind)
    tr = train.RenderTrainer(kw, lrate=5e-4, world_size=world)
    tr.broadcast_parameters()
    n = rays.shape[1] // world
    sl = slice(rank * n, (rank + 1) * n)
    tr.step(*hwf, rays[:, sl].cuda().contiguous(), target[sl].cuda(), randoms={k: v[sl].cuda() for k, v in rnd.items()})
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"grad_sum": [x.flat.grad.cpu() for x in nets], "params": [x.flat.detach().cpu() for x in nets]}, out)
    else:
        torch.save({"params": [x.flat.detach().cpu() for x in nets]}, out + ".1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["mlp", "hash", "mlp-overlap"])
def test_two_rank_step_equals_single_process_step_on_the_global_batch(tmp_path, kind, monkeypatch):
    import torch.multiprocessing as mp
    if kind == "mlp-overlap":     # round 6: the fine network's all-reduce under the coarse network's backward (train.py: _backward_two)
        monkeypatch.setenv("SNR_OVERLAP_ALLREDUCE", "1")
        kind = "mlp"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, out, kind), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".1")
    # single process, whole batch
    train, kw, nets, hwf, rays, target, rnd = _setup(kind)
    tr = train.RenderTrainer(kw, lrate=5e-4)
    tr.step(*hwf, rays.cuda(), target.cuda(), randoms={k: v.cuda() for k, v in rnd.items()})
    for i, net in enumerate(nets):
        g1 = net.flat.grad.cpu().double()
        g2 = r0["grad_sum"][i].double() / 2        # the trainer folds 1 / world into the Adam kernel
        rel = float((g1 - g2).norm() / g1.norm())
        # (hash networks: bf16 kernels, and the table gradient is summed by atomics in a different order)
        assert rel < (1e-5 if kind == "mlp" else 2e-3), f"net {i}: reduced gradient differs from the single-process gradient by {rel:.2e}"
        assert torch.equal(r0["params"][i], r1["params"][i]), "replicas diverged"
        # (parameters vs the single-process step: equal up to elements whose gradient rounds to the other sign — Adam moves
        #  every element by lr whatever the gradient's size, see tests/test_gpu_train_step.py)
        d = (r0["params"][i] - net.flat.detach().cpu()).abs()
        assert float((d > 1e-7).float().mean()) < (2e-3 if kind == "mlp" else 0.2)
