"""CPU, 2 processes over gloo: the data-parallel contract of RenderTrainer.

Rays shard by rank; each rank back-propagates the mean loss of its shard; gradients are summed by
one all-reduce per net and scaled by 1/world inside the Adam kernel.  The HIP kernels cannot run
here, so the per-shard gradients come from the CPU oracle and the Adam kernel is replaced by the
oracle's Adam — what is under test is the collective/scale/broadcast logic of spin-nerf_amd/train.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    import spin_nerf_amd as S
    train = importlib.import_module("spin-nerf_amd.train")
    from oracle import nerf_oracle as O

    torch.manual_seed(100 + rank)          # different init per rank: broadcast must fix that
    net = S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)
    tr = train.RenderTrainer({"network_fn": net, "network_fine": None}, lrate=1e-2, world_size=world)
    tr.broadcast_parameters()
    flat0 = net.flat.detach().clone()

    # this rank's shard of a fixed global batch; gradient of the shard-mean loss from the oracle
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(8, 4, 3, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(8, 3, generator=g), dim=-1)
    tgt = torch.randn(8, 4, 4, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)

    def oracle_grad(p, d, t):
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.named_views(flat0).items()}
        loss = ((O.run_network(sd, p, d) - t) ** 2).mean()
        loss.backward()
        return torch.cat([sd[k].grad.reshape(-1) for k in sd])

    net.flat.grad = oracle_grad(pts[sl], dirs[sl], tgt[sl])

    def cpu_adam(params, grads, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        gsc = grads * grad_scale
        m.mul_(beta1).add_(gsc, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gsc, gsc, value=1 - beta2)
        params.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt() + eps, value=-lr)
    train.ops.adam_step_ = cpu_adam
    tr.apply_gradients()

    # single-process reference: full batch mean == average of equal-size shard means
    g_full = oracle_grad(pts, dirs, tgt)
    p_ref, m, v = flat0.clone(), torch.zeros_like(flat0), torch.zeros_like(flat0)
    cpu_adam(p_ref, g_full, m, v, tr.lrate * (0.1 ** (0 / (tr.lrate_decay * 1000))), 1)
    # lr decay uses the post-increment step like run_nerf.py:1616-1620 applies it after the step
    err_g = float((net.flat.grad / world - g_full).abs().max() / g_full.abs().max())
    if rank == 0:
        torch.save({"flat0": flat0, "flat": net.flat.detach(), "err_g": err_g, "p_ref": p_ref}, out)
    # second step, gradient through autograd: the post-accumulate hook starts the all-reduce inside backward()
    net.flat.grad = None
    c = torch.full_like(net.flat.data, float(rank + 1))
    (net.flat * c).sum().backward()
    assert len(tr._works) == 1, "hook did not start the all-reduce"
    tr.apply_gradients()
    assert not tr._works
    assert torch.equal(net.flat.grad, torch.full_like(c, float(sum(range(1, world + 1)))))
    # every rank must hold identical parameters after the step
    gathered = [torch.zeros_like(net.flat.data) for _ in range(world)]
    dist.all_gather(gathered, net.flat.data)
    assert all(torch.equal(gathered[0], x) for x in gathered)

    # round 4: the direct step writes BOTH networks' gradients into the two halves of one buffer and exchanges them with a
    # single all-reduce whose work object stands for both networks (train.py: _step_direct); apply_gradients must wait for it
    # once per network, start no collective of its own, and leave identical replicas
    torch.manual_seed(7)                   # same init on every rank
    nets = [S.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True) for _ in range(2)]
    tr2 = train.RenderTrainer({"network_fn": nets[0], "network_fine": nets[1]}, lrate=1e-2, world_size=world)
    n0 = nets[0].flat.numel()
    g_both = torch.cat([torch.full((n0,), float(rank + 1)), torch.full((nets[1].flat.numel(),), float(10 * (rank + 1)))])
    nets[0].flat.grad, nets[1].flat.grad = g_both[:n0], g_both[n0:]
    work = dist.all_reduce(g_both, op=dist.ReduceOp.SUM, async_op=True)
    tr2._works[0] = tr2._works[1] = work
    started = []
    tr2._start_all_reduce = lambda i, p: started.append(i)
    tr2.apply_gradients()
    assert not started and not tr2._works
    tot = float(sum(range(1, world + 1)))
    assert torch.equal(nets[0].flat.grad, torch.full((n0,), tot)) and torch.equal(nets[1].flat.grad, torch.full_like(nets[1].flat.grad, 10 * tot))
    for n in nets:
        gathered = [torch.zeros_like(n.flat.data) for _ in range(world)]
        dist.all_gather(gathered, n.flat.data)
        assert all(torch.equal(gathered[0], x) for x in gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["err_g"] < 1e-5                      # sum / world == full-batch gradient
    # the parameters moved, and to (nearly) where a single process would have put them
    assert float((r["flat"] - r["flat0"]).abs().max()) > 1e-4
    assert float((r["flat"] - r["p_ref"]).abs().max()) < 2e-4


def _sharded_worker(rank, world, port, out):
    """render_sharded's band arithmetic and all_gather with a stand-in renderer (the HIP kernels need a GPU):
    the assembled frame must equal the single-process frame for heights that do and do not divide evenly."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spin_nerf_amd as S

    def fake_render(H, W, focal, chunk=0, c2w=None, patch=None, **kw):
        i0, j0, h, w = patch
        ii = torch.arange(i0, i0 + h, dtype=torch.float32)[:, None].expand(h, w)
        jj = torch.arange(j0, j0 + w, dtype=torch.float32)[None, :].expand(h, w)
        base = ii * 100 + jj + float(c2w[0, 3])
        return [torch.stack([base, base + .25, base + .5], -1), base * 2, base * 3, base * 4, {}]

    ok = True
    for H in (8, 7, 1):
        W = 5
        c2w = torch.eye(4)[:3, :4].clone(); c2w[0, 3] = 0.125
        got = S.render_sharded(H, W, 10.0, c2w, 64, {}, render_fn=fake_render)
        ref = fake_render(H, W, 10.0, c2w=c2w, patch=(0, 0, H, W))
        ok = ok and all(torch.equal(a, b) for a, b in zip(got, ref[:4]))
    if rank == 0:
        torch.save({"ok": ok}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_frame_equals_single_process(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sharded.pt")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.load(out)["ok"]
