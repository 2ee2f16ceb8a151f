"""GPU: the `nccl` (= RCCL) backend code path of the data-parallel step with ONE rank.  A test box has one
GPU and RCCL refuses two ranks per device, so the 2-rank tests use gloo; what they cannot exercise is the backend itself:
`init_process_group("nccl", device_id=...)`, an `all_reduce(async_op=True)` on the buffer a ctypes-launched kernel has just
written on torch's current stream, `work.wait()` before Adam reads it, and a clean teardown.  With SNR_FORCE_COLLECTIVES=1
RenderTrainer runs exactly that with world_size = 1 (the reduce is the identity): the step must equal the step without
collectives bit for bit — RCCL's stream hand-off neither loses nor reorders the gradient — in the merged route (one
all-reduce of both networks' gradients), the split route and the SPIn-NeRF iteration."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, importlib, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
from test_gpu_dist_step import _setup
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
res = {}
for force in ("1", "0"):
    for split in ("0", "1", "overlap"):
        os.environ["SNR_FORCE_COLLECTIVES"] = force
        os.environ["SNR_SPLIT_ALLREDUCE"] = "1" if split == "1" else "0"
        # round 6: the fine network's all-reduce under the coarse network's backward (two launch sequences; compared with the
        # one-rank run of the SAME launch structure)
        os.environ["SNR_OVERLAP_ALLREDUCE"] = "1" if split == "overlap" else "0"
        train, kw, nets, hwf, rays, target, rnd = _setup("mlp")
        tr = train.RenderTrainer(kw, lrate=5e-4, world_size=1)
        assert tr._dist == (force == "1")
        cu = {k: v.cuda() for k, v in rnd.items()}
        for it in range(3):
            loss, _ = tr.step(*hwf, rays.cuda(), target.cuda(), randoms=cu)
        n = rays.shape[1] // 3
        r3 = [rays[:, i * n:(i + 1) * n].cuda() for i in range(3)]
        rr = [{k: v[i * n:(i + 1) * n].cuda() for k, v in rnd.items()} for i in range(3)]
        l2, _ = tr.spin_iteration(*hwf, r3[0], target[:n].cuda(), r3[1], target[n:2 * n].cuda(), r3[2],
                                  (torch.rand(n, generator=torch.Generator().manual_seed(1)) * 0.3 + 0.1).cuda(), randoms=rr)
        torch.cuda.synchronize()
        res[(force, split)] = (float(loss), float(l2), [x.flat.detach().clone() for x in nets])
for key, (l, l2, params) in res.items():
    base = res[("0", "overlap" if key[1] == "overlap" else "0")]
    # (the loss VALUE is a sum of per-workgroup atomics: its last bits depend on their order; the parameters do not)
    assert abs(l - base[0]) < 1e-6 * abs(base[0]) and abs(l2 - base[1]) < 1e-6 * abs(base[1]), (key, l, base[0], l2, base[1])
    for a, b in zip(params, base[2]):
        assert torch.equal(a, b), key
# ... and the two launch structures agree to the split-K partition of the samples (DESIGN.md 4.2)
for a, b in zip(res[("0", "overlap")][2], res[("0", "0")][2]):
    assert float((a - b).norm() / b.norm()) < 1e-3
dist.barrier()
dist.destroy_process_group()
print("NCCL_ONE_RANK_OK")
'''


def test_nccl_backend_with_one_rank_runs_the_collective_path_and_changes_nothing():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SNR_FORCE_COLLECTIVES", "SNR_SPLIT_ALLREDUCE", "SNR_OVERLAP_ALLREDUCE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
