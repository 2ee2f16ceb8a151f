"""GPU: bench.py's world_size > 1 path (process group, broadcast, hook-started gradient all-reduce, barrier + MAX
timing, clean exit) on the one GPU of the test box: two ranks share cuda:0 and use gloo, because RCCL refuses two
ranks on one device.  The driver's multi-GPU runs use RCCL; everything else in the path is the same code."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_ranks_on_one_gpu(launcher):
    """launcher = "torchrun": the driver's multi-GPU command line; "self": plain `python bench.py --gpus 2` with WORLD_SIZE
    unset — bench.py must start the two ranks itself (fresh children, the parent never touches the GPU) and relay rank 0's
    line (it used to run ONE rank and print n_gpus = 1)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SNR_BENCH_SAME_DEVICE="1", SNR_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-frame"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["global_batch_rays"] == 2048 and d["config"]["parallelism"] == "ray-dp2"
    assert d["value"] > 0 and abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d                    # rank 0 at N=1 only
    # the run describes itself: what the collective layer saw, and the exposed communication time
    dd = d["distributed"]
    assert dd["ranks_seen"] == 2 and dd["backend"] == "gloo" and dd["local_device_of_rank"] == [0, 0]
    assert dd["gradient_bytes_per_step"] == 2 * 595844 * 4
    assert dd["allreduce_ms_per_step_exposed"] is not None and dd["allreduce_ms_per_step_exposed"] >= 0.0
    # every rank's own step time, not only the MAX
    assert len(dd["ms_per_step_per_rank"]) == 2 and all(0 < t <= d["ms_per_step"] * 1.5 for t in dd["ms_per_step_per_rank"])
    assert d["blocks"] == len(d["block_ms"]) >= 1 and abs(sorted(d["block_ms"])[len(d["block_ms"]) // 2] / 6 - d["ms_per_step"]) < 1e-3
