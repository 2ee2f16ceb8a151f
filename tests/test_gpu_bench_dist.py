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
def test_bench_two_ranks_on_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SNR_BENCH_SAME_DEVICE="1", SNR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--no-frame"]
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
    assert d["blocks"] == len(d["block_ms"]) >= 1 and abs(sorted(d["block_ms"])[len(d["block_ms"]) // 2] / 6 - d["ms_per_step"]) < 1e-3
