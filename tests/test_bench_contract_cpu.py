"""CPU: the committed bench line (profiles/rNN_bench_default.json, a plain `python bench.py` on an MI355X) carries the
fields the measurement contract names, with consistent arithmetic; and bench.py's model of the workload (MACs per sample
of the three passes) agrees with the network shapes the oracle defines."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed bench line"
    lines = [l for l in open(files[-1]) if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_committed_bench_line_has_the_contract_fields():
    d = _latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "training rays/sec" and d["unit"] == "rays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value is the whole-job rate of the timed region
    assert abs(d["value"] - d["config"]["global_batch_rays"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # median of the timed blocks
    assert abs(sorted(d["block_ms"])[len(d["block_ms"]) // 2] / d["steps"] - d["ms_per_step"]) < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    # HBM bytes per launch from the PMC file cannot exceed what 8 TB/s moves in the launch's duration
    assert r["traffic"] is None or r["traffic"] < 8e12 * r["avg_launch_ms"] * 1e-3
    assert d["hbm_bytes_per_step"] is None or d["hbm_bytes_per_step"] < 8e12 * d["ms_per_step"] * 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # round 6: the rate is shown to hold for seconds, the frame has a roofline of its own, the CPU baseline
    # is reported at the best thread count AND at os.cpu_count()
    s = d["sustained"]
    assert s["seconds"] >= 2.5 and s["steps"] >= d["steps"] and abs(s["ms_per_step"] - s["seconds"] / s["steps"] * 1e3) < 1e-9
    assert abs(s["ratio_to_ms_per_step"] - s["ms_per_step"] / d["ms_per_step"]) < 1e-9 and s["holds"] == (s["ratio_to_ms_per_step"] <= 1.03)
    f = d["frame_roofline"]
    assert f["bound"] == "mfma" and abs(f["flops_per_frame"] - 57.88e12) < 0.01e12 and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-9
    assert abs(f["achieved"] - f["flops_per_frame"] / (d["ms_per_frame_378x504"] * 1e-3) / 1e12) < 1e-6 * f["achieved"]
    a = c["all_cores"]
    assert a["cores"] >= c["cores"] and a["unit"] == c["unit"] and a["value"] > 0


def test_every_kernels_key_of_the_bench_line_is_a_kernel_of_the_rocprofv3_trace():
    """bench.py's `kernels` keys are the names of the kernels that run (csrc/prof.cpp names them after the
    __global__ functions): every key + "_kernel" must be the prefix of a kernel name in the SAME round's `--kernel-trace --stats`
    CSV, every launch is counted as a launch (no "+ 1"), and the step's launches add up to what the trace shows per step.
    (Rounds before 5 used coarser labels: composite_fwd, make_rays, adam.)"""
    import csv
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    tag = re.match(r"(r\d+)_", os.path.basename(files[-1])).group(1)
    stats = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
    if int(tag[1:]) < 5 or not os.path.exists(stats):
        import pytest
        pytest.skip("no round >= 5 profile committed yet")
    d = _latest_line()
    names = [r["Name"] for r in csv.DictReader(open(stats))]
    for k in d["kernels"]:
        assert any((k + "_kernel") in n for n in names), (k, "not a kernel of", stats)
    assert abs(d["launches_per_step"] - sum(v["launches_per_step"] for v in d["kernels"].values())) < 1e-9


def test_bench_flop_model_matches_the_network_shapes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # forward MACs per sample of the reference network (run_nerf_helpers.py:66-127, viewdirs): 8 trunk layers with the
    # skip after layer 4, feature + alpha heads, the 283 -> 128 views layer, rgb
    mac = 63 * 256 + 3 * 256 * 256 + (256 + 63) * 256 + 3 * 256 * 256 + 256 * 256 + 256 * 1 + (256 + 27) * 128 + 128 * 3
    assert b.MAC_FWD == mac
    # the layer pairs the recompute kernel handles: layer 0 (63 inputs) and seven 256 x 256 layers
    assert b.MAC_WGRAD_PAIR == 63 * 256 + 7 * 256 * 256


def test_bench_gpus_n_without_a_launcher_starts_n_ranks_and_fails_loudly_when_they_fail():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must not quietly measure one GPU: it starts two
    child ranks through torch.distributed.run.  Here there is no GPU, so both ranks stop with bench.py's "needs an MI355X"
    and the launcher must relay that as a non-zero exit code and print no JSON line.  (The working path runs on the GPU
    box: tests/test_gpu_bench_dist.py.)  A --gpus / WORLD_SIZE mismatch is an error as well."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by tests/test_gpu_bench_dist.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "a rank failed" in r.stderr


def test_scale_matrix_fails_loudly_without_gpus(tmp_path):
    """tools/scale_matrix.py (the pre-staged 1/2/4/8-GPU x three all-reduce variants measurement): every bench.py run is a fresh
    child; a run that fails shows as a FAILED row and a non-zero exit code, never as a missing row.  (Its working path on one
    GPU: profiles/r06_scale_matrix_rehearsal.md, `--same-device`.)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by the rehearsal run on a GPU box")
    out = tmp_path / "m.md"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_matrix.py"), "--gpus", "1,2", "--steps", "1", "--warmup", "0",
                        "--out", str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1
    txt = out.read_text()
    assert txt.count("FAILED") == 4 and "| merged | 1 |" in txt and "| overlap | 2 |" in txt and "| split | 2 |" in txt
    assert len(open(str(out)[:-3] + ".jsonl").read().splitlines()) == 4
