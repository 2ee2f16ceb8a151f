#!/usr/bin/env python3
"""The multi-GPU measurement of SURVEY.md §8(e) in ONE call (VERDICT r05 "next round" item 7): bench.py at 1, 2, 4 and 8 ranks x
the three ways the data-parallel step can exchange its gradients, one table.

    python3 tools/scale_matrix.py [--gpus 1,2,4,8] [--steps 20] [--warmup 5] [--out profiles/rNN_scale_matrix.md] [--same-device]

  merged    both networks' backward as one launch sequence, ONE all-reduce of the 4.77 MB gradient (the default route)
  split     the same launch sequence, one all-reduce per network          (SNR_SPLIT_ALLREDUCE=1)
  overlap   fine backward, its all-reduce UNDER the coarse backward       (SNR_OVERLAP_ALLREDUCE=1)

Every run is `python bench.py --gpus N ...` started as a FRESH CHILD process of this script, which never touches the GPU itself
(no torch.cuda call here): bench.py in turn starts its N ranks as fresh children through torch.distributed.run before its own
parent initialises anything — no re-exec from a process that holds the GPU anywhere.  Give this file directly as the program
(`gpurun -- python3 tools/scale_matrix.py`, or behind `rocprofv3 ... --`): no shell, env or launcher wrapper in between.

--same-device: every rank on cuda:0 with the gloo backend (SNR_BENCH_SAME_DEVICE=1 / SNR_BENCH_BACKEND=gloo) — how the
whole script is exercised end to end on a ONE-GPU box today (RCCL refuses two ranks per device); the numbers of such a run
say nothing about xGMI and the table says so in its title.

Columns: whole-job rays/s (bench.py's `value`), ms/step (MAX over ranks), speed-up over the same variant at 1 GPU, each rank's
own ms/step, the EXPOSED all-reduce time per step (HIP events around the waits in apply_gradients), sharded 378x504 frame ms."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("merged", {}), ("split", {"SNR_SPLIT_ALLREDUCE": "1"}), ("overlap", {"SNR_OVERLAP_ALLREDUCE": "1"})]


def run(n, env_extra, ns):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SNR_SPLIT_ALLREDUCE",
                                                            "SNR_OVERLAP_ALLREDUCE")}
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ns.same_device:
        env.update(SNR_BENCH_SAME_DEVICE="1", SNR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(ns.steps), "--warmup", str(ns.warmup),
           "--no-cpu-baseline", "--no-hashgrid", "--sustain-s", str(ns.sustain_s)]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=ns.timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return None, f"exit {r.returncode}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else 'no output'}", time.time() - t0
    return json.loads(lines[0]), None, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sustain-s", type=float, default=2.0)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale_matrix.md"))
    ap.add_argument("--same-device", action="store_true")
    ns = ap.parse_args()
    counts = [int(x) for x in ns.gpus.split(",")]
    rows, raw, base = [], [], {}
    for name, env in VARIANTS:
        for n in counts:
            if n == 1 and name != "merged" and 1 in counts:
                continue      # (one rank has no collective: the three variants differ at N > 1 only — except `overlap`'s launch structure)
            d, err, wall = run(n, env, ns)
            raw.append({"variant": name, "gpus": n, "line": d, "error": err, "driver_wall_s": wall})
            if d is None:
                rows.append(f"| {name} | {n} | FAILED: {err} | | | | | |")
                continue
            if n == 1:
                base[name] = d["value"]
            b = base.get(name) or base.get("merged")
            dist = d.get("distributed") or {}
            per_rank = ", ".join(f"{t:.3f}" for t in dist.get("ms_per_step_per_rank") or [d["ms_per_step"]])
            sus = d.get("sustained") or {}
            rows.append(f"| {name} | {n} | {d['value']:.0f} | {d['ms_per_step']:.4f} | {(d['value'] / b if b else float('nan')):.2f} | {per_rank} | "
                        f"{(dist.get('allreduce_ms_per_step_exposed') if dist else None)} | {dist.get('ms_per_frame_sharded', d.get('ms_per_frame_378x504'))} | "
                        f"{sus.get('ms_per_step')} |")
            print(rows[-1], flush=True)
    title = ("# ONE-GPU REHEARSAL (all ranks on cuda:0, gloo): exercises the script, says nothing about xGMI\n"
             if ns.same_device else "# data-parallel scaling over RCCL / xGMI, one node\n")
    os.makedirs(os.path.dirname(os.path.abspath(ns.out)), exist_ok=True)
    with open(ns.out, "w") as f:
        f.write(title + f"\n`python3 tools/scale_matrix.py --gpus {ns.gpus} --steps {ns.steps} --warmup {ns.warmup}`"
                + (" --same-device" if ns.same_device else "") + "; weak scaling, 1024 rays x (64 + 128) samples per rank, bf16\n\n"
                "| all-reduce variant | GPUs | rays/s (whole job) | ms/step (MAX over ranks) | x over 1 GPU | ms/step of each rank | exposed all-reduce ms/step | frame ms (sharded by rows at N > 1) | sustained ms/step |\n"
                "|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
    with open(os.path.splitext(ns.out)[0] + ".jsonl", "w") as f:
        for r in raw:
            f.write(json.dumps(r) + "\n")
    print("wrote", ns.out, flush=True)
    return 0 if all(r["line"] is not None for r in raw) else 1


if __name__ == "__main__":
    sys.exit(main())
