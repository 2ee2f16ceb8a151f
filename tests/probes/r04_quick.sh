#!/bin/bash
# quick GPU check of a build: backward / train-step tests, then a short bench with the per-kernel breakdown
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_quick_tests.txt
cat gpurun_out/r04_quick_tests.txt
python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame > gpurun_out/r04_quick_bench.json 2> gpurun_out/r04_quick_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_quick_bench.json"):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("step %.4f ms" % d["ms_per_step"], " profiled %.4f" % d["ms_per_step_profiled"])
        for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print("  %-18s %.4f ms  x%.0f" % (n, v["ms_per_step"], v["launches_per_step"]))
PY
tail -5 gpurun_out/r04_quick_bench.err
